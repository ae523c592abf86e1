"""dsl_conv1x1_pair against the two dsl_conv2d launches it replaces, at the layer2 / layer3 shapes of the N = 2 step.
Usage (GPU box): python tools/bench_pair.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dsl_amd import _lib as L
from dsl_amd import ops
import ctypes as C


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for name, P, M in (('layer2', 128, 33600), ('layer3', 256, 8400)):
    for mode in ('forward', 'backward'):
        fwd = mode == 'forward'
        a = torch.randn(M, P, device='cuda').bfloat16()
        wa = (torch.randn(4 * P, P, device='cuda') * P ** -0.5).bfloat16()
        wb = (torch.randn(P, 4 * P, device='cuda') * (4 * P) ** -0.5).bfloat16()
        add = torch.randn(M, 4 * P, device='cuda').bfloat16()
        s1, b1 = (torch.rand(4 * P, device='cuda') + 0.5, torch.randn(4 * P, device='cuda')) if fwd else (None, None)
        s2, b2 = (torch.rand(P, device='cuda') + 0.5, torch.randn(P, device='cuda')) if fwd else (None, None)
        m1 = None if fwd else torch.randn(M, 4 * P, device='cuda').bfloat16()
        m2 = None if fwd else torch.randn(M, P, device='cuda').bfloat16()
        mid = torch.empty(M, 4 * P, device='cuda', dtype=torch.bfloat16)
        out = torch.empty(M, P, device='cuda', dtype=torch.bfloat16)
        ws = torch.empty(64 << 20, dtype=torch.uint8, device='cuda')
        d = ops.pair_desc(a, wa, mid, wb, out, m=M, p=P, scale1=s1, bias1=b1, addend=add, ldadd=4 * P, mask1=m1, ldm1=4 * P, relu1=fwd,
                          scale2=s2, bias2=b2, mask2=m2, ldm2=P, relu2=fwd)
        f1 = (L.CONV_RELU_OUT if fwd else 0) | (0 if fwd else L.CONV_MASK_LAST)
        c1 = ops.conv_desc(a, wa, mid, n=1, grid=[(1, M)], src_hw=[(1, M)], dst_hw=[(1, M)], cs=P, cd=4 * P, cd_pad=4 * P, ldd=4 * P, kh=1,
                           kw=1, flags=f1, scale=s1, bias=b1, addend=add, lda=4 * P, mask=m1, ldm=4 * P, workspace=ws)
        c2 = ops.conv_desc(mid, wb, out, n=1, grid=[(1, M)], src_hw=[(1, M)], dst_hw=[(1, M)], cs=4 * P, cd=P, cd_pad=P, ldd=P, kh=1, kw=1,
                           flags=f1, scale=s2, bias=b2, mask=m2, ldm=P, workspace=ws)
        tp = timeit(lambda: L.lib.dsl_conv1x1_pair(C.byref(d), L.stream_ptr()))
        t2 = timeit(lambda: (L.lib.dsl_conv2d(C.byref(c1), L.stream_ptr()), L.lib.dsl_conv2d(C.byref(c2), L.stream_ptr())))
        fl = 2 * 2.0 * M * 4 * P * P
        print(f'{name} {mode:8s} P={P} M={M}: pair {tp:6.1f} us ({fl / tp / 1e6:5.0f} TF)   two launches {t2:6.1f} us', flush=True)
