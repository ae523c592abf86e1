"""The recurrent path of an RLA block (conv_out -> BN + tanh -> 3x3) on the DSL iteration's shapes: three launches against
dsl_rla_tail_fwd in its two tile forms (library option rla_tail_form).  Usage (GPU box): python tools/bench_rla_tail.py"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch

from dsl_amd import _lib as L
from dsl_amd import ops


def timeit(fn, iters=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


g = torch.Generator().manual_seed(0)
WS = torch.empty(64 << 20, dtype=torch.uint8, device='cuda')
for stage, (h, w, c4, tw) in enumerate([(200, 336, 256, 64), (100, 168, 512, 128), (50, 84, 1024, 128), (25, 42, 2048, 128)]):
    for n in (2, 1):
        P, ldx = n * h * w, c4 + 128
        xh = (torch.randn(P, ldx, generator=g) * 0.7).bfloat16().cuda()
        hin = (torch.randn(P, 32, generator=g) * 0.5).bfloat16().cuda()
        wco = torch.zeros(64, c4)
        wco[:32] = torch.randn(32, c4, generator=g) / c4 ** 0.5
        wco = wco.bfloat16().cuda()
        wrc = torch.zeros(64, 3, 3, tw)
        wrc[:32, :, :, :32] = torch.randn(32, 3, 3, 32, generator=g) * 0.08
        wrc = wrc.bfloat16().cuda()
        sc, bi = (1 + 0.3 * torch.randn(32, generator=g)).cuda(), (0.2 * torch.randn(32, generator=g)).cuda()
        u = torch.zeros(P, 32, dtype=torch.bfloat16, device='cuda')
        t = torch.zeros(P, tw, dtype=torch.bfloat16, device='cuda')
        nxt = torch.zeros(P, ldx, dtype=torch.bfloat16, device='cuda')
        hout = nxt.data_ptr() + c4 * 2
        d1 = ops.conv_desc(xh, wco, u, n=n, grid=[(h, w)], src_hw=[(h, w)], dst_hw=[(h, w)], cs=c4, cd=32, cd_pad=64, ldd=32, kh=1, kw=1,
                           stride=1, pad=0, lds=ldx, addend=hin, lda=32, workspace=WS)
        d2 = ops.conv_desc(t, wrc, hout, n=n, grid=[(h, w)], src_hw=[(h, w)], dst_hw=[(h, w)], cs=tw, cd=32, cd_pad=64, ldd=ldx, kh=3,
                           kw=3, stride=1, pad=1, workspace=WS)
        sp = L.stream_ptr()

        def three():
            L.lib.dsl_conv2d(C.byref(d1), sp)
            L.lib.dsl_bn_tanh_fwd(L.ptr(u), 32, L.ptr(sc), L.ptr(bi), L.ptr(t), tw, P, 32, sp)
            L.lib.dsl_conv2d(C.byref(d2), sp)

        def fused():
            L.lib.dsl_rla_tail_fwd(L.ptr(xh), ldx, L.ptr(hin), 32, L.ptr(wco), c4, L.ptr(sc), L.ptr(bi), L.ptr(wrc), tw, L.ptr(u), L.ptr(t),
                                   C.c_void_p(hout), ldx, n, h, w, sp)
        res = [f'stage {stage} n={n} {h}x{w} c4={c4}: three launches {timeit(three):6.1f} us']
        for form in (1, 2):
            L.lib.dsl_set_option(b'rla_tail_form', form)
            res.append(f'form {form}: {timeit(fused):6.1f} us')
        L.lib.dsl_set_option(b'rla_tail_form', 0)
        print('  '.join(res), flush=True)
