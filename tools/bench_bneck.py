"""Micro-benchmark of the fused bottleneck forward (dsl_bottleneck_fwd, csrc/bneck.hip) against the three dsl_conv2d launches it replaces,
on ResNet-50's layer2 / layer3 shapes of N images of 800 x 1344 (back-to-back replays, HIP events).
Usage (GPU box): python tools/bench_bneck.py [N]"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch

from dsl_amd import _lib as L
from dsl_amd import ops

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2
dev = 'cuda'
WS = torch.empty(128 << 20, dtype=torch.uint8, device=dev)


def timeit(fn, iters=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3          # us


for name, P, H, W, NBLK in (('layer2', 128, 100, 168, 3), ('layer3', 256, 50, 84, 5)):
    C4 = 4 * P
    xs = [torch.randn(N, H, W, C4, device=dev).bfloat16() for _ in range(NBLK + 1)]       # the stage as it runs: block b reads xs[b], writes xs[b + 1]
    a1 = torch.empty(N, H, W, P, device=dev, dtype=torch.bfloat16)
    a2 = torch.empty_like(a1)
    blocks = []
    for b in range(NBLK):
        w1 = (torch.randn(P, C4, device=dev) / C4 ** 0.5).bfloat16()
        w2 = (torch.randn(P, 9 * P, device=dev) / (9 * P) ** 0.5).bfloat16()
        w3 = (torch.randn(C4, P, device=dev) / P ** 0.5).bfloat16()
        sb = [(torch.rand(c, device=dev) + 0.5, torch.randn(c, device=dev) * 0.3) for c in (P, P, C4)]
        x, out = xs[b], xs[b + 1]
        d1 = ops.conv_desc(x, w1, a1, n=N, grid=[(H, W)], src_hw=[(H, W)], dst_hw=[(H, W)], cs=C4, cd=P, cd_pad=P, ldd=P, kh=1, kw=1, stride=1, pad=0,
                           flags=L.CONV_RELU_OUT, scale=sb[0][0], bias=sb[0][1], workspace=WS)
        d2 = ops.conv_desc(a1, w2, a2, n=N, grid=[(H, W)], src_hw=[(H, W)], dst_hw=[(H, W)], cs=P, cd=P, cd_pad=P, ldd=P, kh=3, kw=3, stride=1, pad=1,
                           flags=L.CONV_RELU_OUT, scale=sb[1][0], bias=sb[1][1], workspace=WS)
        d3 = ops.conv_desc(a2, w3, out, n=N, grid=[(H, W)], src_hw=[(H, W)], dst_hw=[(H, W)], cs=P, cd=C4, cd_pad=C4, ldd=C4, kh=1, kw=1, stride=1, pad=0,
                           flags=L.CONV_RELU_OUT, scale=sb[2][0], bias=sb[2][1], addend=x, lda=C4, workspace=WS)
        bd = ops.bneck_desc(x, w1, w2, w3, x, (sb[0][0].data_ptr(), sb[0][1].data_ptr()), (sb[1][0].data_ptr(), sb[1][1].data_ptr()),
                            (sb[2][0].data_ptr(), sb[2][1].data_ptr()), a1, a2, out, n=N, hin=H, win=W, h=H, w=W, planes=P, cin=C4)
        blocks.append((d1, d2, d3, bd, (w1, w2, w3, sb)))
    d1, d2, d3, bd, _ = blocks[0]
    t = [timeit(lambda d=d: L.check(L.lib.dsl_conv2d(C.byref(d), L.stream_ptr()))) for d in (d1, d2, d3)]

    def three():
        for d1_, d2_, d3_, _, _ in blocks:
            for d in (d1_, d2_, d3_):
                L.lib.dsl_conv2d(C.byref(d), L.stream_ptr())

    def fused():
        for blk in blocks:
            L.lib.dsl_bottleneck_fwd(C.byref(blk[3]), L.stream_ptr())
    t3 = timeit(three) / NBLK
    tf = timeit(fused) / NBLK
    tf1 = timeit(lambda: L.check(L.lib.dsl_bottleneck_fwd(C.byref(bd), L.stream_ptr())))
    flops = 2.0 * N * H * W * (P * C4 + 9 * P * P + C4 * P)
    if os.environ.get('DSL_BNECK_DBG_PROBE'):          # ablation build only (tools/build_ablate.sh): the kernel cut off behind conv1 / conv2
        ph = []
        for dbg in (1, 2):
            os.environ['DSL_BNECK_DBG'] = str(dbg)
            ph.append(timeit(lambda: L.check(L.lib.dsl_bottleneck_fwd(C.byref(bd), L.stream_ptr()))))
        os.environ['DSL_BNECK_DBG'] = '0'
        print(f'{name}: fused kernel by phase (cut-off probe): conv1 + halo {ph[0]:5.1f} us, + conv2 {ph[1]:5.1f} us, whole {tf1:5.1f} us')
    print(f'{name}: conv1 {t[0]:6.1f}  conv2 {t[1]:6.1f}  conv3 {t[2]:6.1f}  per block of a {NBLK}-block chain: three launches {t3:6.1f} us ({flops / t3 / 1e6:5.0f} TF)'
          f'   fused {tf:6.1f} us ({flops / tf / 1e6:5.0f} TF)   [one block replayed: fused {tf1:6.1f} us]')
