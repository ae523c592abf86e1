"""Micro-benchmark of the conv kernels on the real FCOS R50-FPN shapes (N images of 800x1344).
Usage (GPU box): python tools/bench_conv.py [N]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time

import torch

from dsl_amd import _lib as L
from dsl_amd import ops

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2
FORCE = [int(x) for x in sys.argv[2].split(',')] if len(sys.argv) > 2 else [0]
LEVELS = [(100, 168), (50, 84), (25, 42), (13, 21), (7, 11)]
dev = 'cuda'
WS = torch.empty(128 << 20, dtype=torch.uint8, device=dev)


def timeit(fn, iters=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def bench(name, ci, co, k, s, sizes_in):
    p = k // 2
    sizes_out = [((h + 2 * p - k) // s + 1, (w + 2 * p - k) // s + 1) for h, w in sizes_in]
    pin = sum(h * w for h, w in sizes_in) * N
    pout = sum(h * w for h, w in sizes_out) * N
    x = torch.randn(pin, ci, device=dev).bfloat16()
    w = (torch.randn(co, k, k, ci, device=dev) * 0.05).bfloat16()
    wt = (torch.randn(ci, k, k, co, device=dev) * 0.05).bfloat16()
    y = torch.empty(pout, co, device=dev, dtype=torch.bfloat16)
    dy = torch.randn(pout, co, device=dev).bfloat16()
    dx = torch.empty(pin, ci, device=dev, dtype=torch.bfloat16)
    dw = torch.empty(co, k, k, ci, device=dev)
    flops = 2.0 * pout * co * ci * k * k
    import ctypes as C
    res = [f'{name:28s}']
    for force in FORCE:
        bco = {1: 256, 2: 256, 3: 128, 4: 128, 5: 64, 6: 128, 7: 64, 8: 64, 9: 256, 10: 256}.get(force & 15)
        if bco and co % bco:
            res.append(f'f{force}: n/a')
            continue
        dfw = ops.conv_desc(x, w, y, n=N, grid=sizes_out, src_hw=sizes_in, dst_hw=sizes_out, cs=ci, cd=co, cd_pad=co,
                            ldd=co, kh=k, kw=k, stride=s, pad=p, flags=L.CONV_RELU_OUT | ((force & 15) << 8), workspace=WS if force < 16 else None)
        t_f = timeit(lambda: L.lib.dsl_conv2d(C.byref(dfw), L.stream_ptr()))
        res.append(f'f{force}: {t_f*1e6:6.1f}us {flops/t_f/1e12:5.0f}TF')
    if s == 1:
        ddg = ops.conv_desc(dy, wt, dx, n=N, grid=sizes_in, src_hw=sizes_out, dst_hw=sizes_in, cs=co, cd=ci, cd_pad=ci,
                            ldd=ci, kh=k, kw=k, stride=s, pad=p, mode=1, mask=x, ldm=ci, flags=L.CONV_MASK_LAST, workspace=WS)
        t_d = timeit(lambda: L.lib.dsl_conv2d(C.byref(ddg), L.stream_ptr()))
        res.append(f'dgrad {t_d*1e6:8.1f} us {flops/t_d/1e12:7.1f} TF')
    if ci % 128 == 0:
        for cfg in (0, None):
            dwg = ops.wgrad_desc(dy, x, dw, n=N, grid=sizes_out, src_hw=sizes_in, cs=ci, cy=co, cd=co, kh=k, kw=k, stride=s, pad=p, force_cfg=cfg)
            t_w = timeit(lambda: L.lib.dsl_conv2d_wgrad(C.byref(dwg), L.stream_ptr()))
            res.append(f'wgrad[{cfg}] {t_w*1e6:7.1f}us {flops/t_w/1e12:5.0f}TF')
    print('  '.join(res), flush=True)


bench('head tower 3x3 256 (5 lvls)', 256, 256, 3, 1, LEVELS)
bench('fpn 3x3 256 @P3', 256, 256, 3, 1, LEVELS[:1])
bench('l1 3x3 64->64 @200x336', 64, 64, 3, 1, [(200, 336)])
bench('l1 1x1 64->256', 64, 256, 1, 1, [(200, 336)])
bench('l1 1x1 256->64', 256, 64, 1, 1, [(200, 336)])
bench('l2 3x3 128 @100x168', 128, 128, 3, 1, [(100, 168)])
bench('l2 1x1 128->512', 128, 512, 1, 1, [(100, 168)])
bench('l2 1x1 512->128', 512, 128, 1, 1, [(100, 168)])
bench('l3 3x3 256 @50x84', 256, 256, 3, 1, [(50, 84)])
bench('l3 1x1 256->1024', 256, 1024, 1, 1, [(50, 84)])
bench('l3 1x1 1024->256', 1024, 256, 1, 1, [(50, 84)])
bench('l4 3x3 512 @25x42', 512, 512, 3, 1, [(25, 42)])
bench('l4 1x1 2048->512', 2048, 512, 1, 1, [(25, 42)])
bench('l4 1x1 512->2048', 512, 2048, 1, 1, [(25, 42)])
bench('l4 3x3 512 @25x42', 512, 512, 3, 1, [(25, 42)])
bench('l4 1x1 512->2048', 512, 2048, 1, 1, [(25, 42)])
bench('l4 1x1 2048->512', 2048, 512, 1, 1, [(25, 42)])
bench('l3.0 ds 1x1 s2 512->1024', 512, 1024, 1, 2, [(100, 168)])
