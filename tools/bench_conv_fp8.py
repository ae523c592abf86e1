"""Head-tower convolution (3x3 256->256 over the 5 FPN levels, N images of 800x1344): bf16 kernel vs the fp8 kernel, with and
without the activation quantisation pass.  Usage (GPU box): python tools/bench_conv_fp8.py [N]"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dsl_amd import _lib as L
from dsl_amd import ops
from dsl_amd.engine import OpList
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2
LEVELS = [(100, 168), (50, 84), (25, 42), (13, 21), (7, 11)]
P = sum(h * w for h, w in LEVELS) * N
ci = co = 256
x = torch.relu(torch.randn(P, ci, device='cuda')).bfloat16()
w32 = torch.randn(co, 9 * ci, device='cuda') * 0.05
w16 = w32.bfloat16()
y = torch.empty(P, co, device='cuda', dtype=torch.bfloat16)
x8 = torch.empty(P, ci, dtype=torch.uint8, device='cuda')
w8 = torch.empty(co, 9 * ci, dtype=torch.uint8, device='cuda')
comb = torch.empty(co, device='cuda')
ws = torch.empty(64 << 20, dtype=torch.uint8, device='cuda')
L.check(L.lib.dsl_quant_fp8_weights(L.ptr(w32), L.ptr(w8), L.ptr(comb), None, co, co, 9 * ci, 1 / 16.0, L.stream_ptr()))
d16 = ops.conv_desc(x, w16, y, n=N, grid=LEVELS, src_hw=LEVELS, dst_hw=LEVELS, cs=ci, cd=co, cd_pad=co, ldd=co, kh=3, kw=3, stride=1, pad=1, workspace=ws)
d8 = ops.conv_desc(x8, w8, y, n=N, grid=LEVELS, src_hw=LEVELS, dst_hw=LEVELS, cs=ci, cd=co, cd_pad=co, ldd=co, kh=3, kw=3, stride=1, pad=1,
                   flags=L.CONV_FP8, scale=comb)
flops = 2.0 * P * co * ci * 9


def run(fn, reps=50):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def oplist(descs):
    ol = OpList()
    for _ in range(16):
        for d in descs:
            ol.conv(d)
    return ol


o16, o8 = oplist([d16]), oplist([d8])
t16 = run(o16.run, 10) / 16
t8 = run(o8.run, 10) / 16
tq = run(lambda: L.lib.dsl_quant_fp8(L.ptr(x), L.ptr(x8), P, ci, ci, 16.0, L.stream_ptr()))
print(f'head 3x3 256->256, {P} px: bf16 {t16:.1f} us = {flops / t16 / 1e6:.0f} TFLOP/s | fp8 {t8:.1f} us = {flops / t8 / 1e6:.0f} TFLOP/s '
      f'| activation quantisation pass {tq:.1f} us ({P * ci * 3 / tq / 1e3:.0f} GB/s)')
