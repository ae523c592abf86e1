"""layer1's middle convolution (3x3 / 1, 64 -> 64, BatchNorm + ReLU, N x 200 x 336): the activation-stationary kernel of
csrc/patch3.hip against the implicit-GEMM kernels of dsl_conv2d on the same operands.
Usage (GPU box): python tools/bench_patch3.py [N]"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch

from dsl_amd import _lib as L
from dsl_amd import ops

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2
H, W = 200, 336
dev = 'cuda'


def timeit(fn, iters=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


x = torch.randn(N, H, W, 64, device=dev).bfloat16()
w = (torch.randn(64, 3, 3, 64, device=dev) / 24).bfloat16()
sc, bi = torch.rand(64, device=dev) + 0.5, torch.randn(64, device=dev)
flops = 2.0 * N * H * W * 64 * 64 * 9
byts = 2.0 * N * H * W * 64 * 2
outs = {}
for name, env, force in (('patch3', '1', 0), ('implicit GEMM (auto)', '0', 0), ('implicit GEMM 128x128', '0', 3), ('implicit GEMM 128x64', '0', 5),
                         ('implicit GEMM 64x64', '0', 6)):
    os.environ['DSL_PATCH3'] = env
    y = torch.empty(N, H, W, 64, device=dev, dtype=torch.bfloat16)
    d = ops.conv_desc(x, w, y, n=N, grid=[(H, W)], src_hw=[(H, W)], dst_hw=[(H, W)], cs=64, cd=64, cd_pad=64, ldd=64, kh=3, kw=3,
                      stride=1, pad=1, flags=L.CONV_RELU_OUT | (force << 8), scale=sc, bias=bi)
    t = timeit(lambda: L.lib.dsl_conv2d(C.byref(d), L.stream_ptr()))
    outs[name] = y.clone()
    print(f'{name:24s} {t * 1e6:7.1f} us  {flops / t / 1e12:6.0f} TFLOP/s  {byts / t / 1e9:6.0f} GB/s algorithmic', flush=True)
ref = outs['patch3']
for k, v in outs.items():
    print(f'{k:24s} bit-identical to patch3: {bool(torch.equal(v, ref))}')
