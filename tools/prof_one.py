"""Run a handful of launches of the head-tower conv (fwd) and wgrad for PMC profiling."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dsl_amd import _lib as L
from dsl_amd import ops
N = 2
LEVELS = [(100, 168), (50, 84), (25, 42), (13, 21), (7, 11)]
P = sum(h * w for h, w in LEVELS) * N
dev = 'cuda'
x = torch.randn(P, 256, device=dev).bfloat16()
w = (torch.randn(256, 3, 3, 256, device=dev) * 0.05).bfloat16()
y = torch.empty(P, 256, device=dev, dtype=torch.bfloat16)
dy = torch.randn(P, 256, device=dev).bfloat16()
dw = torch.empty(256, 3, 3, 256, device=dev)
ws = torch.empty(128 << 20, dtype=torch.uint8, device=dev)
dfw = ops.conv_desc(x, w, y, n=N, grid=LEVELS, src_hw=LEVELS, dst_hw=LEVELS, cs=256, cd=256, cd_pad=256, ldd=256, kh=3, kw=3,
                    stride=1, pad=1, flags=L.CONV_RELU_OUT, workspace=ws)
cfgs = [int(a) for a in sys.argv[1:]] or [0, 1]
for it in range(3):
    L.lib.dsl_conv2d(C.byref(dfw), L.stream_ptr())
    for cfg in cfgs:
        d = ops.wgrad_desc(dy, x, dw, n=N, grid=LEVELS, src_hw=LEVELS, cs=256, cy=256, cd=256, kh=3, kw=3, stride=1, pad=1, force_cfg=cfg)
        L.lib.dsl_conv2d_wgrad(C.byref(d), L.stream_ptr())
torch.cuda.synchronize()
print('done')
