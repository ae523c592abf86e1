"""A handful of launches of one conv shape for PMC profiling (tools/pmc_conv.sh).
Usage: python tools/prof_conv.py SHAPE [force_cfg]     SHAPE: head | fpn | a name of tools/conv_cost.py (l3a, l3b1, ...)"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from dsl_amd import _lib as L
from dsl_amd import ops
from conv_cost import SHAPES
which = sys.argv[1] if len(sys.argv) > 1 else 'head'
force = int(sys.argv[2]) if len(sys.argv) > 2 else 0
LEVELS = [(100, 168), (50, 84), (25, 42), (13, 21), (7, 11)]
N = 1 if which.endswith('1') else 2
if which == 'head':
    ci, co, k, lv = 256, 256, 3, LEVELS
elif which == 'fpn':
    ci, co, k, lv = 256, 256, 3, LEVELS[:1]
else:
    ci, co, k, hw = SHAPES[which.rstrip('1')]
    lv = [hw]
P = sum(h * w for h, w in lv) * N
dev = 'cuda'
x = torch.randn(P, ci, device=dev).bfloat16()
w = (torch.randn(co, k, k, ci, device=dev) * 0.05).bfloat16()
y = torch.empty(P, co, device=dev, dtype=torch.bfloat16)
ws = torch.empty(128 << 20, dtype=torch.uint8, device=dev)
d = ops.conv_desc(x, w, y, n=N, grid=lv, src_hw=lv, dst_hw=lv, cs=ci, cd=co, cd_pad=co, ldd=co, kh=k, kw=k,
                  stride=1, pad=k // 2, flags=L.CONV_RELU_OUT | (force << 8), workspace=ws)
for it in range(6):
    L.lib.dsl_conv2d(C.byref(d), L.stream_ptr())
torch.cuda.synchronize()
print('done')
