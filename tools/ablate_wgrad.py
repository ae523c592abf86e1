"""Head-tower weight gradient under the ablation knobs (needs tools/build_ablate.sh):
  for a in 0 1 2 3 4 5 6 7; do DSL_HIP_LIB=$PWD/dsl_amd/lib/libdsl_hip_ablate.so DSL_ABLATE=$a python tools/ablate_wgrad.py; done
bits (wgrad_pipe): 1 = no DMA after the ring's first fill, 2 = no MFMA, 4 = no partial-tile writes, 8 = no LDS fragment reads,
16 = no barrier.  FORCE_CFG=<n> forces a tile configuration (default: the product's choice)."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dsl_amd import _lib as L
from dsl_amd import ops
N = 2
LEVELS = [(100, 168), (50, 84), (25, 42), (13, 21), (7, 11)]
P = sum(h * w for h, w in LEVELS) * N
dev = 'cuda'
x = torch.randn(P, 256, device=dev).bfloat16()
dy = torch.randn(P, 256, device=dev).bfloat16()
dw = torch.empty(256, 3, 3, 256, device=dev)
d = ops.wgrad_desc(dy, x, dw, n=N, grid=LEVELS, src_hw=LEVELS, cs=256, cy=256, cd=256, kh=3, kw=3, stride=1, pad=1, force_cfg=(int(os.environ['FORCE_CFG']) if 'FORCE_CFG' in os.environ else None))
for _ in range(3):
    L.lib.dsl_conv2d_wgrad(C.byref(d), L.stream_ptr())
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    L.lib.dsl_conv2d_wgrad(C.byref(d), L.stream_ptr())
e1.record(); torch.cuda.synchronize()
print('DSL_ABLATE=%s  %.1f us per wgrad op (incl. reduce)' % (os.environ.get('DSL_ABLATE', '0'), e0.elapsed_time(e1) * 100))
