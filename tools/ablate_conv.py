import ctypes as C, os, sys
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
force = int(sys.argv[1]) if len(sys.argv) > 1 else 1
d = ops.conv_desc(x, w, y, n=N, grid=LEVELS, src_hw=LEVELS, dst_hw=LEVELS, cs=256, cd=256, cd_pad=256, ldd=256, kh=3, kw=3,
                  stride=1, pad=1, flags=L.CONV_RELU_OUT | (force << 8))
for _ in range(3):
    L.lib.dsl_conv2d(C.byref(d), L.stream_ptr())
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    L.lib.dsl_conv2d(C.byref(d), L.stream_ptr())
e1.record(); torch.cuda.synchronize()
print('cfg %d DSL_ABLATE=%s  %.1f us per conv' % (force, os.environ.get('DSL_ABLATE', '0'), e0.elapsed_time(e1) * 50))
