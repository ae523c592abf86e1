"""Times the GroupNorm+ReLU forward / backward kernels on the FCOS tower shape (N x 5 levels x 256 ch)."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dsl_amd import _lib as L
from dsl_amd import ops
N = 2
LEVELS = [(100, 168), (50, 84), (25, 42), (13, 21), (7, 11)]
M = sum(h * w for h, w in LEVELS) * N
x = torch.randn(M, 256, device='cuda').bfloat16()
y = torch.empty_like(x)
dy = torch.randn(M, 256, device='cuda').bfloat16()
dx = torch.empty_like(x)
gamma = torch.rand(256, device='cuda') + 0.5
beta = torch.randn(256, device='cuda') * 0.1
stats = torch.zeros(5 * N * 32, 2, device='cuda')
dg, db = torch.zeros(256, device='cuda'), torch.zeros(256, device='cuda')
df = ops.gn_desc(x, y, gamma, beta, stats, n=N, hw=LEVELS)
dbw = ops.gn_desc(x, y, gamma, beta, stats, n=N, hw=LEVELS, dy=dy, dx=dx, dgamma=dg, dbeta=db, dbias=torch.zeros(256, device='cuda'))


def timeit(fn, it=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3


tf = timeit(lambda: L.lib.dsl_groupnorm_relu_fwd(C.byref(df), L.stream_ptr()))
tb = timeit(lambda: L.lib.dsl_groupnorm_relu_bwd(C.byref(dbw), L.stream_ptr()))
mb = M * 256 * 2 / 1e6
print(f'GN fwd {tf:.1f} us ({3 * mb / tf * 1e-3 * 1e3:.0f} GB/s on the 3-pass floor {3 * mb:.0f} MB)   '
      f'GN bwd {tb:.1f} us ({5 * mb / tb * 1e-3 * 1e3:.0f} GB/s on the 5-pass floor {5 * mb:.0f} MB)')
