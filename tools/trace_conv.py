"""Where one workgroup's K loop spends its cycles: s_memtime stamps of conv_pipe_kernel (tools/build_trace.sh) on the head-tower
shape (or a tools/conv_cost.py shape).  Per wave and K tile, the gaps between the six stamps:
  issue   0 -> 1   reads + DMA pieces + every pre-barrier MFMA issued (the in-order wave blocks on the busy matrix pipe here)
  lgkm    1 -> 2   this wave's outstanding LDS reads
  vmcnt   2 -> 3   the next tile's DMA still in flight
  barrier 3 -> 4   waiting for the other waves
  post    4 -> 5   first reads of the next tile, first DMA pieces, the held-back MFMAs
Usage: DSL_TRACE_WG=40 python tools/trace_conv.py [head|l3b|...] [force_cfg]"""
import ctypes as C
import os
import sys
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
os.environ.setdefault('DSL_HIP_LIB', os.path.join(ROOT, 'dsl_amd', 'lib', 'libdsl_hip_trace.so'))
os.environ.setdefault('DSL_TRACE_WG', '40')
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import numpy as np
import torch
from dsl_amd import _lib as L
from dsl_amd import ops
SHAPES = {'p3': (256, 256, 3, (100, 168)), 'l3a': (1024, 256, 1, (50, 84)), 'l3b': (256, 256, 3, (50, 84)), 'l3c': (256, 1024, 1, (50, 84)), 'l2b': (128, 128, 3, (100, 168)),
          'l4b': (512, 512, 3, (25, 42))}
which = sys.argv[1] if len(sys.argv) > 1 else 'head'
force = int(sys.argv[2]) if len(sys.argv) > 2 else 0
LEVELS = [(100, 168), (50, 84), (25, 42), (13, 21), (7, 11)]
N = 2
if which == 'head':
    ci, co, k, lv = 256, 256, 3, LEVELS
else:
    ci, co, k, hw = SHAPES[which]
    lv = [hw]
P = sum(h * w for h, w in lv) * N
x = torch.randn(P, ci, device='cuda').bfloat16()
w = (torch.randn(co, k, k, ci, device='cuda') * 0.05).bfloat16()
y = torch.empty(P, co, device='cuda', dtype=torch.bfloat16)
ws = torch.empty(128 << 20, dtype=torch.uint8, device='cuda')
d = ops.conv_desc(x, w, y, n=N, grid=lv, src_hw=lv, dst_hw=lv, cs=ci, cd=co, cd_pad=co, ldd=co, kh=k, kw=k, stride=1, pad=k // 2,
                  flags=L.CONV_RELU_OUT | (force << 8), workspace=ws)
for _ in range(5):
    L.lib.dsl_conv2d(C.byref(d), L.stream_ptr())
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    L.lib.dsl_conv2d(C.byref(d), L.stream_ptr())
e1.record()
torch.cuda.synchronize()
print(f'{which}: {e0.elapsed_time(e1) * 100:.1f} us per launch (instrumented build)')
buf = np.zeros(8 * 40 * 8 + 8 * 16 + 8, np.uint64)
L.lib.dsl_debug_conv_trace.argtypes = [C.c_void_p]
assert L.lib.dsl_debug_conv_trace(buf.ctypes.data) == 0
nw = int(buf[8 * 40 * 8 + 8 * 16])
epi = buf[8 * 40 * 8:8 * 40 * 8 + 8 * 16].reshape(8, 16)[:nw].astype(np.int64)
t = buf[:8 * 40 * 8].reshape(8, 40, 8)[:nw].astype(np.int64)
iters = min(40, ci * k * k // 64 - 1)           # the loop body runs ktiles - 1 times (split-K launches: fewer - check the period line)
t = t[:, :iters, :6]
t0 = t[:, 0, 0].min()
names = ['issue', 'lgkm', 'vmcnt', 'barrier', 'post']
gaps = np.diff(t, axis=2)                         # [wave][iter][5]
period = np.diff(t[:, :, 0], axis=1)              # loop top to loop top
print(f'{nw} waves, {iters} K tiles stamped; s_memtime ticks (100 MHz constant clock if the stamps look quantised; else shader cycles)')
print('loop period per K tile (ticks): mean %.0f  min %d  max %d' % (period.mean(), period.min(), period.max()))
print('wave  ' + '  '.join(f'{n:>8s}' for n in names) + '    sum')
for wv in range(nw):
    g = gaps[wv, 2:].mean(0)
    print(f'{wv:4d}  ' + '  '.join(f'{v:8.0f}' for v in g) + f'  {g.sum():6.0f}')
print('every wave, stamps of K tiles 4..5 relative to the earliest tile-4 loop top:')
base = t[:, 4, 0].min()
for wv in range(nw):
    print(f'   wave {wv}:', (t[wv, 4] - base).tolist(), (t[wv, 5] - base).tolist())
print('arrival skew at the barrier (stamp 3, max - min over waves), K tiles 2..: mean %.0f' % (t[:, 2:, 3].max(0) - t[:, 2:, 3].min(0)).mean())
print('release skew after the barrier (stamp 4): mean %.0f' % (t[:, 2:, 4].max(0) - t[:, 2:, 4].min(0)).mean())

# epilogue: stamps relative to the earliest epilogue entry
e0 = epi[:, 0].min()
print('epilogue stamps per wave, cycles from the first wave entering the epilogue: entry | per slab: barrier 1, slab staged + barrier 2, stores issued | all stores acknowledged')
npt = 3 if which in ('head', 'p3') and force in (0, 1) else 2
for wv in range(nw):
    print(f'   wave {wv}:', int(epi[wv, 0] - e0), [[int(epi[wv, 1 + 3 * r + j] - e0) for j in range(3)] for r in range(npt)], int(epi[wv, 15] - e0))
print('K loop end -> epilogue entry (last stamp 5 to entry, wave 0): %d' % (epi[0, 0] - t[0, iters - 1, 5]))
