"""Where the host's time per training step goes.  Two runs of the same loop under cProfile:
  * as shipped (the host runs ahead of the GPU until the runtime's queues push back), and
  * with every kernel launch of dsl_run_ops skipped (`lib.skip_kinds`: ordering ops still run), which leaves the Python + ctypes +
    event bookkeeping cost alone - the difference between the two is what the launches themselves cost the host.
Usage: python tools/host_profile.py [--skip]"""
import cProfile
import os
import pstats
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if '--skip' in sys.argv:
    os.environ['DSL_TUNE'] = (os.environ.get('DSL_TUNE', '') + ',lib.skip_kinds=%d' % 0x3fffffff).lstrip(',')
import torch
import bench
from dsl_amd import detectors  # noqa: F401
from dsl_amd.optim import FlatSGD
from dsl_amd.registry import build_detector
model = build_detector(bench.model_cfg()).cuda()
model.lazy_log = True
opt = FlatSGD(model, lr=0.01, momentum=0.9, weight_decay=1e-4, paramwise_cfg=dict(bias_lr_mult=2., bias_decay_mult=0.))
batch = bench.synth_batch(0, 2)


def step():
    out = model.train_step(batch, opt)
    out['loss'].backward()
    opt.step()


for _ in range(6):
    step()
torch.cuda.synchronize()
K = 40
marks = []
t0 = time.perf_counter()
for _ in range(K):
    a = time.perf_counter()
    step()
    marks.append(time.perf_counter() - a)
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
marks.sort()
print('%s: host queues a step in %.3f ms mean (min %.3f, median %.3f, max %.3f); the GPU completes one every %.3f ms' % (
    'kernel launches skipped' if '--skip' in sys.argv else 'as shipped', t_host / K * 1e3, marks[0] * 1e3, marks[K // 2] * 1e3,
    marks[-1] * 1e3, t_all / K * 1e3))
pr = cProfile.Profile()
pr.enable()
for _ in range(K):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats('tottime')
print('per-step host time by function (tottime / %d steps), top 28:' % K)
rows = sorted(st.stats.items(), key=lambda kv: -kv[1][2])[:28]
for (fn, line, name), (cc, nc, tt, ct, _) in rows:
    print('  %8.1f us  %6.1f calls  %s:%d %s' % (tt / K * 1e6, nc / K, os.path.basename(fn), line, name))
