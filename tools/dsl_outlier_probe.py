"""Where do the long intervals of the semi-supervised iteration come from?  Runs one variant of bench.py's dsl_iteration timing
(refresh, rla, async as 0/1; default 1 1 1) for more iterations and prints, per iteration: the GPU interval between the iterations'
end events, the host's time in the iteration, garbage collections that ran inside it, and the caching allocator's device
allocations / frees inside it.   python tools/dsl_outlier_probe.py [refresh rla async [steps]]"""
import gc
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from dsl_amd import detectors  # noqa: F401

v = tuple(bool(int(a)) for a in (sys.argv[1:4] if len(sys.argv) >= 4 else ('1', '1', '1')))
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 60
warm = 6
rows = []
gcs = []


def on_gc(phase, info):
    if phase == 'start':
        gcs.append([info['generation'], time.perf_counter(), None])
    elif gcs:
        gcs[-1][2] = time.perf_counter()


gc.callbacks.append(on_gc)


class Probe:
    priority = 95

    def __init__(self):
        self.t0 = None

    def __getattr__(self, name):
        return lambda r: None

    def before_train_iter(self, r):
        self.t0 = time.perf_counter()
        st = torch.cuda.memory_stats()
        self.a0 = (st.get('num_device_alloc', 0), st.get('num_device_free', 0), st.get('num_alloc_retries', 0))

    def after_train_iter(self, r):
        t1 = time.perf_counter()
        st = torch.cuda.memory_stats()
        a1 = (st.get('num_device_alloc', 0), st.get('num_device_free', 0), st.get('num_alloc_retries', 0))
        g = [(g_[0], round((g_[2] - g_[1]) * 1e3, 2)) for g_ in gcs if g_[2] is not None and g_[1] >= self.t0 and g_[1] <= t1]
        rows.append(dict(it=r.iter, host_ms=round((t1 - self.t0) * 1e3, 3), t_end=t1, gc=g,
                         dev_alloc=a1[0] - self.a0[0], dev_free=a1[1] - self.a0[1], retries=a1[2] - self.a0[2]))


out = bench.dsl_iteration_timing(steps=steps, warm=warm, variants=(v,), extra_hook=Probe(), raw=True)
key = [k for k in out['raw']][0]
gaps = out['raw'][key]
med = sorted(gaps)[len(gaps) // 2]
print(key, 'median %.3f ms, min %.3f, max %.3f over %d intervals' % (med, min(gaps), max(gaps), len(gaps)))
# interval k lies between the end events of iterations warm - 1 + k and warm + k
byit = {r_['it']: r_ for r_ in rows}
prev_end = None
for k, g_ in enumerate(gaps):
    it = warm + k
    r_ = byit.get(it, {})
    between = round((r_['t_end'] - byit[it - 1]['t_end']) * 1e3, 3) if it - 1 in byit and r_ else None
    flag = ' <-- long' if g_ > 1.3 * med else ''
    print('it %3d  gpu interval %8.3f ms  host in-iteration %7.3f ms  host end-to-end %8s ms  gc %s  device alloc/free/retry %s/%s/%s%s' % (
        it, g_, r_.get('host_ms', -1), between, r_.get('gc'), r_.get('dev_alloc'), r_.get('dev_free'), r_.get('retries'), flag))
