mkdir -p gpurun_out
python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_dsl_outlier_benchorder.txt
import sys, os, time, gc
sys.path.insert(0, os.getcwd())
import torch, bench
from dsl_amd import detectors
rows = []
class Probe:
    priority = 95
    def __getattr__(self, name): return lambda r: None
    def before_train_iter(self, r):
        self.t0 = time.perf_counter(); st = torch.cuda.memory_stats(); self.a0 = (st.get('num_device_alloc', 0), st.get('num_device_free', 0))
    def after_train_iter(self, r):
        st = torch.cuda.memory_stats()
        rows.append((r.iter, round((time.perf_counter() - self.t0) * 1e3, 2), st.get('num_device_alloc', 0) - self.a0[0], st.get('num_device_free', 0) - self.a0[1]))
out = bench.dsl_iteration_timing(extra_hook=Probe(), raw=True)
print(out['spread'])
for k, v in out['raw'].items(): print(k, v)
print('host rows (iter, host ms, device allocs, frees) with allocs or > 6 ms:', [r for r in rows if r[2] or r[3] or r[1] > 6])
PY
