#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "groupnorm" 2>&1 | tail -5 > gpurun_out/r2c_tests.log
python tools/bench_gn.py > gpurun_out/r2c_bench_gn.log 2>&1
python tools/probe_cu_mask.py > gpurun_out/r2c_cumask.log 2>&1
for k in 0 8 12 16 20; do
  DSL_SIDE_CUS=$k python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-prof > gpurun_out/r2c_bench_cus$k.log 2>&1
done
cat gpurun_out/r2c_tests.log gpurun_out/r2c_bench_gn.log gpurun_out/r2c_cumask.log
for k in 0 8 12 16 20; do grep -h '"value"' gpurun_out/r2c_bench_cus$k.log | python -c "
import sys, json
for l in sys.stdin:
    j = json.loads(l); print('cus', $k, j['value'], j['ms_per_step'], j['final_losses']['loss'])
"; done
