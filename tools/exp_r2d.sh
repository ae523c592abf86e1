#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 > gpurun_out/r2d_tests.log
python tools/bench_gn.py > gpurun_out/r2d_bench_gn.log 2>&1
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-prof > gpurun_out/r2d_bench.log 2>&1
cat gpurun_out/r2d_tests.log gpurun_out/r2d_bench_gn.log
grep -h '"value"' gpurun_out/r2d_bench.log | python -c "
import sys, json
for l in sys.stdin:
    j = json.loads(l); print(j['value'], j['ms_per_step'], j['final_losses'])
"
