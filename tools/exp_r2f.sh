#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for k in 256 224 192 160 128; do
  DSL_WGRAD_SLOTS=$k python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-prof > gpurun_out/r2f_bench_slots$k.log 2>&1
  grep -h '"value"' gpurun_out/r2f_bench_slots$k.log | python -c "
import sys, json
for l in sys.stdin:
    j = json.loads(l); print('slots', $k, j['value'], j['ms_per_step'])
"
done
