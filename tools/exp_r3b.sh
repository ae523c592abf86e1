#!/bin/bash
# grid: workgroup budget of the multi / group weight-gradient launches x the towers' group
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for ws in 96 128 160 192; do for ts in 64 96 128; do
  DSL_WGRAD_SLOTS=$ws DSL_TOWER_SLOTS=$ts python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-prof --no-dsl > gpurun_out/r3b_bench_${ws}_${ts}.log 2>&1
  echo "wgrad_slots=$ws tower_slots=$ts $(grep -h '"value"' gpurun_out/r3b_bench_${ws}_${ts}.log | python -c "import sys,json; j=json.loads(sys.stdin.readline()); print(j['value'], j['ms_per_step'])")"
done; done
