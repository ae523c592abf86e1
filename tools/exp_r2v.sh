#!/bin/bash
# stream priorities: side streams at the lowest priority (a: weight-gradient stream only, b: all three)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for rep in 1 2; do for pr in 0 a b; do
  DSL_SIDE_PRIO=$pr python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-prof --no-dsl > gpurun_out/r2v_bench_${pr}.log 2>&1
  echo "side_prio=$pr $(grep -h '"value"' gpurun_out/r2v_bench_${pr}.log | python -c "import sys,json; j=json.loads(sys.stdin.readline()); print(j['value'], j['ms_per_step'])")"
done; done
