#!/bin/bash
# backward of the two head towers on two streams: on / off
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for rep in 1 2; do for m in 0 1; do
  DSL_BWD_TOWERS=$m python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-prof --no-dsl > gpurun_out/r2q_bench_t${m}.log 2>&1
  echo "bwd_towers=$m $(grep -h '"value"' gpurun_out/r2q_bench_t${m}.log | python -c "import sys,json; j=json.loads(sys.stdin.readline()); print(j['value'], j['ms_per_step'])")"
done; done
