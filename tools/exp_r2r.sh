#!/bin/bash
# backward: FPN / lateral / downsample branches on side stream 3: on / off
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for rep in 1 2; do for m in 0 1; do
  DSL_BWD_BRANCH=$m python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-prof --no-dsl > gpurun_out/r2r_bench_b${m}.log 2>&1
  echo "bwd_branch=$m $(grep -h '"value"' gpurun_out/r2r_bench_b${m}.log | python -c "import sys,json; j=json.loads(sys.stdin.readline()); print(j['value'], j['ms_per_step'])")"
done; done
