#!/bin/bash
# when the weight gradients leave for the side stream: tower halves (two groups of 4 early) vs one group of 8 at the end;
# layer2 (last segment) grouped at its end vs per block as soon as ready
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for h in 0 1; do for g in 0 1; do
  DSL_TOWER_HALVES=$h DSL_GROUP_LAST=$g python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-prof --no-dsl > gpurun_out/r2t_bench_h${h}_g${g}.log 2>&1
  echo "halves=$h group_last=$g $(grep -h '"value"' gpurun_out/r2t_bench_h${h}_g${g}.log | python -c "import sys,json; j=json.loads(sys.stdin.readline()); print(j['value'], j['ms_per_step'])")"
done; done
