#!/bin/bash
# workgroup slots of the weight-gradient launches with the multi-launch structure
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for s in 128 160 192 224 256; do
  DSL_WGRAD_SLOTS=$s python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-prof --no-dsl > gpurun_out/r2o_bench_s${s}.log 2>&1
  echo "slots=$s $(grep -h '"value"' gpurun_out/r2o_bench_s${s}.log | python -c "import sys,json; j=json.loads(sys.stdin.readline()); print(j['value'], j['ms_per_step'])")"
done
