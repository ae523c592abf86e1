#!/bin/bash
# predictor weight gradients at the start of the backward pass vs with the towers' at the end of their chains
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for rep in 1 2; do for e in 0 1; do
  DSL_PRED_EARLY=$e python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-prof --no-dsl > gpurun_out/r2u_bench_e${e}.log 2>&1
  echo "pred_early=$e $(grep -h '"value"' gpurun_out/r2u_bench_e${e}.log | python -c "import sys,json; j=json.loads(sys.stdin.readline()); print(j['value'], j['ms_per_step'])")"
done; done
