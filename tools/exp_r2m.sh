#!/bin/bash
# multi-launch weight gradients on / off
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for m in 0 1; do
  DSL_WGRAD_MULTI=$m python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-prof --no-dsl > gpurun_out/r2m_bench_m${m}.log 2>&1
  echo "multi=$m $(grep -h '"value"' gpurun_out/r2m_bench_m${m}.log | python -c "import sys,json; j=json.loads(sys.stdin.readline()); print(j['value'], j['ms_per_step'], j['final_losses'])")"
  tail -3 gpurun_out/r2m_bench_m${m}.log | grep -v '"value"' | cut -c1-300
done
