#!/bin/bash
# pipelined frozen prefix (next step's image layout + stem + pool + layer1 under the previous backward's tail): off / on
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for rep in 1 2; do for m in 0 1; do
  DSL_BENCH_PIPE=$m python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-prof --no-dsl > gpurun_out/r3d_bench_${m}.log 2>&1
  echo "pipe_prefix=$m $(grep -h '"value"' gpurun_out/r3d_bench_${m}.log | python -c "import sys,json; j=json.loads(sys.stdin.readline()); print(j['value'], j['ms_per_step'], j['final_losses'])")"
  tail -2 gpurun_out/r3d_bench_${m}.log | grep -v '"value"' | cut -c1-200
done; done
