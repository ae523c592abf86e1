#!/bin/bash
# regression tower's last data gradient: FORK -> side stream -> JOIN (0) vs JOIN + on the caller's stream (1)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for rep in 1 2 3; do for m in 0 1; do
  DSL_REG_FINAL_MAIN=$m python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-prof --no-dsl > gpurun_out/r3e_bench_${m}.log 2>&1
  echo "reg_final_main=$m $(grep -h '"value"' gpurun_out/r3e_bench_${m}.log | python -c "import sys,json; j=json.loads(sys.stdin.readline()); print(j['value'], j['ms_per_step'])")"
done; done
