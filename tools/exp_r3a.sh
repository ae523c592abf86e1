#!/bin/bash
# fused conv3 -> next conv1 pairs in the forward pass: off / layer3 / layer2+3
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for rep in 1 2; do for m in "" 3 23; do
  DSL_PAIR_FWD=$m python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-prof --no-dsl > gpurun_out/r3a_bench_${m:-off}.log 2>&1
  echo "pair_fwd=${m:-off} $(grep -h '"value"' gpurun_out/r3a_bench_${m:-off}.log | python -c "import sys,json; j=json.loads(sys.stdin.readline()); print(j['value'], j['ms_per_step'], j['final_losses']['loss'])")"
done; done
