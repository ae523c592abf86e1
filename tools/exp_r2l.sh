#!/bin/bash
# data-gradient weight packs on the side stream x both towers' weight gradients as one launch of 8
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for ps in 0 1; do for g8 in 0 1; do
  DSL_PACK_SIDE=$ps DSL_TOWER_GROUP8=$g8 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-prof --no-dsl > gpurun_out/r2l_bench_p${ps}_g${g8}.log 2>&1
  echo "pack_side=$ps tower8=$g8 $(grep -h '"value"' gpurun_out/r2l_bench_p${ps}_g${g8}.log | python -c "import sys,json; j=json.loads(sys.stdin.readline()); print(j['value'], j['ms_per_step'])")"
done; done
