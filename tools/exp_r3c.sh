#!/bin/bash
# tower weight gradients in two halves (early start) now that their launches are confined to few CUs
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for rep in 1 2; do for cfg in "0 96" "1 48" "1 64" "1 96"; do
  set -- $cfg
  DSL_TOWER_HALVES=$1 DSL_TOWER_SLOTS=$2 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-prof --no-dsl > gpurun_out/r3c_bench_h$1_s$2.log 2>&1
  echo "halves=$1 tower_slots=$2 $(grep -h '"value"' gpurun_out/r3c_bench_h$1_s$2.log | python -c "import sys,json; j=json.loads(sys.stdin.readline()); print(j['value'], j['ms_per_step'])")"
done; done
