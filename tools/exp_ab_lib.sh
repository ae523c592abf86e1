#!/bin/bash
# A/B of two builds of the library in one box: dsl_amd/lib/libdsl_base.so (a copy of an earlier build) vs the in-tree build
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for rep in 1 2 3; do for v in base new; do
  if [ $v = base ]; then export DSL_HIP_LIB=$R/dsl_amd/lib/libdsl_base.so; else unset DSL_HIP_LIB; fi
  python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-prof --no-dsl > gpurun_out/ab_${v}.log 2>&1
  echo "$v $(grep -h '"value"' gpurun_out/ab_${v}.log | python -c "import sys,json; j=json.loads(sys.stdin.readline()); print(j['value'], j['ms_per_step'])")"
done; done
