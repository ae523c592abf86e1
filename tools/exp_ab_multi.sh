#!/bin/bash
# A/B/C... of several builds of the library in one box: tools/exp_ab_multi.sh name1 name2 ...  (dsl_amd/lib/libdsl_<name>.so; "tree" = the in-tree build)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for rep in 1 2; do for v in "$@"; do
  if [ $v = tree ]; then unset DSL_HIP_LIB; else export DSL_HIP_LIB=$R/dsl_amd/lib/libdsl_$v.so; fi
  python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-prof --no-dsl > gpurun_out/ab_${v}.log 2>&1
  echo "$v $(grep -h '"value"' gpurun_out/ab_${v}.log | python -c "import sys,json; j=json.loads(sys.stdin.readline()); print(j['value'], j['ms_per_step'])")"
done; done
