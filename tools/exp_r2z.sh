#!/bin/bash
# workgroup budget of the towers' weight-gradient group (it runs beside the FPN / layer4 / layer3 backward chains)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for rep in 1 2; do for s in 48 64 80 96 128 160 192; do
  DSL_TOWER_SLOTS=$s python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-prof --no-dsl > gpurun_out/r2z_bench_${s}.log 2>&1
  echo "tower_slots=$s $(grep -h '"value"' gpurun_out/r2z_bench_${s}.log | python -c "import sys,json; j=json.loads(sys.stdin.readline()); print(j['value'], j['ms_per_step'])")"
done; done
