#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
python bench.py --steps 3000 --warmup 20 --no-cpu-baseline --no-prof --no-dsl 2>&1 | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.readline())
print('3000 steps:', j['value'], j['ms_per_step'], j.get('final_losses'))
" | tee gpurun_out/r04_soak.txt
