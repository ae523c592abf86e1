#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
for rep in 1 2 3; do
timeout 300 python tools/td_first.py 2>/dev/null | tail -2
done 2>&1 | tee gpurun_out/r4_td_first2.txt
python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-prof 2>/dev/null | grep '"value"' | python -c "
import sys,json; j=json.loads(sys.stdin.readline()); print('bench', j['value'], j['ms_per_step']); e=j['extra']; print({k:v for k,v in e['dsl_iteration'].items() if k.startswith('ms_')}); print('td', e['train_detector']['imgs_per_s'], 'fp8', e['fp8_towers']['imgs_per_s'])" | tee -a gpurun_out/r4_td_first2.txt
timeout 900 python -m pytest tests/test_step_gpu.py tests/test_runner_gpu.py tests/test_resume_gpu.py -m gpu -q 2>&1 | tail -3
