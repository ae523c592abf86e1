#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_sweep_gpu.py -m gpu -q 2>&1 | tail -3
for rep in 1 2; do
timeout 300 python tools/td_first.py 2>/dev/null | tail -2
python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-prof --no-dsl 2>/dev/null | grep '"value"' | python -c "import sys,json; j=json.loads(sys.stdin.readline()); print('bench alone', j['value'], j['ms_per_step'])"
done 2>&1 | tee gpurun_out/r4_td_first.txt
