#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
DSL_TAIL_SPREAD=2 timeout 1200 python -m pytest tests/test_step_gpu.py -m gpu -q -x -k "reproducible or vs_reference or pipelined or momentum or eager" 2>&1 | tail -3
bash tools/exp_ab_env.sh "-" "DSL_TAIL_SPREAD=1" "DSL_TAIL_SPREAD=2" 2>&1 | tee gpurun_out/r04_tail_spread.txt
