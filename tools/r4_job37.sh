#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_runner_gpu.py tests/test_resume_gpu.py -m gpu -q -x 2>&1 | tail -3
for rep in 1 2 3; do python tools/td_first.py 2>&1 | tail -1; done | tee gpurun_out/r04_td_first2.txt
