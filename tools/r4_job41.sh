#!/bin/bash
# the allocator-cache rule (fresh_allocations) against the second-model slowdown; the logger's single read; runner tests
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_runner_gpu.py tests/test_step_gpu.py tests/test_resume_gpu.py -m gpu -q -x 2>&1 | tail -3
for rep in 1 2 3 4 5 6 7 8; do
  echo "[fresh_allocations, models dropped in between] $(python tools/second_model_probe.py 0 2>&1 | tail -1 | cut -c1-120)"
done | tee gpurun_out/r04_second_model_rule.txt
for rep in 1 2 3; do python tools/td_first.py 2>&1 | tail -2 | tr '\n' ' '; echo; done | tee gpurun_out/r04_td_first2.txt
