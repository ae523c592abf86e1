#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
K="pipelined_prefix or reproducible or deferred or eager or enqueue or async_sweep or momentum"
for i in 1 2 3 4 5 6; do
  echo "[fixed] $(timeout 600 python -m pytest tests/test_step_gpu.py tests/test_runner_gpu.py -m gpu -q -k "$K" 2>&1 | grep -E "^FAILED|passed|failed" | tr '\n' ' ' | cut -c1-400)"
done
