#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
K="pipelined_prefix or reproducible or deferred or eager or enqueue or async_sweep"
for cfg in "DSL_ROLE_STREAMS=1" "DSL_ROLE_STREAMS=0"; do
for i in 1 2 3 4 5 6; do
  echo "[$cfg] $(env $cfg timeout 600 python -m pytest tests/test_step_gpu.py tests/test_runner_gpu.py -m gpu -q -k "$K" 2>&1 | grep -E "^FAILED|passed|failed" | tr '\n' ' ' | cut -c1-400)"
done; done
