#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_step_gpu.py tests/test_sweep_gpu.py -m gpu -q -x 2>&1 | tail -3
for rep in 1 2 3 4 5 6; do
  echo "[arena 0] $(DSL_PLAN_ARENA=0 python tools/second_model_probe.py 0 2>&1 | tail -1 | cut -c1-120)"
  echo "[arena 1] $(DSL_PLAN_ARENA=1 python tools/second_model_probe.py 0 2>&1 | tail -1 | cut -c1-120)"
done | tee gpurun_out/r04_second_model_arena.txt
