#!/bin/bash
# round 4, fourth GPU call: bisect the reproducibility failures of job 3 (early layer2 weight gradients / planner / deferral)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
T="tests/test_step_gpu.py::test_training_step_is_bit_reproducible tests/test_step_gpu.py::test_eager_backward_same_gradients tests/test_step_gpu.py::test_pipelined_prefix_equals_inline_forward"
for cfg in "-" "DSL_L2_EARLY=0" "DSL_L2_EARLY=0 DSL_WGRAD_PLAN=0" "DSL_WGRAD_PLAN=0" "DSL_L2_EARLY=0 DSL_DEFER_HEAD=0"; do
  if [ "$cfg" = "-" ]; then e=""; else e="$cfg"; fi
  echo "=== [$cfg]"
  env $e timeout 600 python -m pytest $T -m gpu -q 2>&1 | grep -E "passed|failed|^FAILED" 
done 2>&1 | tee gpurun_out/r4_job4_bisect.log
timeout 1200 bash tools/exp_ab_env.sh "DSL_DEFER_HEAD=0 DSL_L2_EARLY=0" "DSL_L2_EARLY=0" "DSL_L2_EARLY=0 DSL_DEFER_SLOTS=72" "DSL_L2_EARLY=0 DSL_DEFER_SLOTS=104" 2>&1 | tee gpurun_out/r4_job4_ab.log
