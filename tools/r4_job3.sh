#!/bin/bash
# round 4, third GPU call: deferred head update with the optimizer steps on the right hardware queues, early layer2 weight gradients
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_step_gpu.py tests/test_runner_gpu.py tests/test_ddp_gpu.py -m gpu -q > gpurun_out/r4_job3_tests.log 2>&1
tail -8 gpurun_out/r4_job3_tests.log
timeout 1800 bash tools/exp_ab_env.sh "DSL_DEFER_HEAD=0 DSL_L2_EARLY=0" "DSL_DEFER_HEAD=0" "-" "DSL_L2_EARLY=0" "DSL_DEFER_SLOTS=72" "DSL_DEFER_SLOTS=216" 2>&1 | tee gpurun_out/r4_job3_ab.log
timeout 600 bash tools/exp_prof.sh r4c > gpurun_out/r4c_prof_out.log 2>&1
tail -32 gpurun_out/r4c_prof_out.log
