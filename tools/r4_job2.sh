#!/bin/bash
# round 4, second GPU call: deferred head update - parity tests, A/B in one box, traced profile
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_step_gpu.py tests/test_ddp_gpu.py -m gpu -x -q > gpurun_out/r4_job2_tests.log 2>&1
tail -5 gpurun_out/r4_job2_tests.log
timeout 1500 bash tools/exp_ab_env.sh "DSL_DEFER_HEAD=0" "-" "DSL_DEFER_SLOTS=72" "DSL_DEFER_SLOTS=216" 2>&1 | tee gpurun_out/r4_job2_ab.log
timeout 600 bash tools/exp_prof.sh r4b > gpurun_out/r4b_prof_out.log 2>&1
tail -32 gpurun_out/r4b_prof_out.log
