#!/bin/bash
# round 4, first GPU call: wgrad / step parity with the planner, A/B of the planner in one box, traced profile
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_step_gpu.py -m gpu -x -q -k "wgrad or step or reproducible" > gpurun_out/r4_job1_tests.log 2>&1
tail -3 gpurun_out/r4_job1_tests.log
timeout 900 bash tools/exp_ab_env.sh "DSL_WGRAD_PLAN=0" "-" 2>&1 | tee gpurun_out/r4_job1_ab.log
timeout 600 bash tools/exp_prof.sh r4a > gpurun_out/r4a_prof_out.log 2>&1
tail -40 gpurun_out/r4a_prof_out.log
