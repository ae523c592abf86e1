#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 2400 python -m pytest tests/test_abi.py tests/test_fcos_loss_gpu.py tests/test_step_gpu.py tests/test_runner_gpu.py tests/test_ddp_gpu.py tests/test_kernels_gpu.py -m gpu -q -x 2>&1 | tail -8
bash tools/exp_ab_env.sh "DSL_LOG_TOTAL=0" "-" 2>&1 | tee gpurun_out/r04_logvec_ab.txt
