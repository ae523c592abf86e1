#!/bin/bash
# GroupNorm backward records in the data-gradient epilogue: kernel tests, step tests, A/B in one box
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_abi.py -m gpu -q -k "groupnorm or abi" 2>&1 | tail -15
timeout 1200 python -m pytest tests/test_step_gpu.py tests/test_sweep_gpu.py -m gpu -q -x 2>&1 | tail -8
bash tools/exp_ab_env.sh "DSL_GN_FUSE_BWD=0" "-" "DSL_GN_FUSE_BWD=0 DSL_GN_FUSE=0" 2>&1 | tee gpurun_out/r04_gnfuse_bwd_ab.txt
