#!/bin/bash
# conv-epilogue GroupNorm statistics: kernel test, step tests, A/B in one box
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "groupnorm" 2>&1 | tail -15
timeout 1200 python -m pytest tests/test_step_gpu.py tests/test_sweep_gpu.py -m gpu -q -x 2>&1 | tail -8
bash tools/exp_ab_env.sh "DSL_GN_FUSE=0" "-" 2>&1 | tee gpurun_out/r04_gnfuse_ab.txt
