#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "conv_forward or dgrad_transposed or groupnorm" 2>&1 | tail -4
timeout 600 python tools/bench_conv.py 2 0,1,9 2>&1 | tee gpurun_out/r04_conv256_probe.txt
timeout 600 python tools/bench_conv.py 3 0,1,9 2>&1 | head -3 | tee -a gpurun_out/r04_conv256_probe.txt
bash tools/exp_ab_env.sh "-" "DSL_CONV_256=1" 2>&1 | tee gpurun_out/r04_conv256_ab.txt
