#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "half_stage or conv_forward or dgrad_transposed" 2>&1 | tail -6
timeout 600 python tools/bench_conv.py 2 1,10 2>&1 | head -3 | tee gpurun_out/r04_h4_probe.txt
timeout 600 python tools/bench_conv.py 3 1,10 2>&1 | head -3 | tee -a gpurun_out/r04_h4_probe.txt
bash tools/exp_ab_env.sh "-" "DSL_CONV_H4=1" 2>&1 | tee gpurun_out/r04_h4_ab.txt
