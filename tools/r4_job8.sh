#!/bin/bash
# round 4, eighth GPU call: the fused layer1 bottleneck - kernel parity, whole-step goldens, standalone and in-step timing
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "bottleneck64" > gpurun_out/r4_job8_tests_a.log 2>&1
tail -12 gpurun_out/r4_job8_tests_a.log
timeout 900 python -m pytest tests/test_step_gpu.py tests/test_sweep_gpu.py -m gpu -q > gpurun_out/r4_job8_tests_b.log 2>&1
tail -6 gpurun_out/r4_job8_tests_b.log
timeout 300 python tools/bench_bneck.py 2 2>&1 | tail -3 | tee gpurun_out/r4_bench_bneck.txt
timeout 900 bash tools/exp_ab_env.sh "DSL_BNECK64=0" "-" 2>&1 | tee gpurun_out/r4_job8_ab.log
