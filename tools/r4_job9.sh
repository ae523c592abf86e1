#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "bottleneck64" 2>&1 | tail -3
for dbg in 0 1 2 4 8 16 3 7 31; do
  echo "[DSL_BNECK_DBG=$dbg] $(DSL_BNECK_DBG=$dbg timeout 300 python tools/bench_bneck.py 2 2>&1 | grep 'BNECK64=1')"
done 2>&1 | tee gpurun_out/r4_bneck_probe.txt
timeout 300 python tools/bench_bneck.py 2 2>&1 | tail -2
timeout 900 bash tools/exp_ab_env.sh "DSL_BNECK64=0" "-" 2>&1 | tee gpurun_out/r4_job9_ab.log
