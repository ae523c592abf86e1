#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "groupnorm" 2>&1 | tail -5
python tools/bench_gn_fuse.py 2>&1 | tee gpurun_out/r04_gn_fuse_probe.txt
bash tools/exp_ab_env.sh "DSL_GN_FUSE_BWD=0" "-" 2>&1 | tee gpurun_out/r04_gnfuse_bwd_ab2.txt
