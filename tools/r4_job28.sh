#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "groupnorm" 2>&1 | tail -3
for rep in 1 2 3; do
for e in "DSL_GN_FUSE_BWD=0" "DSL_X=1"; do
  echo "[$e] $(env $e python tools/bench_dsl_variant.py 0 0 0 2>&1 | tail -1 | cut -c1-110)"
  echo "[$e rla] $(env $e python tools/bench_dsl_variant.py 0 1 0 2>&1 | tail -1 | cut -c1-130)"
done; done 2>&1 | tee gpurun_out/r04_gnfuse_dsl_ab2.txt
bash tools/exp_ab_env.sh "DSL_GN_FUSE_BWD=0" "-" 2>&1 | tail -6
