#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
bash tools/exp_ab_env.sh "-" "DSL_CONV_256=2" "DSL_CONV_256=2 DSL_GN_FUSE_BWD=0" 2>&1 | tee gpurun_out/r04_conv256_ab2.txt
for rep in 1 2; do
for e in "DSL_X=1" "DSL_CONV_256=1" "DSL_CONV_256=2"; do
  echo "[$e] $(env $e python tools/bench_dsl_variant.py 0 0 0 2>&1 | tail -1 | cut -c1-110)"
done; done 2>&1 | tee -a gpurun_out/r04_conv256_ab2.txt
