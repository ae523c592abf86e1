#!/bin/bash
# the N = 3 DSL iteration with / without the GroupNorm records from the convolutions' epilogues; then the RLA profile of the round
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
for rep in 1 2; do
for e in "DSL_GN_FUSE=0 DSL_GN_FUSE_BWD=0" "DSL_GN_FUSE_BWD=0" "DSL_X=1"; do
  echo "[$e] $(env $e python tools/bench_dsl_variant.py 0 0 0 2>&1 | tail -1 | cut -c1-300)"
  echo "[$e rla] $(env $e python tools/bench_dsl_variant.py 0 1 0 2>&1 | tail -1 | cut -c1-300)"
done; done 2>&1 | tee gpurun_out/r04_gnfuse_dsl_ab.txt
bash tools/exp_prof_rla.sh r04_rla
