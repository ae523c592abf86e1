#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
bash tools/exp_ab_env.sh "-" "DSL_BNECK64=1" "DSL_BNECK64=1 DSL_BNECK_GRID=192" "DSL_BNECK64=1 DSL_BNECK_GRID=128" "DSL_BNECK64=1 DSL_BNECK_GRID=96" "DSL_BNECK64=1 DSL_BNECK_GRID=64" 2>&1 | tee gpurun_out/r04_bneck_grid.txt
