#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
bash tools/exp_ab_env.sh "-" "DSL_TOWER_SLOTS=56" "DSL_TOWER_SLOTS=96" "DSL_DEFER_HEAD=1" "DSL_BNECK64=1" "DSL_PREFIX_AT=3" 2>&1 | tee gpurun_out/r04_knobs_after_gn.txt
