#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
python tools/host_time.py --eager 2>&1 | tail -6 | tee gpurun_out/r04_host_time.txt
bash tools/exp_ab_env.sh "-" "DSL_PREFIX_DS_INLINE=1" "DSL_TAIL_SPREAD=2 DSL_PREFIX_DS_INLINE=1" "DSL_TAIL_SPREAD=1 DSL_PREFIX_DS_INLINE=1" 2>&1 | tee gpurun_out/r04_tail_spread2.txt
