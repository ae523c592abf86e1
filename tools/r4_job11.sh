#!/bin/bash
# round 4: the new data-parallel test, then the round's profiles on the default tree
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ddp_gpu.py -m gpu -q 2>&1 | tail -8 | tee gpurun_out/r4_job11_ddp.log
timeout 900 bash tools/exp_prof.sh r04 > gpurun_out/r04_prof_out.log 2>&1
tail -30 gpurun_out/r04_prof_out.log
timeout 1500 bash tools/exp_pmc.sh r04 > gpurun_out/r04_pmc_out.log 2>&1
tail -45 gpurun_out/r04_pmc_out.log
timeout 900 python tools/op_table.py > gpurun_out/r04_op_table.txt 2>gpurun_out/r04_op_table.err
head -12 gpurun_out/r04_op_table.txt
