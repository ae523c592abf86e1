#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_trajectory_gpu.py tests/test_step_gpu.py tests/test_ddp_gpu.py -x -q -m gpu -s 2>&1 | grep -v Warning | tail -60 > gpurun_out/r2i_tests.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r2i_bench.log 2>&1
cat gpurun_out/r2i_tests.log
tail -3 gpurun_out/r2i_bench.log
