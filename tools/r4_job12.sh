#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
(time timeout 1800 python -m pytest tests -m gpu -q --durations=15) > gpurun_out/r04_gpu_tests.log 2>&1
tail -30 gpurun_out/r04_gpu_tests.log
(time python bench.py) > gpurun_out/r04_bench_full.log 2>&1
tail -c 6000 gpurun_out/r04_bench_full.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
