#!/bin/bash
# round 4, final tree: the whole GPU suite, the default bench, the RLA iteration's trace
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
(time timeout 1800 python -m pytest tests -m gpu -q --durations=10) > gpurun_out/r04_gpu_tests.log 2>&1
tail -5 gpurun_out/r04_gpu_tests.log
(time python bench.py) > gpurun_out/r04_bench_full.log 2>&1
tail -c 800 gpurun_out/r04_bench_full.log
bash tools/exp_prof_rla.sh r04_rla > gpurun_out/r04_rla_prof_out.log 2>&1
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
