#!/bin/bash
# round 4, final tree: the whole GPU suite, the default bench, then the round's profiles (kernel trace, PMC passes, op table, RLA trace)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
(time timeout 1800 python -m pytest tests -m gpu -q --durations=10) > gpurun_out/r04_gpu_tests.log 2>&1
tail -5 gpurun_out/r04_gpu_tests.log
(time python bench.py) > gpurun_out/r04_bench_full.log 2>&1
tail -c 1500 gpurun_out/r04_bench_full.log
timeout 900 bash tools/exp_prof.sh r04 > gpurun_out/r04_prof_out.log 2>&1
tail -30 gpurun_out/r04_prof_out.log
timeout 1500 bash tools/exp_pmc.sh r04 > gpurun_out/r04_pmc_out.log 2>&1
tail -45 gpurun_out/r04_pmc_out.log
timeout 900 python tools/op_table.py > gpurun_out/r04_op_table.txt 2>gpurun_out/r04_op_table.err
head -12 gpurun_out/r04_op_table.txt
bash tools/exp_prof_rla.sh r04_rla > gpurun_out/r04_rla_prof_out.log 2>&1
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
