#!/bin/bash
# round 4, fifth GPU call: whole GPU suite on the fixed tree, clocks / power under the step, A/B of the opt-in schedules
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
(time timeout 1500 python -m pytest tests -m gpu -q -x) > gpurun_out/r4_job5_tests.log 2>&1
tail -6 gpurun_out/r4_job5_tests.log
# clocks and power while the step runs (is the chip at its power budget?)
python bench.py --steps 400 --warmup 5 --no-cpu-baseline --no-prof --no-dsl > gpurun_out/r4_job5_long.log 2>&1 &
BP=$!
sleep 25
for i in 1 2 3 4 5 6; do rocm-smi --showclocks --showpower --showuse 2>/dev/null | grep -E "sclk|mclk|fclk|Power|GPU use" | tr '\n' ' '; echo; sleep 0.3; done > gpurun_out/r4_job5_smi.txt 2>&1
wait $BP
rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' ' >> gpurun_out/r4_job5_smi.txt
cat gpurun_out/r4_job5_smi.txt
tail -c 600 gpurun_out/r4_job5_long.log | head -c 300; echo
timeout 1200 bash tools/exp_ab_env.sh "-" "DSL_L2_EARLY=1" "DSL_DEFER_HEAD=1 DSL_DEFER_SLOTS=72" "DSL_DEFER_HEAD=1 DSL_DEFER_SLOTS=72 DSL_L2_EARLY=1" 2>&1 | tee gpurun_out/r4_job5_ab.log
