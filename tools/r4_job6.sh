#!/bin/bash
# round 4, sixth GPU call: parity-class stride-2 data gradients + RLA tail budget (tests, A/B of the RLA DSL iteration), clocks under load
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "parity or full_size or pack" > gpurun_out/r4_job6_tests_a.log 2>&1
tail -4 gpurun_out/r4_job6_tests_a.log
timeout 1200 python -m pytest tests/test_rla_gpu.py tests/test_resume_gpu.py tests/test_sweep_gpu.py -m gpu -q > gpurun_out/r4_job6_tests_b.log 2>&1
tail -4 gpurun_out/r4_job6_tests_b.log
for rep in 1 2; do
for cfg in "DSL_S2_CLASSES=0 DSL_RLA_TAIL_SLOTS=128" "DSL_S2_CLASSES=1 DSL_RLA_TAIL_SLOTS=128" "DSL_S2_CLASSES=1 DSL_RLA_TAIL_SLOTS=192" "-"; do
  if [ "$cfg" = "-" ]; then e=""; else e="$cfg"; fi
  echo "[$cfg] $(env $e timeout 300 python tools/bench_dsl_variant.py 1 1 0 2>/dev/null | tail -1 | cut -c1-220)"
  echo "[$cfg async] $(env $e timeout 300 python tools/bench_dsl_variant.py 1 1 1 2>/dev/null | tail -1 | cut -c1-220)"
done; done 2>&1 | tee gpurun_out/r4_job6_rla_ab.log
timeout 300 python tools/clock_probe.py 8 > gpurun_out/r4_clock_probe.txt 2>&1
tail -30 gpurun_out/r4_clock_probe.txt
