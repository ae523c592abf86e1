mkdir -p gpurun_out
(timeout 600 python -m pytest tests/test_kernels_gpu.py -q -k "bottleneck_forward" 2>&1 | tail -8) > gpurun_out/r05_tests_e.log
tail -8 gpurun_out/r05_tests_e.log
python tools/bench_bneck.py 2>&1 | tail -2 | tee gpurun_out/r05_bneck_micro3.txt
(timeout 900 python -m pytest tests/test_step_gpu.py -q -x 2>&1 | tail -3) | tee -a gpurun_out/r05_tests_e.log
for rep in 1 2; do bash tools/exp_env.sh "DSL_TUNE=bneck_fwd=" "DSL_TUNE=bneck_fwd=2" "DSL_TUNE=bneck_fwd=3" "DSL_TUNE=bneck_fwd=23"; done > gpurun_out/r05_bneck_ab3.txt 2>&1
cat gpurun_out/r05_bneck_ab3.txt
bash tools/pmc_bneck.sh > gpurun_out/r05_bneck_pmc3.txt 2>&1; grep -A9 "bneck_fwd" gpurun_out/r05_bneck_pmc3.txt | head -50
