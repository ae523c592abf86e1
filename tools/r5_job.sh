mkdir -p gpurun_out
(timeout 600 python -m pytest tests/test_step_gpu.py -q -x -k "fused_bottleneck" 2>&1 | tail -6) | tee gpurun_out/r05_tests_k.log
bash tools/exp_prof.sh r05 > gpurun_out/r05_prof.log 2>&1; tail -32 gpurun_out/r05_prof.log
bash tools/exp_pmc.sh r05 > gpurun_out/r05_pmc.log 2>&1; tail -30 gpurun_out/r05_pmc.log
python bench.py > gpurun_out/r05_bench_full.log 2>gpurun_out/r05_bench_full.err; tail -c 3000 gpurun_out/r05_bench_full.log
