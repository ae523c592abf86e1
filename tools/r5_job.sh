mkdir -p gpurun_out
bash tools/exp_env.sh "-" "-" 2>&1 | tee gpurun_out/r05_sanity.txt
(timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -8) | tee gpurun_out/r05_gpu_tests.log
