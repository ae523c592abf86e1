mkdir -p gpurun_out
(timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15) > gpurun_out/r05_gpu_tests.log
tail -6 gpurun_out/r05_gpu_tests.log
