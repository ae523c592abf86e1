mkdir -p gpurun_out
(timeout 1500 python -m pytest tests/test_step_gpu.py -q --tb=short -k "vs_reference_and_oracle" 2>&1 | tail -25) | tee gpurun_out/r05_tests_knobs.log
