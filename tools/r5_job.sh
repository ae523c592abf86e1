mkdir -p gpurun_out
(timeout 1500 python -m pytest tests/test_rla_gpu.py -q --tb=short -k "train_step_vs_oracle" 2>&1 | tail -25) | tee gpurun_out/r05_tests_rla_side.log
