mkdir -p gpurun_out
(timeout 1500 python -m pytest tests/test_step_gpu.py -q --tb=short -k "other_batch_sizes" 2>&1 | tail -30) | tee gpurun_out/r05_tests_shapes.log
