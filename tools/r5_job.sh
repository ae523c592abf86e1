mkdir -p gpurun_out
(timeout 600 python -m pytest tests/test_kernels_gpu.py -q -k "addend_epilogue or bottleneck_forward or probe" 2>&1 | tail -15) > gpurun_out/r05_tests_d.log
(timeout 900 python -m pytest tests/test_step_gpu.py -q -x 2>&1 | tail -4) >> gpurun_out/r05_tests_d.log
tail -25 gpurun_out/r05_tests_d.log
python tools/bench_bneck.py 2>&1 | tail -4 | tee gpurun_out/r05_bneck_micro.txt
for rep in 1 2; do bash tools/exp_env.sh "DSL_TUNE=bneck_fwd=" "DSL_TUNE=bneck_fwd=2" "DSL_TUNE=bneck_fwd=3" "DSL_TUNE=bneck_fwd=23" "DSL_TUNE=bneck_fwd=23,img_split="; done > gpurun_out/r05_bneck_ab2.txt 2>&1
cat gpurun_out/r05_bneck_ab2.txt
bash tools/stream_layout_probe.sh > gpurun_out/r05_stream_probe2.txt 2>&1
cat gpurun_out/r05_stream_probe2.txt
