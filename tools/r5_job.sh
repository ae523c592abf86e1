mkdir -p gpurun_out
python bench.py > gpurun_out/r05_bench_full.log 2>gpurun_out/r05_bench_full.err; tail -c 1200 gpurun_out/r05_bench_full.log
