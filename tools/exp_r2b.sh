#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "conv_forward or dgrad or groupnorm" 2>&1 | tail -15 > gpurun_out/r2b_tests.log
python tools/bench_gn.py > gpurun_out/r2b_bench_gn.log 2>&1
python tools/bench_conv.py 2 0,1,2,3,4,5,6,7,8 > gpurun_out/r2b_bench_conv.log 2>&1
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-prof > gpurun_out/r2b_bench.log 2>&1
cat gpurun_out/r2b_tests.log gpurun_out/r2b_bench_gn.log
grep -h '"value"' gpurun_out/r2b_bench.log | python -c "
import sys, json
for l in sys.stdin:
    j = json.loads(l); print(j['value'], j['ms_per_step'], j['final_losses'])
"
