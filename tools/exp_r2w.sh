#!/bin/bash
# end of the backward pass: which of layer2's weight-gradient groups the caller's stream takes (3 = conv3 x4, 2 = conv2 x4, 1 = conv1 x3)
# and the workgroup budget of the last segment's launches
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for tm in 1 2 12 13; do for ts in 0 128 256; do
  DSL_TAIL_MAIN=$tm DSL_TAIL_SLOTS=$ts python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-prof --no-dsl > gpurun_out/r2w_bench_${tm}_${ts}.log 2>&1
  echo "tail_main=$tm tail_slots=$ts $(grep -h '"value"' gpurun_out/r2w_bench_${tm}_${ts}.log | python -c "import sys,json; j=json.loads(sys.stdin.readline()); print(j['value'], j['ms_per_step'])")"
done; done
