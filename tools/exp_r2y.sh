#!/bin/bash
# cross-stream events: device-scope release (default now) vs system-scope release
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for rep in 1 2 3; do for m in 1 0; do
  DSL_EVENT_SYS=$m python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-prof --no-dsl > gpurun_out/r2y_bench_${m}.log 2>&1
  echo "event_sys=$m $(grep -h '"value"' gpurun_out/r2y_bench_${m}.log | python -c "import sys,json; j=json.loads(sys.stdin.readline()); print(j['value'], j['ms_per_step'])")"
done; done
