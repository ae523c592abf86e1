#!/bin/bash
# experiment: step time with and without the side stream (DSL_SIDE), kernel stats of the single-stream run
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export TMPDIR=/tmp
TAG=${1:-r2a}
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-prof > gpurun_out/${TAG}_side1.log 2>&1
DSL_SIDE=0 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-prof > gpurun_out/${TAG}_side0.log 2>&1
cd /tmp
DSL_SIDE=0 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_${TAG} -o ${TAG} -- python $R/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-prof > $R/gpurun_out/${TAG}_rocprof.log 2>&1
cd $R
DB=$(find gpurun_out/prof_${TAG} -name '*_results.db' | head -1)
python tools/rocprof_summary.py $DB > gpurun_out/${TAG}_side0_kernel_stats.txt
python tools/stream_timeline.py $DB 5 > gpurun_out/${TAG}_side0_timeline.txt
rm -rf gpurun_out/prof_${TAG}
grep -h '"value"' gpurun_out/${TAG}_side1.log gpurun_out/${TAG}_side0.log | python -c "
import sys, json
for l in sys.stdin:
    j = json.loads(l); print(j['value'], j['ms_per_step'])
"
