#!/bin/bash
# bench + rocprofv3 kernel trace of the overlapped step; TAG = output prefix
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export TMPDIR=/tmp
TAG=${1:-r2x}
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-prof --no-dsl > gpurun_out/${TAG}_bench.log 2>&1
cd /tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_${TAG} -o ${TAG} -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-prof --no-dsl > $R/gpurun_out/${TAG}_rocprof.log 2>&1
cd $R
DB=$(find gpurun_out/prof_${TAG} -name '*_results.db' | head -1)
python tools/rocprof_summary.py $DB > gpurun_out/${TAG}_kernel_stats.txt
python tools/stream_timeline.py $DB 5 > gpurun_out/${TAG}_timeline.txt
python tools/step_sequence.py $DB > gpurun_out/${TAG}_sequence.txt
rm -rf gpurun_out/prof_${TAG}
grep -h '"value"' gpurun_out/${TAG}_bench.log | python -c "
import sys, json
for l in sys.stdin:
    j = json.loads(l); print(j['value'], j['ms_per_step'])
"
head -30 gpurun_out/${TAG}_timeline.txt
