#!/bin/bash
# rocprofv3 kernel trace of the semi-supervised iteration with the RLA_ResNet backbone (BASELINE.json configs[2] with the DSL
# config's own backbone): kernel stats, per-queue timeline, kernel sequence of one iteration.  TAG = output prefix
R=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
TAG=${1:-r03_rla}
cd /tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_${TAG} -o ${TAG} -- python $R/tools/bench_dsl_variant.py 1 1 0 > $R/gpurun_out/${TAG}_rocprof.log 2>&1
cd $R
DB=$(find gpurun_out/prof_${TAG} -name '*_results.db' | head -1)
python tools/rocprof_summary.py $DB > gpurun_out/${TAG}_kernel_stats.txt
python tools/stream_timeline.py $DB 5 loss_kernel > gpurun_out/${TAG}_timeline.txt
python tools/step_sequence.py $DB loss_kernel > gpurun_out/${TAG}_sequence.txt
rm -rf gpurun_out/prof_${TAG}
head -34 gpurun_out/${TAG}_timeline.txt
