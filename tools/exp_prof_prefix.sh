R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export TMPDIR=/tmp
for cfg in "at0:DSL_PREFIX_AT=0" "def:DSL_NOOP=1"; do
  TAG=s2_${cfg%%:*}; E=${cfg#*:}
  cd /tmp
  env $E rocprofv3 --kernel-trace -d $R/gpurun_out/prof_${TAG} -o ${TAG} -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-prof --no-dsl > $R/gpurun_out/${TAG}_rocprof.log 2>&1
  cd $R
  DB=$(find gpurun_out/prof_${TAG} -name '*_results.db' | head -1)
  python tools/step_sequence.py $DB > gpurun_out/${TAG}_sequence.txt
  python tools/stream_timeline.py $DB 5 > gpurun_out/${TAG}_timeline.txt
  rm -rf gpurun_out/prof_${TAG}
  head -8 gpurun_out/${TAG}_timeline.txt
  grep -n "stem_pool" gpurun_out/${TAG}_sequence.txt
done
