#!/bin/bash
# Stream-layout robustness: the default bench loop with the library's streams in each priority class (lib.stream_probe 0 / 1) and
# with foreign streams of the process created + used before / after the model is built (bench.py --foreign-streams).  One box, two rounds.
cd ${GRAFT_REPO_ROOT:-/root/repo}
run() { env $1 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-prof --no-dsl $2 2>>gpurun_out/stream_probe_stderr.log | grep '"value"' | python -c "import sys,json; j=json.loads(sys.stdin.readline()); print(j['value'], j['ms_per_step'])"; }
for rep in 1 2; do
  for prio in 0 1; do
    for f in none before after before,after; do
      echo "[stream_probe=$prio foreign=$f] $(run "DSL_TUNE=lib.stream_probe=$prio" "--foreign-streams $f")"
    done
  done
done
