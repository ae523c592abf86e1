#!/bin/bash
# Where should the communication stream sit?    One fresh process per placement (the streams are
# created once per process): option comm_queue = 0 (as the runtime deals it), 1 weight gradients', 2 second chain's, 3 frozen
# prefix's, 4 the caller's hardware queue; each runs bench.comm_proxy_timing (none / lib / torch carriers alternated).
# usage: tools/comm_queue_sweep.sh [out-file]
OUT=${1:-gpurun_out/comm_queue_sweep.txt}
mkdir -p "$(dirname "$OUT")"
: > "$OUT"
for q in 1 0 3 2 4 1; do
  echo "== comm_queue=$q" >> "$OUT"
  DSL_TUNE="lib.comm_queue=$q" timeout 600 python - >> "$OUT" 2>&1 <<'PY'
import json, sys, os
sys.path.insert(0, os.getcwd())
import torch
import bench
from dsl_amd import detectors
b = bench.synth_batch(0, 2)
r = bench.comm_proxy_timing(b, steps=30, warm=8, rounds=3)
print(json.dumps({k: r[k] for k in ('ms_per_step', 'cost_frac', 'comm_stream_queue', 'runs')}))
print(json.dumps(r['proxy']))
PY
done
cat "$OUT"
