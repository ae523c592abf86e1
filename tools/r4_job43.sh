#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
for rep in 1 2 3 4 5 6 7 8 9 10 11 12; do
  python tools/second_model_probe.py 0 4 2>&1 | tail -1 | cut -c1-200
done | tee gpurun_out/r04_second_model_windows.txt
