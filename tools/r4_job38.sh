#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
for rep in 1 2 3; do for m in 0 1 2; do python tools/second_model_probe.py $m 2>&1 | tail -1; done; done | tee gpurun_out/r04_second_model.txt
