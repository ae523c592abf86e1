#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
for v in 0 1 2 3 4; do
  echo "[DSL_CONV_H4=$v] $(DSL_CONV_H4=$v timeout 600 python tools/bench_conv.py 2 1 2>&1 | grep -E 'head tower|fpn 3x3' | cut -c1-120 | tr '\n' '|')"
done 2>&1 | tee gpurun_out/r04_h4_variants.txt
DSL_CONV_H4=4 timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "half_stage" 2>&1 | tail -2
bash tools/exp_ab_env.sh "-" "DSL_CONV_H4=1" "DSL_CONV_H4=2" "DSL_CONV_H4=3" "DSL_CONV_H4=4" 2>&1 | tee -a gpurun_out/r04_h4_variants.txt
