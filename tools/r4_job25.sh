#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 2400 python -m pytest tests/test_kernels_gpu.py tests/test_runner_gpu.py tests/test_rla_gpu.py -m gpu -q -x -k "stem or half or semi or dsl_iteration or rla" 2>&1 | tail -8
for rep in 1 2; do
for e in "DSL_HALF_IN_STEM=0" "DSL_X=1"; do
  echo "[$e] $(env $e python tools/bench_dsl_variant.py 0 0 0 2>&1 | tail -1 | cut -c1-120)"
  echo "[$e rla refresh async] $(env $e python tools/bench_dsl_variant.py 1 1 1 2>&1 | tail -1 | cut -c1-200)"
done; done 2>&1 | tee gpurun_out/r04_half_in_stem_ab.txt
