#!/bin/bash
# Builds dsl_amd/lib/libdsl_hip_ablate.so: the same library with the conv kernels' ablation knobs compiled in
# (-DDSL_ABLATE_BUILD; DSL_ABLATE=<bits> at run time: 1 no pixel DMA, 2 no weight DMA, 4 no MFMA).
# Use: DSL_HIP_LIB=$PWD/dsl_amd/lib/libdsl_hip_ablate.so DSL_ABLATE=4 python tools/bench_conv.py
set -e
cd "$(dirname "$0")/.."
python -c "from dsl_amd.build import build_lib; build_lib(verbose=False)" 2>/dev/null
mkdir -p dsl_amd/lib/ablate
for f in conv wgrad bneck; do      # the three sources that carry ablation knobs (bneck: DSL_BNECK_DBG = 1 / 2, phase cut-offs)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -DDSL_ABLATE_BUILD -Iinclude -c dsl_amd/csrc/$f.hip -o dsl_amd/lib/ablate/$f.o 2>/dev/null &
done
wait
objs=$(ls dsl_amd/lib/*.o | grep -v "/conv.o\|/wgrad.o\|/bneck.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs dsl_amd/lib/ablate/conv.o dsl_amd/lib/ablate/wgrad.o dsl_amd/lib/ablate/bneck.o -ldl -o dsl_amd/lib/libdsl_hip_ablate.so
echo built dsl_amd/lib/libdsl_hip_ablate.so
