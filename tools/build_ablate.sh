#!/bin/bash
# Builds dsl_amd/lib/libdsl_hip_ablate.so: the same library with the conv kernels' ablation knobs compiled in
# (-DDSL_ABLATE_BUILD; DSL_ABLATE=<bits> at run time: 1 no pixel DMA, 2 no weight DMA, 4 no MFMA).
# Use: DSL_HIP_LIB=$PWD/dsl_amd/lib/libdsl_hip_ablate.so DSL_ABLATE=4 python tools/bench_conv.py
set -e
cd "$(dirname "$0")/.."
python -c "from dsl_amd.build import build_lib; build_lib(verbose=False)" 2>/dev/null
mkdir -p dsl_amd/lib/ablate
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -DDSL_ABLATE_BUILD -Iinclude -c dsl_amd/csrc/conv.hip -o dsl_amd/lib/ablate/conv.o 2>/dev/null
objs=$(ls dsl_amd/lib/*.o | grep -v "/conv.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs dsl_amd/lib/ablate/conv.o -o dsl_amd/lib/libdsl_hip_ablate.so
echo built dsl_amd/lib/libdsl_hip_ablate.so
