#!/bin/bash
# Builds dsl_amd/lib/libdsl_hip_trace.so: the library with s_memtime stamps in conv_pipe_kernel's K loop (-DDSL_TRACE_BUILD;
# DSL_TRACE_WG=<tile index> at run time).  Use: python tools/trace_conv.py
set -e
cd "$(dirname "$0")/.."
python -c "from dsl_amd.build import build_lib; build_lib(verbose=False)" 2>/dev/null
mkdir -p dsl_amd/lib/trace
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -DDSL_TRACE_BUILD -Iinclude -c dsl_amd/csrc/conv.hip -o dsl_amd/lib/trace/conv.o 2>/dev/null
objs=$(ls dsl_amd/lib/*.o | grep -v "/conv.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs dsl_amd/lib/trace/conv.o -o dsl_amd/lib/libdsl_hip_trace.so
echo built dsl_amd/lib/libdsl_hip_trace.so
