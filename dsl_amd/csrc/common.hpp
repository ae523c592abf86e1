// Shared device/host helpers for libdsl_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/dsl_hip.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

#define WAVE 64

void dsl_set_error(const char* fmt, ...);
int dsl_option(const char* name);        // library options (dsl_set_option, api.hip)
bool dsl_prof_active();
int dsl_prof_begin(int cls, double flops, hipStream_t st, double bytes = 0.0);
void dsl_prof_end(int id, hipStream_t st);
#define DSL_CHECK(cond, ...)          \
  do {                                \
    if (!(cond)) {                    \
      dsl_set_error(__VA_ARGS__);     \
      return -1;                      \
    }                                 \
  } while (0)
#define DSL_LAUNCH_CHECK(name)                                              \
  do {                                                                      \
    hipError_t e_ = hipGetLastError();                                      \
    if (e_ != hipSuccess) {                                                 \
      dsl_set_error("%s launch failed: %s", name, hipGetErrorString(e_));   \
      return -2;                                                            \
    }                                                                       \
  } while (0)

// ---- bf16 <-> fp32 (round to nearest even, as torch .bfloat16()) -------------------------------
__device__ __forceinline__ float bf2f(uint16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
// gfx950 converts in hardware (v_cvt_pk_bf16_f32: round to nearest even, NaN quieted) - one instruction per PAIR where the
// integer formulation (and / compare / add / shift / select per value) costs ~7 per value; in the staged convolution epilogue
// that was most of the ~120 VALU instructions per 16-byte store that made it issue-bound (tools/trace_conv.py).
typedef float dsl_f32x2_ __attribute__((ext_vector_type(2)));
typedef __bf16 dsl_bf16x2_ __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
  const dsl_f32x2_ v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, dsl_bf16x2_));
}
__device__ __forceinline__ uint16_t f2bf(float f) { return (uint16_t)(pack2bf(f, 0.f) & 0xffffu); }
__device__ __forceinline__ float bflo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bfhi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }

// ---- four floats -> four OCP e4m3 bytes (torch.float8_e4m3fn casts, saturating) -------------------
__device__ __forceinline__ uint32_t cvt4_fp8(float a, float b, float c, float d) {
  a = fminf(fmaxf(a, -448.f), 448.f);      // e4m3fn has no infinity: saturate instead of producing NaN
  b = fminf(fmaxf(b, -448.f), 448.f);
  c = fminf(fmaxf(c, -448.f), 448.f);
  d = fminf(fmaxf(d, -448.f), 448.f);
  int r = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
  r = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, r, true);
  return (uint32_t)r;
}

// ---- exact division of p < 2^20 by d < 2^20 via a 40-bit reciprocal ----------------------------
struct FastDiv {
  uint64_t m;
  uint32_t d;
};
static inline FastDiv make_fastdiv(uint32_t d) {
  FastDiv f;
  f.d = d ? d : 1;
  f.m = ((1ull << 40) + f.d - 1) / f.d;
  return f;
}
__device__ __forceinline__ uint32_t fdiv(uint32_t p, const FastDiv& f) {
  return (uint32_t)(((uint64_t)p * f.m) >> 40);
}

// ---- wave / block reductions ---------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float block_sum(float v, float* sh /* >= 16 floats */) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if (lane == 0) sh[w] = v;
  __syncthreads();
  float r = 0.f;
  for (int i = 0; i < nw; ++i) r += sh[i];
  return r;
}
