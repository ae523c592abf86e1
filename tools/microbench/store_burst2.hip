// What makes the conv epilogue's stores slow?  Variants of "every workgroup writes 96 KB": (a) 12 plain stores per thread,
// (b) the epilogue's shape: 3 slabs x {barrier, 8 ds_write_b128, barrier, 4 x (2 ds_read_b128 + store)}, (c) = (b) with the values
// converted fp32 -> bf16 as the epilogue does, (d) = (b) but the 4 stores of a slab issued after all 8 reads.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
template <int MODE>
__global__ __launch_bounds__(512) void epi(u32x4* out, const float* in) {
  extern __shared__ unsigned char smem[];
  constexpr int ROWB = 256 * 4 + 16;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float acc[96];
#pragma unroll
  for (int i = 0; i < 96; ++i) acc[i] = in[(tid + i * 7) & 1023];
  u32x4* base = out + (size_t)blockIdx.x * 12 * 512;
  if (MODE == 0) {
#pragma unroll
    for (int r = 0; r < 12; ++r) {
      u32x4 v = {__float_as_uint(acc[r * 8]), __float_as_uint(acc[r * 8 + 1]), __float_as_uint(acc[r * 8 + 2]), __float_as_uint(acc[r * 8 + 3])};
      base[r * 512 + tid] = v;
    }
    return;
  }
  const int frow = lane & 31, fhalf = lane >> 5, wave_co = wave >> 1, wave_px = wave & 1;
#pragma unroll
  for (int pt = 0; pt < 3; ++pt) {
    lds_barrier();
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int col = wave_co * 64 + ct * 32 + 8 * g + 4 * fhalf;
        f32x4 o = {acc[pt * 32 + ct * 16 + 4 * g], acc[pt * 32 + ct * 16 + 4 * g + 1], acc[pt * 32 + ct * 16 + 4 * g + 2], acc[pt * 32 + ct * 16 + 4 * g + 3]};
        *reinterpret_cast<f32x4*>(smem + (wave_px * 32 + frow) * ROWB + col * 4) = o;
      }
    lds_barrier();
    const int pl0 = tid / 32, cg = tid % 32;
    f32x4 lo[4], hi[4];
    if (MODE == 3) {
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        lo[n] = *reinterpret_cast<const f32x4*>(smem + (pl0 + 16 * n) * ROWB + cg * 32);
        hi[n] = *reinterpret_cast<const f32x4*>(smem + (pl0 + 16 * n) * ROWB + cg * 32 + 16);
      }
    }
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      const int pl = pl0 + 16 * n;
      if (MODE != 3) {
        lo[n] = *reinterpret_cast<const f32x4*>(smem + pl * ROWB + cg * 32);
        hi[n] = *reinterpret_cast<const f32x4*>(smem + pl * ROWB + cg * 32 + 16);
      }
      u32x4 v;
      if (MODE == 2) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          v[e] = (__float_as_uint(fmaxf(lo[n][2 * e], 0.f)) >> 16) | (__float_as_uint(fmaxf(lo[n][2 * e + 1], 0.f)) & 0xffff0000u);
          v[2 + e] = (__float_as_uint(fmaxf(hi[n][2 * e], 0.f)) >> 16) | (__float_as_uint(fmaxf(hi[n][2 * e + 1], 0.f)) & 0xffff0000u);
        }
      } else {
        v = u32x4{__float_as_uint(lo[n][0]), __float_as_uint(lo[n][1]), __float_as_uint(hi[n][0]), __float_as_uint(hi[n][1])};
      }
      // pixel row (pl >> 5) * 96 + pt * 32 + (pl & 31) of the 192-pixel tile, 32 x 16 B per row
      const int row = (pl >> 5) * 96 + pt * 32 + (pl & 31);
      base[row * 32 + cg] = v;
    }
  }
}
int main() {
  const int G = 234;
  u32x4* d; float* in;
  CK(hipMalloc(&d, (size_t)G * 12 * 512 * 16));
  CK(hipMalloc(&in, 4096));
  CK(hipMemset(in, 0, 4096));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const size_t lds = 64 * (256 * 4 + 16);
  auto run = [&](int mode) {
    switch (mode) {
      case 0: hipLaunchKernelGGL(epi<0>, dim3(G), dim3(512), lds, 0, d, in); break;
      case 1: hipLaunchKernelGGL(epi<1>, dim3(G), dim3(512), lds, 0, d, in); break;
      case 2: hipLaunchKernelGGL(epi<2>, dim3(G), dim3(512), lds, 0, d, in); break;
      default: hipLaunchKernelGGL(epi<3>, dim3(G), dim3(512), lds, 0, d, in); break;
    }
  };
  for (int mode = 0; mode < 4; ++mode) {
    float best = 1e9f;
    for (int rep = 0; rep < 6; ++rep) {
      CK(hipEventRecord(e0, 0));
      run(mode);
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (ms < best) best = ms;
    }
    printf("mode %d: %.2f us per launch (incl. ~6 us of launch + event overhead)\n", mode, best * 1e3);
  }
  return 0;
}
