// How fast can every CU write its output tile at once?  G workgroups x 512 threads, each thread R coalesced 16-byte stores (a
// workgroup writes R x 8 KB contiguous = the conv epilogue's pattern: 192 px x 256 ch bf16 = 96 KB for R = 12).
// hipcc --offload-arch=gfx950 -O2 tools/microbench/store_burst.hip -o tools/microbench/store_burst
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int R, int NT>
__global__ __launch_bounds__(512) void burst(u32x4* out, int spin) {
  u32x4 v = {threadIdx.x, blockIdx.x, 3u, 4u};
  float f = (float)threadIdx.x;
  for (int i = 0; i < spin; ++i) f = f * 1.0001f + 0.5f;        // a "K loop" in front, so that the stores of all workgroups coincide
  v[0] += (unsigned)f;
  u32x4* base = out + (size_t)blockIdx.x * R * 512 + threadIdx.x;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    if (NT) __builtin_nontemporal_store(v, base + r * 512);
    else base[r * 512] = v;
  }
}
int main() {
  const int G = 234, R = 12;
  u32x4* d;
  CK(hipMalloc(&d, (size_t)G * R * 512 * 16 * 4));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int nt = 0; nt < 2; ++nt)
    for (int spin : {0, 20000}) {
      float best[2] = {1e9f, 1e9f};
      for (int with = 0; with < 2; ++with)
        for (int rep = 0; rep < 5; ++rep) {
          CK(hipEventRecord(e0, 0));
          if (with) { if (nt) hipLaunchKernelGGL((burst<R, 1>), dim3(G), dim3(512), 0, 0, d, spin); else hipLaunchKernelGGL((burst<R, 0>), dim3(G), dim3(512), 0, 0, d, spin); }
          else { if (nt) hipLaunchKernelGGL((burst<0, 1>), dim3(G), dim3(512), 0, 0, d, spin); else hipLaunchKernelGGL((burst<0, 0>), dim3(G), dim3(512), 0, 0, d, spin); }
          CK(hipEventRecord(e1, 0));
          CK(hipEventSynchronize(e1));
          float ms;
          CK(hipEventElapsedTime(&ms, e0, e1));
          if (ms < best[with]) best[with] = ms;
        }
      const double mb = (double)G * R * 512 * 16 / 1e6;
      printf("nontemporal %d spin %5d: no stores %.2f us, with %d x 16 B stores per thread %.2f us -> +%.2f us for %.1f MB = %.2f TB/s\n", nt, spin,
             best[0] * 1e3, R, best[1] * 1e3, (best[1] - best[0]) * 1e3, mb, mb / ((best[1] - best[0]) * 1e3) / 1e6 * 1e6 / 1e6);
    }
  return 0;
}
