// Cost of cross-stream ordering primitives on the recording / waiting stream's timeline (MI355X, ROCm 7.2):
//   A  100 small kernels back to back on one stream
//   B  the same, with  hipEventRecord(main) + hipStreamWaitEvent(side)  after every kernel           ("FORK")
//   C  the same, with  hipEventRecord(side) + hipStreamWaitEvent(main)  after every kernel           ("JOIN", side idle)
//   D  FORK + a kernel on the side stream + JOIN after every kernel
//   E  as B but hipStreamWriteValue32(main) + hipStreamWaitValue32(side)
//   F  as C but hipStreamWriteValue32(side) + hipStreamWaitValue32(main)
// build: hipcc --offload-arch=gfx950 -O2 tools/microbench/sync_cost.hip -o /tmp/sync_cost
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
__global__ void small(float* x) { x[threadIdx.x] += 1.f; }
int main() {
  float* d; hipMalloc(&d, 4096); hipMemset(d, 0, 4096);
  unsigned* flag; hipMalloc(&flag, 64 * 4); hipMemset(flag, 0, 256);
  hipStream_t m, s; hipStreamCreateWithFlags(&m, hipStreamNonBlocking); hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  const int N = 100;
  hipEvent_t ev[2 * N]; for (auto& e : ev) hipEventCreateWithFlags(&e, hipEventDisableTiming);
  hipEvent_t t0, t1; hipEventCreate(&t0); hipEventCreate(&t1);
  for (int variant = 0; variant < 6; ++variant) {
    float best = 1e9;
    for (int rep = 0; rep < 5; ++rep) {
      hipMemsetAsync(flag, 0, 256, m); hipDeviceSynchronize();
      unsigned val = 0;
      hipEventRecord(t0, m);
      for (int i = 0; i < N; ++i) {
        hipLaunchKernelGGL(small, dim3(1), dim3(64), 0, m, d);
        switch (variant) {
          case 1: hipEventRecord(ev[i], m); hipStreamWaitEvent(s, ev[i], 0); break;
          case 2: hipEventRecord(ev[i], s); hipStreamWaitEvent(m, ev[i], 0); break;
          case 3: hipEventRecord(ev[i], m); hipStreamWaitEvent(s, ev[i], 0); hipLaunchKernelGGL(small, dim3(1), dim3(64), 0, s, d + 1024);
                  hipEventRecord(ev[N + i], s); hipStreamWaitEvent(m, ev[N + i], 0); break;
          case 4: ++val; hipStreamWriteValue32(m, flag, val, 0); hipStreamWaitValue32(s, flag, val, hipStreamWaitValueGte, 0xffffffffu); break;
          case 5: ++val; hipStreamWriteValue32(s, flag + 16, val, 0); hipStreamWaitValue32(m, flag + 16, val, hipStreamWaitValueGte, 0xffffffffu); break;
        }
      }
      hipEventRecord(t1, m);
      hipDeviceSynchronize();
      float ms = 0; hipEventElapsedTime(&ms, t0, t1);
      if (ms < best) best = ms;
    }
    printf("variant %c: %.2f us per iteration (main-stream timeline)\n", 'A' + variant, best * 1e3 / N);
  }
  return 0;
}
