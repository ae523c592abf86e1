// Kernel-boundary cost in one stream: N dependent launches of a tiny kernel, issued (a) one by one, (b) as ONE captured hipGraph
// replay.  hipcc --offload-arch=gfx950 -O2 tools/microbench/graph_gap.hip -o /tmp/graph_gap && /tmp/graph_gap
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void tiny(float* p, int spin) {
  float v = p[blockIdx.x * blockDim.x + threadIdx.x];
  for (int i = 0; i < spin; ++i) v = v * 1.0001f + 0.5f;
  p[blockIdx.x * blockDim.x + threadIdx.x] = v;
}
int main() {
  float* d;
  CK(hipMalloc(&d, 256 * 512 * 4));
  CK(hipMemset(d, 0, 256 * 512 * 4));
  hipStream_t st;
  CK(hipStreamCreate(&st));
  const int N = 200;
  for (int spin : {0, 2000, 20000}) {
    auto run_plain = [&]() { for (int i = 0; i < N; ++i) hipLaunchKernelGGL(tiny, dim3(234), dim3(512), 0, st, d, spin); };
    run_plain();
    CK(hipStreamSynchronize(st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, st));
    run_plain();
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    float ms_plain;
    CK(hipEventElapsedTime(&ms_plain, e0, e1));
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    run_plain();
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, st));
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    CK(hipGraphLaunch(ge, st));
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    float ms_graph;
    CK(hipEventElapsedTime(&ms_graph, e0, e1));
    // one launch alone, for the kernel's own duration
    CK(hipEventRecord(e0, st));
    hipLaunchKernelGGL(tiny, dim3(234), dim3(512), 0, st, d, spin);
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    float ms_one;
    CK(hipEventElapsedTime(&ms_one, e0, e1));
    printf("spin %6d: stream %.2f us per launch, graph %.2f us per node, (one launch between events %.2f us)\n", spin, ms_plain * 1e3 / N,
           ms_graph * 1e3 / N, ms_one * 1e3);
  }
  return 0;
}
