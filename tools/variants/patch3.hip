// Activation-stationary 3x3 / stride 1 / pad 1 convolution, 64 -> 64 channels, NHWC bf16, + folded BatchNorm (scale, bias) + ReLU:
// the middle convolution of the frozen layer1 bottlenecks (reference: mmdet/models/backbones/resnet.py:262-301 `Bottleneck.forward`,
// conv2 / bn2 / relu; layer1 is frozen by `frozen_stages=1`, :616-632).
//
// Why a kernel of its own (the addressing is the one `stem_pool_kernel` proved).  The implicit GEMM fetches
// every pixel row NINE times - once per tap, through L2 into LDS - and at 64 channels a 128 x 128 tile has only 9 K tiles to hide its
// prologue and epilogue behind: 30 us for 9.9 GFLOP / 34 MB (8.6 us of memory time at 4 TB/s).  Here a workgroup stages a pixel tile
// PLUS ITS HALO once, as [row][col][64 + 8] bf16 in LDS; with K ordered [ky][kx][c] (the layout the forward weights already have,
// ParamStore: [cout][kh][kw][cin]) a B fragment of v_mfma_f32_32x32x16_bf16 is ONE aligned 16-byte LDS read at
// patch[r + ky][col + kx][16 j + 8 (lane >> 5)], the nine taps are address arithmetic, and nothing is fetched twice.  The weights never
// touch LDS: 64 couts x 576 k-values are 36 k-steps x 2 row tiles of A fragments = 288 registers a wave loads once and keeps across
// all the tiles its persistent workgroup walks through (one wave per SIMD, 512 registers each).  Per k-step a wave issues one 16-byte
// LDS read per 32 pixels for two MFMAs (64 matrix-pipe cycles): 64 B / clk / CU of LDS reads, half the port.  The next tile's patch is
// in flight (global loads into registers) while the current one is multiplied, and lands in the other LDS buffer afterwards.
#include <hip/hip_runtime.h>

#include "common.hpp"

#pragma clang fp contract(off)

namespace {

constexpr int TH = 8, TW = 16;                    // output pixels per tile: 128 = 4 waves x 32 (two rows of 16 per wave)
constexpr int PR = TH + 2, PC = TW + 2;           // patch rows / columns (one halo line on every side)
constexpr int PIX = 72;                           // bf16 elements per staged pixel: 64 + 8 (144-byte rows: conflict-free 16-byte reads)
constexpr int PATCH = PR * PC * PIX;              // 12 960 elements = 25 920 B per buffer
constexpr int NCHUNK = PR * PC * 8;               // 16-byte chunks per patch
constexpr int P3_T = 256;
constexpr int NLD = (NCHUNK + P3_T - 1) / P3_T;   // 6 chunk loads per thread
constexpr int KSTEPS = 36;                        // 9 taps x 64 channels / 16
constexpr int KG = 6;                             // k-steps per software-pipeline group
constexpr int WROW = KSTEPS * 16 + 8;             // bf16 elements per staged weight row: 576 + 8 (1 168-byte rows)
constexpr int STAGE = TH * TW * PIX;                // one staged output tile
constexpr int SMEM = 2 * PATCH + 2 * STAGE + 256;  // two patches, two output stages, scale[64] + bias[64] as fp32: 89 216 B
static_assert(64 * WROW <= 2 * PATCH + 2 * STAGE, "the weights pass through the patch / stage bytes, not through scale / bias");

struct P3K {
  const uint16_t* src; const uint16_t* w; const float* scale; const float* bias; uint16_t* dst;
  int n, H, W, ld_src, ld_dst, tiles_x, tiles_y, tiles, relu;
};

__device__ __forceinline__ float mul1(float a, float b) { return a * b; }      // (contract(off): one rounding each, as conv.hip's mul_nc / add_nc)
__device__ __forceinline__ float add1(float a, float b) { return a + b; }

// Workgroup barrier that orders LDS traffic only (__syncthreads() also waits for the wave's global stores to be acknowledged:
// once per tile that is the store latency, serialised; conv.hip's lds_barrier)
__device__ __forceinline__ void lds_barrier3() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__global__ __launch_bounds__(P3_T) void conv3x3_c64_patch_kernel(const P3K p) {
  __shared__ __attribute__((aligned(16))) uint16_t smem[SMEM];
  uint16_t* const patch0 = smem;                  // [2][PATCH]
  uint16_t* const stage0 = smem + 2 * PATCH;      // [2][TH * TW][PIX]
  float* const sb = reinterpret_cast<float*>(smem + 2 * PATCH + 2 * STAGE);      // folded BatchNorm: scale[64], bias[64]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fhalf = lane >> 5;
  // ---- weights: A fragment of k-step k, row tile mt = 16 bytes of row (32 mt + lane % 32) at k-values 16 k + 8 (lane / 32).
  // They pass through LDS once (coalesced copy with every load in flight, rows padded to 1 168 B: conflict-free fragment reads):
  // straight from memory the compiler loads the 288 registers two fragments at a time, each pair behind a full memory round
  // trip (the values that live in accumulation registers are copied there as they arrive) - 36 round trips, ~ 15 of the 30 us the
  // first version of this kernel took for N = 2.
  if (tid < 128) sb[tid] = tid < 64 ? p.scale[tid] : p.bias[tid - 64];
  {
    constexpr int NW = 64 * (KSTEPS * 2) / P3_T;  // 18 chunks of 16 bytes per thread
    u32x4 v[NW];
#pragma unroll
    for (int i = 0; i < NW; ++i) {
      const int idx = tid + i * P3_T;
      v[i] = *reinterpret_cast<const u32x4*>(p.w + (size_t)(idx / (KSTEPS * 2)) * (KSTEPS * 16) + (idx % (KSTEPS * 2)) * 8);
    }
#pragma unroll
    for (int i = 0; i < NW; ++i) {
      const int idx = tid + i * P3_T;
      *reinterpret_cast<u32x4*>(smem + (idx / (KSTEPS * 2)) * WROW + (idx % (KSTEPS * 2)) * 8) = v[i];
    }
  }
  __syncthreads();
  bf16x8 A[KSTEPS][2];
#pragma unroll
  for (int k = 0; k < KSTEPS; ++k)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
      A[k][mt] = *reinterpret_cast<const bf16x8*>(smem + (mt * 32 + (lane & 31)) * WROW + 16 * k + 8 * fhalf);
  __syncthreads();                                // the weights are in registers: the same bytes become the patches

  u32x4 ld[NLD];
  auto fetch = [&](int tile) {                    // the tile's patch -> registers (zero outside the image: the padding)
    int t = tile;
    const int tx = t % p.tiles_x;
    t /= p.tiles_x;
    const int ty = t % p.tiles_y, b = t / p.tiles_y;
    const int y0 = ty * TH - 1, x0 = tx * TW - 1;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int idx = tid + i * P3_T;
      const int px = idx >> 3, ch = idx & 7;
      const int r = px / PC, c = px - r * PC;
      const int y = y0 + r, x = x0 + c;
      ld[i] = u32x4{0u, 0u, 0u, 0u};
      if (idx < NCHUNK && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W)
        ld[i] = *reinterpret_cast<const u32x4*>(p.src + (((long long)b * p.H + y) * p.W + x) * p.ld_src + ch * 8);
    }
  };
  auto land = [&](int buf) {                      // registers -> LDS patch `buf`
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int idx = tid + i * P3_T;
      if (idx < NCHUNK) *reinterpret_cast<u32x4*>(patch0 + buf * PATCH + (idx >> 3) * PIX + (idx & 7) * 8) = ld[i];
    }
  };

  // a lane's pixel inside the tile: wave rows 2 w, 2 w + 1; 16 columns each
  const int q = lane & 31;
  const int prow = 2 * wave + (q >> 4), pcol = q & 15;
  const int boff = (prow * PC + pcol) * PIX + 8 * fhalf;

  int tile = blockIdx.x, buf = 0;
  if (tile < p.tiles) {
    fetch(tile);
    land(0);
  }
  __syncthreads();
  // One barrier per tile: the next patch lands in the OTHER buffer before it (so the barrier that publishes the staged outputs
  // publishes the patch too), and the staged outputs alternate between two buffers (a wave still storing tile i is not overtaken
  // by a wave staging tile i + 1; tile i + 2's staging is behind tile i + 1's barrier).
  for (; tile < p.tiles; tile += gridDim.x, buf ^= 1) {
    const int next = tile + gridDim.x;
    uint16_t* const stage = stage0 + buf * STAGE;
    if (next < p.tiles) fetch(next);              // in flight during the multiplication below
    f32x16 acc[2] = {};
    const uint16_t* pb = patch0 + buf * PATCH + boff;
    // One wave per SIMD: nobody else hides the LDS latency, so the B fragments of k-step group g + 1 are requested before group
    // g is multiplied (two register sets of KG fragments; the scheduling barriers keep the compiler from sinking the reads back
    // to their uses - measured: 30.2 us for N = 2 without them, the same as the implicit GEMM)
    auto bfrag = [&](int k) {
      const int tap = k >> 2, j = k & 3;
      const int ky = tap / 3, kx = tap - 3 * ky;
      return *reinterpret_cast<const bf16x8*>(pb + (ky * PC + kx) * PIX + 16 * j);
    };
    bf16x8 Bq[2][KG];
#pragma unroll
    for (int i = 0; i < KG; ++i) Bq[0][i] = bfrag(i);
#pragma unroll
    for (int g = 0; g < KSTEPS / KG; ++g) {
      if (g + 1 < KSTEPS / KG) {
#pragma unroll
        for (int i = 0; i < KG; ++i) Bq[(g + 1) & 1][i] = bfrag((g + 1) * KG + i);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < KG; ++i) {
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[g * KG + i][0], Bq[g & 1][i], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[g * KG + i][1], Bq[g & 1][i], acc[1], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // ---- scale / bias / ReLU / rounding in the accumulator registers (a lane owns 4 consecutive couts of its pixel per group of 8),
    // the tile staged as bf16 rows [pixel][64 + 8] and written out as 16-byte stores
    const int pxl = wave * 32 + q;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int co = mt * 32 + 8 * i + 4 * fhalf;
        const f32x4 s4 = *reinterpret_cast<const f32x4*>(sb + co), b4 = *reinterpret_cast<const f32x4*>(sb + 64 + co);
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[e] = add1(mul1(acc[mt][4 * i + e], s4[e]), b4[e]);
          if (p.relu) v[e] = fmaxf(v[e], 0.f);
        }
        const u32x2 o = {pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
        *reinterpret_cast<u32x2*>(stage + pxl * PIX + co) = o;
      }
    if (next < p.tiles) land(buf ^ 1);            // (patch[buf ^ 1] was last read before the previous barrier; its loads had the whole multiplication)
    lds_barrier3();
    {
      int t = tile;
      const int tx = t % p.tiles_x;
      t /= p.tiles_x;
      const int ty = t % p.tiles_y, b = t / p.tiles_y;
#pragma unroll
      for (int i = 0; i < TH * TW * 8 / P3_T; ++i) {
        const int idx = tid + i * P3_T;
        const int px = idx >> 3, ch = idx & 7;
        const int y = ty * TH + (px >> 4), x = tx * TW + (px & 15);
        if (y < p.H && x < p.W)
          *reinterpret_cast<u32x4*>(p.dst + (((long long)b * p.H + y) * p.W + x) * p.ld_dst + ch * 8) =
              *reinterpret_cast<const u32x4*>(stage + px * PIX + ch * 8);
      }
    }
  }
}

}  // namespace

extern "C" int dsl_conv3x3_c64_patch(const void* src, int ld_src, const void* wgt, const float* scale, const float* bias, void* dst,
                                     int ld_dst, int n, int h, int w, int relu, void* stream) {
  DSL_CHECK(src && wgt && scale && bias && dst, "dsl_conv3x3_c64_patch: null pointer");
  DSL_CHECK(n >= 1 && h >= 1 && w >= 1 && ld_src >= 64 && ld_src % 8 == 0 && ld_dst >= 64 && ld_dst % 8 == 0,
            "dsl_conv3x3_c64_patch: bad shape (n=%d h=%d w=%d ld_src=%d ld_dst=%d)", n, h, w, ld_src, ld_dst);
  P3K k;
  k.src = (const uint16_t*)src; k.w = (const uint16_t*)wgt; k.scale = scale; k.bias = bias; k.dst = (uint16_t*)dst;
  k.n = n; k.H = h; k.W = w; k.ld_src = ld_src; k.ld_dst = ld_dst; k.relu = relu;
  k.tiles_x = (w + TW - 1) / TW; k.tiles_y = (h + TH - 1) / TH;
  const long long tiles = (long long)n * k.tiles_x * k.tiles_y;
  DSL_CHECK(tiles < (1ll << 31), "dsl_conv3x3_c64_patch: too many tiles");
  k.tiles = (int)tiles;
  // persistent workgroups, one per CU (a wave keeps all 288 weight registers): tiles blockIdx.x, + gridDim.x, ...
  const unsigned grid = (unsigned)(tiles < 256 ? tiles : 256);
  hipLaunchKernelGGL(conv3x3_c64_patch_kernel, dim3(grid), dim3(P3_T), 0, (hipStream_t)stream, k);
  DSL_LAUNCH_CHECK("conv3x3_c64_patch_kernel");
  return 0;
}
