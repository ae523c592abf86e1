// The frozen stem of the ResNet as ONE kernel: image layout (fp32 NCHW -> bf16) + conv1 7x7 / stride 2 / pad 3 (3 -> 64) +
// eval-mode BatchNorm (folded scale, bias) + ReLU + MaxPool 3x3 / stride 2 / pad 1
// (reference: mmdet/models/backbones/resnet.py:598-645 `_make_stem_layer`, :634-638 the forward's first three lines).
//
// Why a kernel of its own.  As three launches the stem was the implicit-GEMM convolution on an 8-channel copy of the image with
// K = 49 taps x 8 = 392 -> 448 stored columns for 147 real ones, wrote 69 MB of 400 x 672 x 64 activations per two images and read
// them back for the pooling: 72 us alone, 140 - 180 us inside the step beside the previous backward pass's tail.  Here a workgroup
// owns a 4 x 12 tile of POOLED outputs: it stages the 23 x 55 x 3 input patch that tile depends on in LDS as bf16 (7.7 KB, read
// once, coalesced along the image rows), computes the 9 x 25 stem outputs under the tile (one row / column of halo: 17 % more MFMA
// work than the ideal) as a 64 x 256 x 176 GEMM on v_mfma_f32_32x32x16_bf16 straight out of that patch - the 21 (kx, c) values of
// one tap row are CONTIGUOUS in the patch, so a B fragment is four aligned 4-byte LDS reads, K = 7 tap rows x 24 = 168 -> 176 -
// applies scale / bias / ReLU / bf16 rounding in the accumulator registers exactly as the convolution's epilogue does, keeps the
// rounded tile in LDS and pools it.  HBM traffic: 26 MB of image in, 17 MB of pooled activations out.
#include <hip/hip_runtime.h>

#include "common.hpp"

#pragma clang fp contract(off)

namespace {

constexpr int TPH = 4, TPW = 12;                 // pooled outputs per tile: 9 x 25 = 225 stem positions = 8 MFMA column tiles, 2 per wave
constexpr int SR = 2 * TPH + 1, SC = 2 * TPW + 1;      // stem rows / columns under them (3x3 / 2 pooling, one halo line)
constexpr int IR = 2 * SR + 5, IC = 2 * SC + 5;        // input rows / columns under those (7x7 / 2)
constexpr int PROW = IC * 3 + 3;                  // bf16 elements per patch row: 55 x 3 = 165, padded (group reads run to 6 (SC - 1) + 24 = 168)
static_assert(6 * (SC - 1) + 24 <= PROW && PROW % 2 == 0, "a B fragment stays inside its patch row, 4-byte aligned");
constexpr int NPX = SR * SC;                     // 225 stem positions
constexpr int NT = (NPX + 31) / 32;              // 8 MFMA column tiles
constexpr int SROW = 72;                         // bf16 elements per staged stem position: 64 + 8 (144 B rows: conflict-free 16-byte reads)
constexpr int KSTEPS = 11;                       // 22 groups of 8 k-values: 7 tap rows x 3 groups, + 1 group of zero weights
constexpr int STEM_T = 256;

struct StemK {
  const float* img; const uint16_t* w; const float* scale; const float* bias; uint16_t* out;
  int n, H, W, SH, SW, PH, PW, ld_out, tiles_x, tiles_y, tiles;
  int half_last;     // 1: image n - 1 is not in memory - it is image n - 2 at half size (bilinear) in the top-left corner of a zero canvas
};

__device__ __forceinline__ float mul1(float a, float b) { return a * b; }      // (contract(off): one rounding each, as conv.hip's mul_nc / add_nc)
__device__ __forceinline__ float add1(float a, float b) { return a + b; }

__device__ __forceinline__ uint32_t max2(uint32_t a, uint32_t b) {      // two non-negative bf16: integer order == value order
  const uint32_t lo = max(a & 0xffffu, b & 0xffffu), hi = max(a >> 16, b >> 16);
  return lo | (hi << 16);
}

__global__ __launch_bounds__(STEM_T) void stem_pool_kernel(const StemK p) {
  __shared__ __attribute__((aligned(16))) uint16_t patch[IR * PROW + 8];
  __shared__ __attribute__((aligned(16))) uint16_t stem[NT * 32 * SROW];
  __shared__ __attribute__((aligned(16))) float sb[128];             // folded BatchNorm: scale[64], bias[64]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // ---- weights: [22 groups][64 couts][8] bf16, group g = tap row g / 3, (kx, c) values 8 (g % 3) .. + 8 of its 21 (zero beyond):
  // a wave keeps its A fragments in registers across all the tiles its (persistent) workgroup walks through.  (Measured, N = 2:
  // fragments in registers + batched patch loads 56.8 us at one wave per SIMD; fragments in LDS 60.8 us - the register allocator
  // takes the freed registers for the unrolled loops and stays at one wave; rolled patch loads 61.6 us at two waves per SIMD.)
  bf16x8 A[KSTEPS][2];
#pragma unroll
  for (int k = 0; k < KSTEPS; ++k)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
      A[k][mt] = *reinterpret_cast<const bf16x8*>(p.w + ((size_t)(2 * k + (lane >> 5)) * 64 + mt * 32 + (lane & 31)) * 8);
  if (tid < 128) sb[tid] = tid < 64 ? p.scale[tid] : p.bias[tid - 64];
  for (int tile = blockIdx.x; tile < p.tiles; tile += gridDim.x) {
  int t = tile;
  const int tx = t % p.tiles_x;
  t /= p.tiles_x;
  const int ty = t % p.tiles_y, b = t / p.tiles_y;
  const int py0 = ty * TPH, px0 = tx * TPW;
  const int sy0 = 2 * py0 - 1, sx0 = 2 * px0 - 1;
  const int iy0 = 2 * sy0 - 3, ix0 = 2 * sx0 - 3;
  __syncthreads();           // the previous tile's pooling has read `stem`; its MFMA loop has read `patch`
  // ---- input patch: plane by plane, consecutive threads along the image row; zero outside the image (the convolution's padding).
  // All of a thread's loads are issued before the first is used (a rolled load -> convert -> store loop pays the memory latency
  // fifteen times per tile)
  {
    constexpr int NLD = (3 * IR * IC + STEM_T - 1) / STEM_T;
    float v[NLD];
    if (p.half_last && b == p.n - 1) {
      // SemiEpochBasedRunner's scale-invariant copy (semi_epoch_based_runner.py:186-204: F.interpolate(img[-1:], size = (H / 2, W / 2),
      // mode = 'bilinear') pasted into a zero canvas) read straight from its source image: for an exact factor of two the bilinear
      // sample of output (y, x) is 0.5 * (0.5 * v00 + 0.5 * v01) + 0.5 * (0.5 * v10 + 0.5 * v11) - halvings are exact in fp32, so it is
      // ((v00 + v01) + (v10 + v11)) / 4 with those three roundings, bit for bit what the framework kernel writes
      const int hh = p.H >> 1, hw_ = p.W >> 1;
#pragma unroll
      for (int i = 0; i < NLD; ++i) {
        const int idx = tid + i * STEM_T;
        const int c = idx / (IR * IC), rem = idx - c * (IR * IC);
        const int r = rem / IC, col = rem - r * IC;
        const int iy = iy0 + r, ix = ix0 + col;
        v[i] = 0.f;
        if (idx < 3 * IR * IC && (unsigned)iy < (unsigned)hh && (unsigned)ix < (unsigned)hw_) {
          const float* s0 = p.img + ((long long)((b - 1) * 3 + c) * p.H + 2 * iy) * p.W + 2 * ix;
          const float2 t0 = *reinterpret_cast<const float2*>(s0), t1 = *reinterpret_cast<const float2*>(s0 + p.W);
          v[i] = mul1(add1(add1(t0.x, t0.y), add1(t1.x, t1.y)), 0.25f);
        }
      }
    } else {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int idx = tid + i * STEM_T;
      const int c = idx / (IR * IC), rem = idx - c * (IR * IC);
      const int r = rem / IC, col = rem - r * IC;
      const int iy = iy0 + r, ix = ix0 + col;
      v[i] = 0.f;
      if (idx < 3 * IR * IC && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W)
        v[i] = p.img[((long long)(b * 3 + c) * p.H + iy) * p.W + ix];
    }
    }
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int idx = tid + i * STEM_T;
      const int c = idx / (IR * IC), rem = idx - c * (IR * IC);
      const int r = rem / IC, col = rem - r * IC;
      if (idx < 3 * IR * IC) patch[r * PROW + col * 3 + c] = f2bf(v[i]);
    }
  }
  for (int idx = tid; idx < IR * 3 + 8; idx += STEM_T) {      // the padding the last group of a row reads (zero weights: any finite value)
    if (idx < IR * 3) patch[(idx / 3) * PROW + IC * 3 + idx % 3] = 0;
    else patch[IR * PROW + idx - IR * 3] = 0;
  }
  __syncthreads();
  for (int nt = wave; nt < NT; nt += STEM_T / 64) {
    const int pxl = nt * 32 + (lane & 31);
    const bool valid = pxl < NPX;
    const int q = valid ? pxl : 0;
    const int sr = q / SC, sc = q - sr * SC;
    f32x16 acc[2] = {};
#pragma unroll
    for (int k = 0; k < KSTEPS; ++k) {
      const int g = min(2 * k + (lane >> 5), 20);         // (group 21 has zero weights: re-read group 20)
      const int ky = g / 3, j = g - ky * 3;
      const uint32_t* src = reinterpret_cast<const uint32_t*>(patch + (2 * sr + ky) * PROW + 6 * sc + 8 * j);
      const u32x4 raw = {src[0], src[1], src[2], src[3]};
      const bf16x8 B = __builtin_bit_cast(bf16x8, raw);
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[k][0], B, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[k][1], B, acc[1], 0, 0, 0);
    }
    // scale / bias / ReLU / rounding in the accumulator registers; stem positions outside the feature map are the pooling's
    // padding: 0 stands for -inf there because every window holds at least one real, non-negative value
    const int sy = sy0 + sr, sx = sx0 + sc;
    const bool inside = valid && (unsigned)sy < (unsigned)p.SH && (unsigned)sx < (unsigned)p.SW;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int co = mt * 32 + 8 * i + 4 * (lane >> 5);
        const f32x4 s4 = *reinterpret_cast<const f32x4*>(sb + co), b4 = *reinterpret_cast<const f32x4*>(sb + 64 + co);
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(add1(mul1(acc[mt][4 * i + e], s4[e]), b4[e]), 0.f);
        u32x2 o = {pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
        if (!inside) o = u32x2{0u, 0u};
        *reinterpret_cast<u32x2*>(stem + pxl * SROW + co) = o;
      }
  }
  __syncthreads();
  // ---- 3x3 / 2 max pooling of the staged tile: TPH x TPW outputs x 8 channel groups of 8
  for (int it = tid; it < TPH * TPW * 8; it += STEM_T) {
    const int cg = it & 7, op = it >> 3;
    const int oy = op / TPW, ox = op - oy * TPW;
    const int py = py0 + oy, px = px0 + ox;
    if (py >= p.PH || px >= p.PW) continue;
    u32x4 m = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(stem + ((2 * oy + dy) * SC + 2 * ox + dx) * SROW + cg * 8);
#pragma unroll
        for (int e = 0; e < 4; ++e) m[e] = max2(m[e], v[e]);
      }
    *reinterpret_cast<u32x4*>(p.out + (((long long)b * p.PH + py) * p.PW + px) * p.ld_out + cg * 8) = m;
  }
  }
}

}  // namespace

extern "C" int dsl_stem_pool_half(const float* img, const void* w_groups, const float* scale, const float* bias, void* out, int ld_out,
                                  int n, int h, int w, int half_last, void* stream);
extern "C" int dsl_stem_pool(const float* img, const void* w_groups, const float* scale, const float* bias, void* out, int ld_out,
                             int n, int h, int w, void* stream) {
  return dsl_stem_pool_half(img, w_groups, scale, bias, out, ld_out, n, h, w, 0, stream);
}
extern "C" int dsl_stem_pool_half(const float* img, const void* w_groups, const float* scale, const float* bias, void* out, int ld_out,
                                  int n, int h, int w, int half_last, void* stream) {
  DSL_CHECK(img && w_groups && scale && bias && out, "dsl_stem_pool: null pointer");
  DSL_CHECK(n >= 1 && h >= 7 && w >= 7 && ld_out >= 64 && ld_out % 8 == 0, "dsl_stem_pool: bad shape");
  DSL_CHECK(!half_last || (n >= 2 && h % 2 == 0 && w % 2 == 0), "dsl_stem_pool: half_last needs n >= 2 and even sizes (n=%d h=%d w=%d)", n, h, w);
  StemK k;
  k.half_last = half_last ? 1 : 0;
  k.img = img; k.w = (const uint16_t*)w_groups; k.scale = scale; k.bias = bias; k.out = (uint16_t*)out;
  k.n = n; k.H = h; k.W = w;
  k.SH = (h + 6 - 7) / 2 + 1; k.SW = (w + 6 - 7) / 2 + 1;
  k.PH = (k.SH + 2 - 3) / 2 + 1; k.PW = (k.SW + 2 - 3) / 2 + 1;
  k.ld_out = ld_out;
  k.tiles_x = (k.PW + TPW - 1) / TPW; k.tiles_y = (k.PH + TPH - 1) / TPH;
  const long long blocks = (long long)n * k.tiles_x * k.tiles_y;
  DSL_CHECK(blocks < (1ll << 31), "dsl_stem_pool: too many tiles");
  k.tiles = (int)blocks;
  // persistent workgroups, two per CU: a wave loads its weight fragments once and walks through tiles blockIdx.x, + gridDim.x, ...
  const unsigned grid = (unsigned)min(blocks, 512ll);
  hipLaunchKernelGGL(stem_pool_kernel, dim3(grid), dim3(STEM_T), 0, (hipStream_t)stream, k);
  DSL_LAUNCH_CHECK("stem_pool_kernel");
  return 0;
}
