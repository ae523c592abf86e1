// A whole 64-plane bottleneck of the FROZEN layer1 as ONE kernel (round 4):
//   out = relu( bn3(conv3_1x1( relu(bn2(conv2_3x3( relu(bn1(conv1_1x1(x))) ))) )) + identity ),
//   identity = x (blocks 1, 2: Cin = 256)  or  bn_ds(conv_ds_1x1(x)) (block 0: Cin = 64)
// (reference: mmdet/models/backbones/resnet.py:262-301 `Bottleneck.forward`, caffe style, stride 1; layer1 is frozen by
// `frozen_stages=1`, :616-632, so no intermediate is needed by any backward pass.)
//
// Why.  Step-level ablation (tools/step_ablation.sh, profiles/r04_step_ablation.txt): with layer1's ten launches
// skipped the training step is 0.30 ms (6.5 %) shorter - they are 0.4 ms of kernel time (134 400 pixels x small channel counts: the
// 1x1 64 -> 256 convolution is ONE K tile, all prologue and epilogue; every intermediate goes through memory) that overlaps with
// the previous backward pass's tail on paper and costs three quarters of its length in practice.  Here a workgroup walks 8 x 16
// pixel tiles: conv1 is computed for the tile plus a one-pixel halo straight from global memory into an LDS patch (zero outside
// the image: conv2's padding), conv2 is csrc/patch3.hip's activation-stationary 3x3 out of that patch (weights in 288 registers
// per wave), conv3 (+ the downsample convolution of block 0) reads the staged conv2 tile out of LDS, and the residual epilogue
// leaves through an fp32 staging round as 16-byte stores.  Per block: x read once (+ the halo), out written once.
//
// Arithmetic = the three (four) launches it replaces, operation for operation: same K order per MFMA chain (ci ascending / tap
// major), a1 / a2 / the downsample identity rounded to bf16 where the separate launches store them, epilogues mul, add, (+ identity),
// ReLU, round with one rounding each (fp contract off) - tests/test_kernels_gpu.py::test_bottleneck64_fused compares bits.
#include <hip/hip_runtime.h>

#include "common.hpp"

#pragma clang fp contract(off)

namespace {

constexpr int TH = 8, TW = 16;                    // output pixels per tile: 128 = 4 waves x 32 (two rows of 16 per wave)
constexpr int PR = TH + 2, PC = TW + 2;           // conv1 patch (tile + halo)
constexpr int NPATCH = PR * PC;                   // 180 pixels, computed as 6 MFMA columns of 32 (192)
constexpr int PIX = 72;                           // bf16 elements per staged pixel (64 + 8: 144-byte rows, conflict-free 16-byte reads)
constexpr int PATCH = 192 * PIX;                  // elements
constexpr int STAGE = TH * TW * PIX;
constexpr int KSTEPS = 36, KG = 6;
constexpr int WROW2 = KSTEPS * 16 + 8;            // staged conv2 weight row (startup only)
constexpr int W3ROW = 72;
constexpr int FROW = 68;                          // fp32 elements per staged pixel of an output quarter (64 + 4)
constexpr int BN_T = 256;
constexpr int EPI = (TH * TW * FROW * 2) + STAGE;    // fp32 quarter (as bf16-element count) + identity quarter
constexpr int WORK = (PATCH + STAGE) > EPI ? (PATCH + STAGE) : EPI;
constexpr int NSB = 64 * 4 + 256 * 4;             // s1 b1 s2 b2 [64], s3 b3 sds bds [256]

struct BnK {
  const uint16_t* x; uint16_t* out;
  const uint16_t* w1; const uint16_t* w2; const uint16_t* w3; const uint16_t* wds;
  const float* s1; const float* b1; const float* s2; const float* b2; const float* s3; const float* b3; const float* sds; const float* bds;
  int n, H, W, ld_x, ld_out, tiles_x, tiles_y, tiles;
  int dbg;      // timing probes (DSL_BNECK_DBG, results are wrong): 1 no x loads, 2 no identity loads, 4 no stores, 8 no conv2, 16 no epilogue
};

template <int CIN, bool DS>
constexpr int bn_lds_elems() { return 64 * (CIN + 8) + 256 * W3ROW * (DS ? 2 : 1) + WORK + NSB * 2; }

__device__ __forceinline__ float mul1(float a, float b) { return a * b; }
__device__ __forceinline__ float add1(float a, float b) { return a + b; }
__device__ __forceinline__ bf16x8 zero8() { return __builtin_bit_cast(bf16x8, u32x4{0u, 0u, 0u, 0u}); }
__device__ __forceinline__ void lds_barrier_b() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int CIN, bool DS>
__global__ __launch_bounds__(BN_T) void bottleneck64_kernel(const BnK p) {
  extern __shared__ __attribute__((aligned(16))) uint16_t smem[];
  constexpr int W1ROW = CIN + 8;
  constexpr int K1 = CIN / 16;
  uint16_t* const w1s = smem;                                  // [64][W1ROW]
  uint16_t* const w3s = w1s + 64 * W1ROW;                      // [256][W3ROW]
  uint16_t* const wdss = w3s + 256 * W3ROW;                    // [256][W3ROW] (DS only)
  uint16_t* const work = wdss + (DS ? 256 * W3ROW : 0);
  uint16_t* const patch = work;                                // [192][PIX]   phases 1-2
  uint16_t* const stage = work + PATCH;                        // [128][PIX]   phases 2-3
  float* const fq = reinterpret_cast<float*>(work);            // [128][FROW]  epilogue (aliases patch / stage)
  uint16_t* const iq = work + TH * TW * FROW * 2;              // [128][PIX]   epilogue: identity quarter (DS)
  float* const sb = reinterpret_cast<float*>(work + WORK);     // folded BatchNorms
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fhalf = lane >> 5, q = lane & 31;

  // ---- conv2's weights: through LDS once, then 288 registers per wave for the life of the workgroup (csrc/patch3.hip)
  {
    constexpr int NW = 64 * (KSTEPS * 2) / BN_T;
    u32x4 v[NW];
#pragma unroll
    for (int i = 0; i < NW; ++i) {
      const int idx = tid + i * BN_T;
      v[i] = *reinterpret_cast<const u32x4*>(p.w2 + (size_t)(idx / (KSTEPS * 2)) * (KSTEPS * 16) + (idx % (KSTEPS * 2)) * 8);
    }
#pragma unroll
    for (int i = 0; i < NW; ++i) {
      const int idx = tid + i * BN_T;
      *reinterpret_cast<u32x4*>(smem + (idx / (KSTEPS * 2)) * WROW2 + (idx % (KSTEPS * 2)) * 8) = v[i];
    }
  }
  __syncthreads();
  bf16x8 A2[KSTEPS][2];
#pragma unroll
  for (int k = 0; k < KSTEPS; ++k)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) A2[k][mt] = *reinterpret_cast<const bf16x8*>(smem + (mt * 32 + q) * WROW2 + 16 * k + 8 * fhalf);
  __syncthreads();
  // ---- conv1 / conv3 / downsample weights and the folded BatchNorms stay in LDS
  for (int idx = tid; idx < 64 * (CIN / 8); idx += BN_T) {
    const int r = idx / (CIN / 8), c = idx % (CIN / 8);
    *reinterpret_cast<u32x4*>(w1s + r * W1ROW + c * 8) = *reinterpret_cast<const u32x4*>(p.w1 + (size_t)r * CIN + c * 8);
  }
  for (int idx = tid; idx < 256 * 8; idx += BN_T) {
    const int r = idx >> 3, c = idx & 7;
    *reinterpret_cast<u32x4*>(w3s + r * W3ROW + c * 8) = *reinterpret_cast<const u32x4*>(p.w3 + (size_t)r * 64 + c * 8);
    if (DS) *reinterpret_cast<u32x4*>(wdss + r * W3ROW + c * 8) = *reinterpret_cast<const u32x4*>(p.wds + (size_t)r * 64 + c * 8);
  }
  for (int i = tid; i < NSB; i += BN_T) {
    float v = 0.f;
    if (i < 64) v = p.s1[i];
    else if (i < 128) v = p.b1[i - 64];
    else if (i < 192) v = p.s2[i - 128];
    else if (i < 256) v = p.b2[i - 192];
    else if (i < 512) v = p.s3[i - 256];
    else if (i < 768) v = p.b3[i - 512];
    else if (DS && i < 1024) v = p.sds[i - 768];
    else if (DS) v = p.bds[i - 1024];
    sb[i] = v;
  }
  __syncthreads();
  const float* const s1 = sb;
  const float* const b1 = sb + 64;
  const float* const s2 = sb + 128;
  const float* const b2 = sb + 192;
  const float* const s3 = sb + 256;
  const float* const b3 = sb + 512;
  const float* const sds = sb + 768;
  const float* const bds = sb + 1024;

  // a lane's pixel inside the tile (conv2 / conv3): wave rows 2 w, 2 w + 1; 16 columns each
  const int prow = 2 * wave + (q >> 4), pcol = q & 15;
  const int boff = (prow * PC + pcol) * PIX + 8 * fhalf;
  const int pxl = wave * 32 + q;

  for (int tile = blockIdx.x; tile < p.tiles; tile += gridDim.x) {
    int t = tile;
    const int tx = t % p.tiles_x;
    t /= p.tiles_x;
    const int ty = t % p.tiles_y, b = t / p.tiles_y;
    const int y0 = ty * TH - 1, x0 = tx * TW - 1;
    const long long img0 = (long long)b * p.H * p.W;

    // ============ phase 1: a1 = relu(bn1(conv1(x))) for the 10 x 18 patch, zero outside the image (conv2's padding) ============
    for (int pt = wave; pt < 6; pt += 4) {
      const int pix = 32 * pt + q;
      const int r = pix / PC, c = pix - r * PC;
      const int y = y0 + r, x = x0 + c;
      const bool inimg = pix < NPATCH && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W && !(p.dbg & 1);
      const uint16_t* src = p.x + (img0 + (long long)y * p.W + x) * p.ld_x + 8 * fhalf;
      // the pixel's Cin channels come straight from memory, every 16-byte piece in flight before the first MFMA
      constexpr int KC = K1;          // (all of a pixel's channels in flight at once)
      f32x16 acc[2] = {};
#pragma unroll
      for (int j0 = 0; j0 < K1; j0 += KC) {
        bf16x8 bq[KC];
#pragma unroll
        for (int j = 0; j < KC; ++j) {
          bq[j] = zero8();
          if (inimg) bq[j] = *reinterpret_cast<const bf16x8*>(src + 16 * (j0 + j));
        }
#pragma unroll
        for (int j = 0; j < KC; ++j) {
#pragma unroll
          for (int mt = 0; mt < 2; ++mt) {
            const bf16x8 a = *reinterpret_cast<const bf16x8*>(w1s + (mt * 32 + q) * W1ROW + 16 * (j0 + j) + 8 * fhalf);
            acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bq[j], acc[mt], 0, 0, 0);
          }
        }
      }
      if (pix < NPATCH) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int co = mt * 32 + 8 * i + 4 * fhalf;
            const f32x4 s4 = *reinterpret_cast<const f32x4*>(s1 + co), b4 = *reinterpret_cast<const f32x4*>(b1 + co);
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              v[e] = fmaxf(add1(mul1(acc[mt][4 * i + e], s4[e]), b4[e]), 0.f);
              if (!inimg) v[e] = 0.f;
            }
            const u32x2 o = {pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
            *reinterpret_cast<u32x2*>(patch + pix * PIX + co) = o;
          }
      }
    }
    lds_barrier_b();

    // (requested here, consumed in the epilogue: conv2 and conv3 run while they are in flight)
    // this lane's output pixel and (block 0) its x fragments for the downsample convolution
    const int oy = ty * TH + prow, ox = tx * TW + pcol;
    const bool oin = oy < p.H && ox < p.W;
    bf16x8 bx[4];
    if (DS) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        bx[j] = zero8();
        if (oin) bx[j] = *reinterpret_cast<const bf16x8*>(p.x + (img0 + (long long)oy * p.W + ox) * p.ld_x + 16 * j + 8 * fhalf);
      }
    }
    // the store loop's chunks of this thread (4 per quarter): pixel and identity address; the identity of quarter qd + 1 is requested
    // while quarter qd is staged and stored (one wave per SIMD: a load issued where it is consumed costs its whole latency, sixteen
    // times per tile - the first version of this kernel spent more than half its time there)
    constexpr int NCH = TH * TW * 8 / BN_T;
    long long gpc[NCH];
    bool okc[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int idx = tid + i * BN_T;
      const int px = idx >> 3;
      const int y = ty * TH + (px >> 4), x = tx * TW + (px & 15);
      okc[i] = y < p.H && x < p.W;
      gpc[i] = img0 + (long long)y * p.W + x;
    }
    u32x4 idc[NCH];
    auto load_id = [&](int qd_) {
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        idc[i] = u32x4{0u, 0u, 0u, 0u};
        if (!DS && okc[i] && !(p.dbg & 2)) idc[i] = *reinterpret_cast<const u32x4*>(p.x + gpc[i] * p.ld_x + 64 * qd_ + ((tid + i * BN_T) & 7) * 8);
      }
    };
    load_id(0);
    // ============ phase 2: a2 = relu(bn2(conv2(a1))) out of the LDS patch (csrc/patch3.hip) ============
    {
      f32x16 acc[2] = {};
      const uint16_t* pb = patch + boff;
      auto bfrag = [&](int k) {
        const int tap = k >> 2, j = k & 3;
        const int ky = tap / 3, kx = tap - 3 * ky;
        return *reinterpret_cast<const bf16x8*>(pb + (ky * PC + kx) * PIX + 16 * j);
      };
      bf16x8 Bq[2][KG];
#pragma unroll
      for (int i = 0; i < KG; ++i) Bq[0][i] = bfrag(i);
#pragma unroll
      for (int g = 0; g < ((p.dbg & 8) ? 0 : KSTEPS / KG); ++g) {
        if (g + 1 < KSTEPS / KG) {
#pragma unroll
          for (int i = 0; i < KG; ++i) Bq[(g + 1) & 1][i] = bfrag((g + 1) * KG + i);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < KG; ++i) {
          acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A2[g * KG + i][0], Bq[g & 1][i], acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A2[g * KG + i][1], Bq[g & 1][i], acc[1], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int co = mt * 32 + 8 * i + 4 * fhalf;
          const f32x4 s4 = *reinterpret_cast<const f32x4*>(s2 + co), b4 = *reinterpret_cast<const f32x4*>(b2 + co);
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(add1(mul1(acc[mt][4 * i + e], s4[e]), b4[e]), 0.f);
          const u32x2 o = {pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
          *reinterpret_cast<u32x2*>(stage + pxl * PIX + co) = o;
        }
    }
    lds_barrier_b();

    // ============ phase 3: conv3 (1x1, 64 -> 256) on the staged a2 tile: every wave its 32 pixels x all 256 couts, 64 couts at a
    // time inside the epilogue loop below (a wave's a2 fragments stay in registers: 128 accumulator registers for all 256 couts at
    // once made the allocator spill beside conv2's 288 weight registers) ============
    bf16x8 b3f[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) b3f[j] = *reinterpret_cast<const bf16x8*>(stage + pxl * PIX + 16 * j + 8 * fhalf);
    lds_barrier_b();          // every wave has read its a2 fragments: the patch / stage bytes become the epilogue's staging area

    // ============ epilogue, 64 couts at a time: bn3 in fp32 -> LDS; + identity, ReLU, round; 16-byte stores ============
#pragma nounroll
    for (int qd = 0; qd < ((p.dbg & 16) ? 0 : 4); ++qd) {
      f32x16 acc3[2] = {};
#pragma unroll
      for (int m2 = 0; m2 < 2; ++m2)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const bf16x8 a = *reinterpret_cast<const bf16x8*>(w3s + ((2 * qd + m2) * 32 + q) * W3ROW + 16 * j + 8 * fhalf);
          acc3[m2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b3f[j], acc3[m2], 0, 0, 0);
        }
#pragma unroll
      for (int m2 = 0; m2 < 2; ++m2)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int cl = m2 * 32 + 8 * i + 4 * fhalf, co = 64 * qd + cl;
          const f32x4 s4 = *reinterpret_cast<const f32x4*>(s3 + co), b4 = *reinterpret_cast<const f32x4*>(b3 + co);
          f32x4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = add1(mul1(acc3[m2][4 * i + e], s4[e]), b4[e]);
          *reinterpret_cast<f32x4*>(fq + pxl * FROW + cl) = v;
        }
      if (DS) {
        // identity quarter = bn_ds(conv_ds(x)), rounded to bf16 as the separate launch stores it
        f32x16 accd[2] = {};
#pragma unroll
        for (int m2 = 0; m2 < 2; ++m2)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const bf16x8 a = *reinterpret_cast<const bf16x8*>(wdss + ((2 * qd + m2) * 32 + q) * W3ROW + 16 * j + 8 * fhalf);
            accd[m2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bx[j], accd[m2], 0, 0, 0);
          }
#pragma unroll
        for (int m2 = 0; m2 < 2; ++m2)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int cl = m2 * 32 + 8 * i + 4 * fhalf, co = 64 * qd + cl;
            const f32x4 s4 = *reinterpret_cast<const f32x4*>(sds + co), b4 = *reinterpret_cast<const f32x4*>(bds + co);
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = add1(mul1(accd[m2][4 * i + e], s4[e]), b4[e]);
            const u32x2 o = {pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
            *reinterpret_cast<u32x2*>(iq + pxl * PIX + cl) = o;
          }
      }
      lds_barrier_b();
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        const int idx = tid + i * BN_T;
        const int px = idx >> 3, ch = idx & 7;
        if (okc[i]) {
          const long long gp = gpc[i];
          const f32x4 f0 = *reinterpret_cast<const f32x4*>(fq + px * FROW + ch * 8);
          const f32x4 f1 = *reinterpret_cast<const f32x4*>(fq + px * FROW + ch * 8 + 4);
          u32x4 id = idc[i];
          if (DS) id = *reinterpret_cast<const u32x4*>(iq + px * PIX + ch * 8);
          float v[8] = {f0[0], f0[1], f0[2], f0[3], f1[0], f1[1], f1[2], f1[3]};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v[2 * e] = fmaxf(add1(v[2 * e], bflo(id[e])), 0.f);
            v[2 * e + 1] = fmaxf(add1(v[2 * e + 1], bfhi(id[e])), 0.f);
          }
          const u32x4 o = {pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7])};
          if (!(p.dbg & 4)) *reinterpret_cast<u32x4*>(p.out + gp * p.ld_out + 64 * qd + ch * 8) = o;
        }
      }
      if (qd < 3) load_id(qd + 1);        // (its registers are free again: in flight across the barrier, the next quarter's MFMAs and staging)
      lds_barrier_b();        // the staging area is free for the next quarter / the next tile's patch
    }
  }
}

template <int CIN, bool DS>
int launch_bneck(const BnK& k, hipStream_t st) {
  const size_t lds = (size_t)bn_lds_elems<CIN, DS>() * 2;
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute((const void*)bottleneck64_kernel<CIN, DS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
      dsl_set_error("dsl_bottleneck64: cannot reserve %zu bytes of LDS", lds);
      return -2;
    }
    attr = true;
  }
  // persistent workgroups, one per CU at most; DSL_BNECK_GRID caps the grid (a narrower prefix leaves CUs to the pass it runs beside)
  static const int cap = [] { const char* e = getenv("DSL_BNECK_GRID"); const int v = e ? atoi(e) : 256; return v >= 8 && v <= 256 ? v : 256; }();
  const unsigned grid = (unsigned)(k.tiles < cap ? k.tiles : cap);
  hipLaunchKernelGGL((bottleneck64_kernel<CIN, DS>), dim3(grid), dim3(BN_T), lds, st, k);
  return 0;
}

}  // namespace

extern "C" int dsl_bottleneck64(const dsl_bneck64_desc* d, void* stream) {
  DSL_CHECK(d && d->x && d->out && d->w1 && d->w2 && d->w3 && d->s1 && d->b1 && d->s2 && d->b2 && d->s3 && d->b3,
            "dsl_bottleneck64: null pointer");
  DSL_CHECK(d->cin == 64 || d->cin == 256, "dsl_bottleneck64: Cin = %d (64 or 256)", d->cin);
  const bool ds = d->wds != nullptr;
  DSL_CHECK(ds ? (d->cin == 64 && d->sds && d->bds) : d->cin == 256,
            "dsl_bottleneck64: Cin 64 needs the downsample convolution (wds, sds, bds), Cin 256 uses x as the identity");
  DSL_CHECK(d->n >= 1 && d->h >= 1 && d->w >= 1 && d->ld_x >= d->cin && d->ld_x % 8 == 0 && d->ld_out >= 256 && d->ld_out % 8 == 0,
            "dsl_bottleneck64: bad shape (n=%d h=%d w=%d ld_x=%d ld_out=%d)", d->n, d->h, d->w, d->ld_x, d->ld_out);
  BnK k;
  k.x = (const uint16_t*)d->x; k.out = (uint16_t*)d->out;
  k.w1 = (const uint16_t*)d->w1; k.w2 = (const uint16_t*)d->w2; k.w3 = (const uint16_t*)d->w3; k.wds = (const uint16_t*)d->wds;
  k.s1 = d->s1; k.b1 = d->b1; k.s2 = d->s2; k.b2 = d->b2; k.s3 = d->s3; k.b3 = d->b3; k.sds = d->sds; k.bds = d->bds;
  k.n = d->n; k.H = d->h; k.W = d->w; k.ld_x = d->ld_x; k.ld_out = d->ld_out;
  k.tiles_x = (d->w + TW - 1) / TW; k.tiles_y = (d->h + TH - 1) / TH;
  const long long tiles = (long long)d->n * k.tiles_x * k.tiles_y;
  DSL_CHECK(tiles < (1ll << 31), "dsl_bottleneck64: too many tiles");
  k.tiles = (int)tiles;
  { static const int dbg = [] { const char* e = getenv("DSL_BNECK_DBG"); return e ? atoi(e) : 0; }(); k.dbg = dbg; }
  hipStream_t st = (hipStream_t)stream;
  const int rc = ds ? launch_bneck<64, true>(k, st) : launch_bneck<256, false>(k, st);
  if (rc) return rc;
  DSL_LAUNCH_CHECK("bottleneck64_kernel");
  return 0;
}
