// Two back-to-back 1x1 convolutions as ONE launch: the expand conv of a bottleneck and the reduce conv of the next one
//   forward   out_b  = relu(bn3(W3 . a2_b) + identity)          a1_{b+1} = relu(bn1(W1 . out_b))      (resnet.py:262-301)
//   backward  g_in   = mask_xin . (W1^T . g_a1_{b+1} + g_out)    g_a2_b   = mask_a2 . (W3^T . g_in)
// Both have the shape  mid[m][4P] = epi1(A[m][P] . Wa[4P][P]^T),  out[m][P] = epi2(mid[m][4P] . Wb[P][4P]^T)  with P = 128 | 256.
// The wide tensor `mid` is still written (the next block's residual / the weight gradients need it) but never read back: a
// workgroup owns 64 pixels, produces mid in 64-channel chunks and feeds each chunk - rounded to bf16, exactly the values the
// second launch would load - straight into the second GEMM as a K slab, whose accumulators stay in registers.
// At 8 400 / 33 600 pixels the two separate launches are latency-bound (28 + 25 us in the step for 4.4 + 4.4 GFLOP); fused,
// the pair is bound by streaming the two weight matrices through the CU's 64 B/clk vector-memory path once per workgroup.
// Results are bit-identical to the two-launch path (same MFMA instruction, same K order, same bf16 rounding point).
//
// LDS (P = 256: 153 KB): A tile [P/64 slabs][64 px][128 B], Wa ring 2 x [P/64][64 ch][128 B], Wb slab [P ch][128 B],
// fp32 staging [64][64], bf16 chunk [64 px][128 B]; 128-byte rows with the 16-byte-chunk XOR swizzle of the conv kernels.
#include <stdlib.h>

#include "common.hpp"

namespace {

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
__device__ __attribute__((aligned(16))) unsigned int g_pair_zero[4] = {0u, 0u, 0u, 0u};

struct PairK {
  int m, lda, ldadd, ldm1, ldmid, ldm2, ldo, relu1, relu2;
  int dbg;       // ablation (DSL_PAIR_ABLATE): 1 no weight DMA after chunk 0, 2 no first-GEMM math, 4 no residual / mask / mid traffic, 8 no second GEMM, 16 no barriers 2/3
  const uint16_t* a;
  const uint16_t* wa;
  const uint16_t* wb;
  const float* scale1;
  const float* bias1;
  const float* scale2;
  const float* bias2;
  const uint16_t* addend;
  const uint16_t* mask1;
  const uint16_t* mask2;
  uint16_t* mid;
  uint16_t* out;
};

constexpr int PAIR_BM = 64, PAIR_NC = 64, PAIR_T = 512;

// MFMA fragment reads through inline asm: given a plain LDS load hipcc waits for every in-flight `global_load ... lds`
// (vmcnt(0)) first, which would put the weight DMA of the next chunk in front of this chunk's math.  The fragments are
// consumed only after an explicit lgkmcnt(0) that names them (see the weight-gradient kernel).
union FragU {
  u32x4 u;
  bf16x8 b;
};
__device__ __forceinline__ void lds_read_b128_asm(u32x4& v, unsigned addr) {
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_lds(u32x4 (&f)[N]) {
  static_assert(N == 8 || N == 12 || N == 16, "fragment counts of the pair kernel");
  if constexpr (N == 8)
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7]) : : "memory");
  else if constexpr (N == 12)
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7]),
                 "+v"(f[8]), "+v"(f[9]), "+v"(f[10]), "+v"(f[11]) : : "memory");
  else
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7]),
                 "+v"(f[8]), "+v"(f[9]), "+v"(f[10]), "+v"(f[11]), "+v"(f[12]), "+v"(f[13]), "+v"(f[14]), "+v"(f[15]) : : "memory");
  __builtin_amdgcn_sched_barrier(0);
}
constexpr int SCR_ROWB = PAIR_NC * 4 + 16;

template <int P>
__global__ __launch_bounds__(PAIR_T) void pair1x1_kernel(const PairK p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int SLABS = P / 64;                       // K slabs of the first GEMM
  constexpr int A_BYTES = SLABS * PAIR_BM * 128;      // = P * 128
  constexpr int WA_BYTES = SLABS * PAIR_NC * 128;     // one ring slot
  constexpr int WB_BYTES = P * 128;
  constexpr int OFF_WA = A_BYTES, OFF_WB = OFF_WA + 2 * WA_BYTES, OFF_SCR = OFF_WB + WB_BYTES;
  constexpr int OFF_OUTC = OFF_SCR + PAIR_BM * SCR_ROWB;
  constexpr int NCHUNK = 4 * P / PAIR_NC;
  constexpr int CT2 = P / 128;                        // 32-channel MFMA tiles per wave in the second GEMM
  constexpr int ROWB2 = P * 4 + 16;                   // final staging row
  static_assert(PAIR_BM * ROWB2 <= OFF_WB, "final staging reuses the A tile + Wa ring");

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m0 = blockIdx.x * PAIR_BM;
  const gptr_t zero = (gptr_t)g_pair_zero;
  const int lrow = lane >> 3;                          // row within an 8-row DMA instruction
  const int frow = lane & 31, fhalf = lane >> 5, fswz = (frow >> 1) & 7;

  // ---- DMA helpers: one wave instruction = 8 rows x 128 B; lane (row, slot) fetches the source chunk that belongs in `slot`
  auto dma_a = [&]() {
#pragma unroll
    for (int q = wave; q < SLABS * 8; q += 8) {
      const int slab = q >> 3, row = (q & 7) * 8 + lrow;
      const int chunk = (lane & 7) ^ ((row >> 1) & 7);
      const gptr_t g = (m0 + row < p.m) ? (gptr_t)(p.a + (long long)(m0 + row) * p.lda + slab * 64 + chunk * 8) : zero;
      __builtin_amdgcn_global_load_lds(g, (lptr_t)(smem + slab * (PAIR_BM * 128) + (q & 7) * 1024), 16, 0, 0);
    }
  };
  auto dma_wa = [&](int c, int slot) {
#pragma unroll
    for (int q = wave; q < SLABS * 8; q += 8) {
      const int slab = q >> 3, row = (q & 7) * 8 + lrow;
      const int chunk = (lane & 7) ^ ((row >> 1) & 7);
      const gptr_t g = (gptr_t)(p.wa + (long long)(c * PAIR_NC + row) * P + slab * 64 + chunk * 8);
      __builtin_amdgcn_global_load_lds(g, (lptr_t)(smem + OFF_WA + slot * WA_BYTES + slab * (PAIR_NC * 128) + (q & 7) * 1024), 16, 0, 0);
    }
  };
  auto dma_wb = [&](int c) {
#pragma unroll
    for (int q = wave; q < P / 8; q += 8) {
      const int row = q * 8 + lrow;
      const int chunk = (lane & 7) ^ ((row >> 1) & 7);
      const gptr_t g = (gptr_t)(p.wb + (long long)row * (4 * P) + c * PAIR_NC + chunk * 8);
      __builtin_amdgcn_global_load_lds(g, (lptr_t)(smem + OFF_WB + q * 1024), 16, 0, 0);
    }
  };
  const unsigned lds0 = (unsigned)(size_t)smem;        // low 32 bits of a flat LDS address = the LDS offset
  auto faddr = [&](int base, int row0, int kk) -> unsigned {
    return lds0 + base + (row0 + frow) * 128 + (((2 * kk + fhalf) ^ fswz) << 4);
  };

  f32x16 acc2[CT2];
#pragma unroll
  for (int t = 0; t < CT2; ++t)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc2[t][j] = 0.f;

  dma_a();
  dma_wa(0, 0);
  const int wm2 = wave >> 2, wn2 = wave & 3;           // second GEMM: 2 pixel halves x 4 channel quarters
  const int wm1 = (wave >> 1) & 1, wn1 = wave & 1;     // first GEMM (waves 0-3): 2 pixel halves x 2 channel halves of the chunk
  const int e_px = tid >> 3, e_cg = tid & 7;           // epilogue item of this thread: pixel, 8-channel group of the chunk
  for (int c = 0; c < NCHUNK; ++c) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                   // A tile and Wa(c) landed; second GEMM of chunk c-1 done everywhere
    // the residual / mask rows of this chunk's epilogue: fetched now (before the DMA instructions, so that waiting for them does
    // not wait for the DMA), consumed after the first GEMM
    const int e_n = c * PAIR_NC + e_cg * 8;
    const long long e_gp = m0 + e_px;
    u32x4 e_aa = {0u, 0u, 0u, 0u}, e_mm = {0u, 0u, 0u, 0u};
    f32x4 e_s0 = {1.f, 1.f, 1.f, 1.f}, e_s1 = e_s0, e_b0 = {0.f, 0.f, 0.f, 0.f}, e_b1 = e_b0;
    u32x4 e_out = {0u, 0u, 0u, 0u};
    if (e_gp < p.m && !(p.dbg & 4)) {
      if (p.addend) e_aa = *reinterpret_cast<const u32x4*>(p.addend + e_gp * p.ldadd + e_n);
      if (p.mask1) e_mm = *reinterpret_cast<const u32x4*>(p.mask1 + e_gp * p.ldm1 + e_n);
      if (p.scale1) {
        e_s0 = *reinterpret_cast<const f32x4*>(p.scale1 + e_n);
        e_s1 = *reinterpret_cast<const f32x4*>(p.scale1 + e_n + 4);
      }
      if (p.bias1) {
        e_b0 = *reinterpret_cast<const f32x4*>(p.bias1 + e_n);
        e_b1 = *reinterpret_cast<const f32x4*>(p.bias1 + e_n + 4);
      }
    }
    if (!(p.dbg & 1) || c == 0) {
      dma_wb(c);
      if (c + 1 < NCHUNK) dma_wa(c + 1, (c + 1) & 1);
    }
    if (wave < 4 && !(p.dbg & 2)) {
      f32x16 acc1;
#pragma unroll
      for (int j = 0; j < 16; ++j) acc1[j] = 0.f;
      const int wbase = OFF_WA + (c & 1) * WA_BYTES;
#pragma unroll
      for (int s2 = 0; s2 < SLABS; s2 += 2) {          // 8 K-steps (two slabs) per batch of fragment reads
        u32x4 f[16];
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) {
            lds_read_b128_asm(f[(s * 4 + kk) * 2], faddr(wbase + (s2 + s) * (PAIR_NC * 128), wn1 * 32, kk));
            lds_read_b128_asm(f[(s * 4 + kk) * 2 + 1], faddr((s2 + s) * (PAIR_BM * 128), wm1 * 32, kk));
          }
        wait_lds<16>(f);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          FragU fa, fb;
          fa.u = f[2 * i];
          fb.u = f[2 * i + 1];
          acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa.b, fb.b, acc1, 0, 0, 0);
        }
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int col = wn1 * 32 + 8 * g + 4 * fhalf;
        *reinterpret_cast<f32x4*>(smem + OFF_SCR + (wm1 * 32 + frow) * SCR_ROWB + col * 4) =
            f32x4{acc1[4 * g], acc1[4 * g + 1], acc1[4 * g + 2], acc1[4 * g + 3]};
      }
    }
    __syncthreads();
    {                                                  // epilogue 1: one (pixel, 8 channels) item per thread
      const int n = e_n;
      const long long gp = e_gp;
      const f32x4 lo = *reinterpret_cast<const f32x4*>(smem + OFF_SCR + e_px * SCR_ROWB + e_cg * 32);
      const f32x4 hi = *reinterpret_cast<const f32x4*>(smem + OFF_SCR + e_px * SCR_ROWB + e_cg * 32 + 16);
      float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      u32x4 o = {0u, 0u, 0u, 0u};
      if (gp < p.m) {
        if (p.scale1) {
#pragma unroll
          for (int e = 0; e < 4; ++e) { v[e] *= e_s0[e]; v[4 + e] *= e_s1[e]; }
        }
        if (p.bias1) {
#pragma unroll
          for (int e = 0; e < 4; ++e) { v[e] += e_b0[e]; v[4 + e] += e_b1[e]; }
        }
        if (p.addend) {
          const u32x4 aa = e_aa;
#pragma unroll
          for (int e = 0; e < 4; ++e) { v[2 * e] += bflo(aa[e]); v[2 * e + 1] += bfhi(aa[e]); }
        }
        if (p.mask1) {
          const u32x4 mm = e_mm;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v[2 * e] *= bflo(mm[e]) > 0.f ? 1.f : 0.f;
            v[2 * e + 1] *= bfhi(mm[e]) > 0.f ? 1.f : 0.f;
          }
        }
        if (p.relu1) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        o = u32x4{pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7])};
      }
      // the chunk as the second GEMM's K slab (B operand rows = pixels), same swizzle as a DMA'd tile
      *reinterpret_cast<u32x4*>(smem + OFF_OUTC + e_px * 128 + ((e_cg ^ ((e_px >> 1) & 7)) << 4)) = o;
      e_out = o;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // Wb(c) landed (this wave's part); nothing else is in flight here
    __syncthreads();
    // the HBM copy of the chunk leaves now: its write acknowledgement is waited for at the top of the next chunk, behind
    // the second GEMM, not in front of it
    if (e_gp < p.m && !(p.dbg & 4)) *reinterpret_cast<u32x4*>(p.mid + e_gp * p.ldmid + e_n) = e_out;
    if (!(p.dbg & 8)) {
      u32x4 f[4 * (1 + CT2)];
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        lds_read_b128_asm(f[kk * (1 + CT2)], faddr(OFF_OUTC, wm2 * 32, kk));
#pragma unroll
        for (int t = 0; t < CT2; ++t) lds_read_b128_asm(f[kk * (1 + CT2) + 1 + t], faddr(OFF_WB, wn2 * (P / 4) + t * 32, kk));
      }
      wait_lds<4 * (1 + CT2)>(f);
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        FragU fb;
        fb.u = f[kk * (1 + CT2)];
#pragma unroll
        for (int t = 0; t < CT2; ++t) {
          FragU fa;
          fa.u = f[kk * (1 + CT2) + 1 + t];
          acc2[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa.b, fb.b, acc2[t], 0, 0, 0);
        }
      }
    }
  }
  // ---- epilogue 2: stage [64 px][P] fp32 over the A tile / Wa ring, then (pixel, 8 channels) items
  __syncthreads();
#pragma unroll
  for (int t = 0; t < CT2; ++t)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int col = wn2 * (P / 4) + t * 32 + 8 * g + 4 * fhalf;
      *reinterpret_cast<f32x4*>(smem + (wm2 * 32 + frow) * ROWB2 + col * 4) =
          f32x4{acc2[t][4 * g], acc2[t][4 * g + 1], acc2[t][4 * g + 2], acc2[t][4 * g + 3]};
    }
  __syncthreads();
  constexpr int GPR = P / 8;
  for (int id = tid; id < PAIR_BM * GPR; id += PAIR_T) {
    const int px = id / GPR, cg = id - px * GPR;
    const long long gp = m0 + px;
    if (gp >= p.m) continue;
    const int n = cg * 8;
    const f32x4 lo = *reinterpret_cast<const f32x4*>(smem + px * ROWB2 + cg * 32);
    const f32x4 hi = *reinterpret_cast<const f32x4*>(smem + px * ROWB2 + cg * 32 + 16);
    float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    if (p.scale2) {
      const f32x4 s0 = *reinterpret_cast<const f32x4*>(p.scale2 + n), s1 = *reinterpret_cast<const f32x4*>(p.scale2 + n + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[e] *= s0[e]; v[4 + e] *= s1[e]; }
    }
    if (p.bias2) {
      const f32x4 b0 = *reinterpret_cast<const f32x4*>(p.bias2 + n), b1 = *reinterpret_cast<const f32x4*>(p.bias2 + n + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[e] += b0[e]; v[4 + e] += b1[e]; }
    }
    if (p.mask2) {
      const u32x4 mm = *reinterpret_cast<const u32x4*>(p.mask2 + gp * p.ldm2 + n);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[2 * e] *= bflo(mm[e]) > 0.f ? 1.f : 0.f;
        v[2 * e + 1] *= bfhi(mm[e]) > 0.f ? 1.f : 0.f;
      }
    }
    if (p.relu2) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
    }
    *reinterpret_cast<u32x4*>(p.out + gp * p.ldo + n) =
        u32x4{pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7])};
  }
}

}  // namespace

extern "C" int dsl_conv1x1_pair(const dsl_pair_desc* d, void* stream) {
  DSL_CHECK(d != nullptr, "dsl_conv1x1_pair: null descriptor");
  DSL_CHECK(d->p == 128 || d->p == 256, "dsl_conv1x1_pair: inner channels %d (128 or 256)", d->p);
  DSL_CHECK(d->m > 0 && d->a && d->wa && d->wb && d->mid && d->out, "dsl_conv1x1_pair: null tensor pointer / empty");
  DSL_CHECK(d->lda >= d->p && d->lda % 8 == 0 && d->ldmid >= 4 * d->p && d->ldmid % 8 == 0 && d->ldo >= d->p && d->ldo % 8 == 0,
            "dsl_conv1x1_pair: row strides must cover the channels and be multiples of 8");
  DSL_CHECK((!d->addend || d->ldadd % 8 == 0) && (!d->mask1 || d->ldm1 % 8 == 0) && (!d->mask2 || d->ldm2 % 8 == 0),
            "dsl_conv1x1_pair: addend / mask row strides must be multiples of 8");
  PairK k;
  k.m = d->m; k.lda = d->lda; k.ldadd = d->ldadd; k.ldm1 = d->ldm1; k.ldmid = d->ldmid; k.ldm2 = d->ldm2; k.ldo = d->ldo;
  k.relu1 = d->relu1; k.relu2 = d->relu2;
  { static const int dbg = [] { const char* e = getenv("DSL_PAIR_ABLATE"); return e ? atoi(e) : 0; }(); k.dbg = dbg; }
  k.a = (const uint16_t*)d->a; k.wa = (const uint16_t*)d->wa; k.wb = (const uint16_t*)d->wb;
  k.scale1 = d->scale1; k.bias1 = d->bias1; k.scale2 = d->scale2; k.bias2 = d->bias2;
  k.addend = (const uint16_t*)d->addend; k.mask1 = (const uint16_t*)d->mask1; k.mask2 = (const uint16_t*)d->mask2;
  k.mid = (uint16_t*)d->mid; k.out = (uint16_t*)d->out;
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((d->m + PAIR_BM - 1) / PAIR_BM);
  const double flops = 2.0 * 2.0 * d->m * 4.0 * d->p * d->p;
  const double bytes = (double)d->m * (d->p * 2.0 * 2 + 4.0 * d->p * 2.0 * (1 + (d->addend ? 1 : 0) + (d->mask1 ? 1 : 0)) + (d->mask2 ? d->p * 2.0 : 0.0)) +
                       2.0 * 4.0 * d->p * d->p * 2.0;
  const int prof = dsl_prof_active() ? dsl_prof_begin(2, flops, st, bytes) : -1;
  if (d->p == 128) {
    const size_t lds = (size_t)128 * 128 * 4 + PAIR_BM * SCR_ROWB + PAIR_BM * 128;
    static bool a = false;
    if (!a) { hipFuncSetAttribute((const void*)pair1x1_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); a = true; }
    hipLaunchKernelGGL((pair1x1_kernel<128>), grid, dim3(PAIR_T), lds, st, k);
  } else {
    const size_t lds = (size_t)256 * 128 * 4 + PAIR_BM * SCR_ROWB + PAIR_BM * 128;
    static bool a = false;
    if (!a) { hipFuncSetAttribute((const void*)pair1x1_kernel<256>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); a = true; }
    hipLaunchKernelGGL((pair1x1_kernel<256>), grid, dim3(PAIR_T), lds, st, k);
  }
  dsl_prof_end(prof, st);
  DSL_LAUNCH_CHECK("pair1x1_kernel");
  return 0;
}
