// Implicit-GEMM convolution kernels for gfx950 (MI355X): forward / data-gradient ("gather GEMM")
// and weight-gradient (split-K over pixels with LDS transpose reads).
//
// Replaces F.conv2d (+ its autograd backward) at mmdet/models/backbones/resnet.py:262-301,598-645,
// mmdet/models/necks/fpn.py:150-202, mmdet/models/dense_heads/anchor_free_head.py:197-217 and
// fcos_head.py:154-156 of the reference.
//
// Layout: activations NHWC bf16; weights bf16 [CoutPad][kh][kw][Cin] (K contiguous).
// GEMM roles are swapped w.r.t. the textbook: A = weights (rows = cout), B = pixels, so that in
// the v_mfma_f32_32x32x16_bf16 result each lane owns ONE pixel and 4-channel runs of couts
// -> the epilogue does per-pixel index math once per lane and 8/16-byte channel-contiguous
// stores into NHWC.
#include <stdlib.h>
#include <algorithm>

#include <mutex>
#include <type_traits>
#include <vector>

#include "common.hpp"

namespace {

constexpr int BPX = 128;   // pixels per workgroup tile
constexpr int BK = 64;     // K elements per LDS stage

struct ConvK {
  int nseg, n;
  int gh[DSL_MAX_SEG], gw[DSL_MAX_SEG], sh[DSL_MAX_SEG], sw[DSL_MAX_SEG];
  int dh[DSL_MAX_SEG], dw[DSL_MAX_SEG], ah[DSL_MAX_SEG], aw[DSL_MAX_SEG];
  int pxstart[DSL_MAX_SEG + 1];
  long long soff[DSL_MAX_SEG], doff[DSL_MAX_SEG], aoff[DSL_MAX_SEG];   // segment starts, in pixels
  int cs, cd, ldd, lda, ldm, kh, kw, stride, pad, mode, os, flags;
  int lds;                            // source pixel stride in elements (>= cs: the source may be a channel slice of wider rows)
  int ktiles, kc;
  int ident;                          // 1: destination pixel index == compute-grid pixel index (os 1, same sizes)
  int addfast;                        // 1: the addend tile can be DMA'd into LDS (ident, 16-byte aligned rows, 32-bit offsets)
  int dbg;                            // ablation knobs, compiled in only by tools/build_ablate.sh (-DDSL_ABLATE_BUILD; DSL_ABLATE env)
  int splits, kt_per_split, cd_pad;   // split-K over K tiles (v2 kernel): fp32 partials -> ws, then conv_splitk_epilogue_kernel
  int gx, gy, xcd_chunk;              // v3: tile grid (cout tiles, pixel tiles) and tiles per XCD of the 1-D XCD-aware launch
  float* ws;
  long long wrow;
  const uint16_t* src;
  const uint16_t* wgt;
  void* dst;
  const float* scale;
  const float* bias;
  const uint16_t* addend;
  const uint16_t* mask;
  float* gnws;                        // GroupNorm statistics records of the output (dsl_conv_desc.gn_ws), or null
  const uint16_t* gnx;                // not null: the output is dY of a GroupNorm + ReLU whose input was gnx; gnws takes the BACKWARD
  const float* gngamma;               // records (what gn_bwd_reduce_kernel computes from dY and x)
  const float* gnbeta;
  const float* gnstats;               // [(segment, image)][cd / 8][mean, rstd] of the forward pass
};

__device__ __forceinline__ u32x4 relu_bf16x8(u32x4 v) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    uint32_t w = v[i];
    uint32_t neg = (w >> 15) & 0x00010001u;       // sign bits of the two halves
    v[i] = w & ~(neg * 0xffffu);
  }
  return v;
}

struct PixRow {
  long long base;   // pixel index of (seg, img, 0, 0) in the source tensor
  int y, x, sh, sw;
  bool ok;
};

// Workgroup barrier that orders LDS traffic only: __syncthreads() also drains vmcnt, i.e. makes every wave wait until its global
// STORES have been acknowledged - in an epilogue that alternates "stage a slab in LDS" and "store it" that serialises the store
// latency (1 - 2 us under load) once per slab.  The stores need no ordering against the LDS reuse: their data left the registers
// at issue.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

__device__ __forceinline__ void decode_pixel(const ConvK& p, int gp, int& seg, int& img, int& y, int& x) {
  seg = 0;
#pragma unroll
  for (int s = 1; s < DSL_MAX_SEG; ++s)
    if (s < p.nseg && gp >= p.pxstart[s]) seg = s;
  const int q = gp - p.pxstart[seg];
  const int hw = p.gh[seg] * p.gw[seg];
  img = q / hw;
  const int rem = q - img * hw;
  y = rem / p.gw[seg];
  x = rem - y * p.gw[seg];
}


// Epilogue for 4 consecutive output channels of one pixel (shared by every conv kernel).
__device__ __forceinline__ void conv_epilogue4(const ConvK& p, long long dpix, long long apix, int co, float v[4]) {
  const bool out_f32 = (p.flags & DSL_CONV_OUT_F32) != 0;
  const bool relu_out = (p.flags & DSL_CONV_RELU_OUT) != 0;
  const bool mask_first = (p.flags & DSL_CONV_MASK_FIRST) != 0 && p.mask != nullptr;
  const bool mask_last = (p.flags & DSL_CONV_MASK_LAST) != 0 && p.mask != nullptr;
  const bool has_mask = mask_first || mask_last;
  if (co + 3 < p.cd) {
    if (p.scale) {
      const f32x4 s4 = *reinterpret_cast<const f32x4*>(p.scale + co);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] *= s4[e];
    }
    if (p.bias) {
      const f32x4 b4 = *reinterpret_cast<const f32x4*>(p.bias + co);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] += b4[e];
    }
    float m[4] = {1.f, 1.f, 1.f, 1.f};
    if (has_mask) {
      const u32x2 mm = *reinterpret_cast<const u32x2*>(p.mask + dpix * p.ldm + co);
      m[0] = bflo(mm[0]) > 0.f ? 1.f : 0.f;
      m[1] = bfhi(mm[0]) > 0.f ? 1.f : 0.f;
      m[2] = bflo(mm[1]) > 0.f ? 1.f : 0.f;
      m[3] = bfhi(mm[1]) > 0.f ? 1.f : 0.f;
    }
    if (mask_first) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] *= m[e];
    }
    if (p.addend) {
      const u32x2 aa = *reinterpret_cast<const u32x2*>(p.addend + apix * p.lda + co);
      v[0] += bflo(aa[0]);
      v[1] += bfhi(aa[0]);
      v[2] += bflo(aa[1]);
      v[3] += bfhi(aa[1]);
    }
    if (mask_last) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] *= m[e];
    }
    if (relu_out) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
    }
    if (out_f32) {
      f32x4 o = {v[0], v[1], v[2], v[3]};
      *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.dst) + dpix * p.ldd + co) = o;
    } else {
      u32x2 o = {pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
      *reinterpret_cast<u32x2*>(reinterpret_cast<uint16_t*>(p.dst) + dpix * p.ldd + co) = o;
    }
  } else {   // ragged channel tail (e.g. conv_reg+centerness = 5 channels): element-wise
    for (int e = 0; e < 4 && co + e < p.cd; ++e) {
      float t = v[e];
      if (p.scale) t *= p.scale[co + e];
      if (p.bias) t += p.bias[co + e];
      float mk = 1.f;
      if (has_mask) mk = bf2f(p.mask[dpix * p.ldm + co + e]) > 0.f ? 1.f : 0.f;
      if (mask_first) t *= mk;
      if (p.addend) t += bf2f(p.addend[apix * p.lda + co + e]);
      if (mask_last) t *= mk;
      if (relu_out) t = fmaxf(t, 0.f);
      if (out_f32)
        reinterpret_cast<float*>(p.dst)[dpix * p.ldd + co + e] = t;
      else
        reinterpret_cast<uint16_t*>(p.dst)[dpix * p.ldd + co + e] = f2bf(t);
    }
  }
}

// Epilogue for 8 consecutive channels of one pixel: 16-byte loads / stores (used by the LDS-staged epilogue
// of the v2 kernel).  Falls back to two 4-channel epilogues on a ragged channel tail.
// the per-channel scale / bias of 8 consecutive output channels, loaded once per thread where a thread's channel group is fixed
struct Affine8 {
  f32x4 s0, s1, b0, b1;
  bool full;                               // co + 7 < cd: the vector path applies
};
__device__ __forceinline__ Affine8 conv_affine8(const ConvK& p, int co) {
  Affine8 a;
  a.full = co + 7 < p.cd;
  a.s0 = a.s1 = f32x4{1.f, 1.f, 1.f, 1.f};
  a.b0 = a.b1 = f32x4{0.f, 0.f, 0.f, 0.f};
  if (a.full) {
    if (p.scale) {
      a.s0 = *reinterpret_cast<const f32x4*>(p.scale + co);
      a.s1 = *reinterpret_cast<const f32x4*>(p.scale + co + 4);
    }
    if (p.bias) {
      a.b0 = *reinterpret_cast<const f32x4*>(p.bias + co);
      a.b1 = *reinterpret_cast<const f32x4*>(p.bias + co + 4);
    }
  }
  return a;
}
__device__ __forceinline__ void conv_epilogue8a(const ConvK& p, long long dpix, long long apix, int co, float v[8], const Affine8& a);
__device__ __forceinline__ void conv_epilogue8(const ConvK& p, long long dpix, long long apix, int co, float v[8]) {
  conv_epilogue8a(p, dpix, apix, co, v, conv_affine8(p, co));
}
__device__ __forceinline__ void conv_epilogue8a(const ConvK& p, long long dpix, long long apix, int co, float v[8], const Affine8& a) {
  if (!a.full) {
    conv_epilogue4(p, dpix, apix, co, v);
    if (co + 4 < p.cd) conv_epilogue4(p, dpix, apix, co + 4, v + 4);
    return;
  }
  const bool mask_first = (p.flags & DSL_CONV_MASK_FIRST) != 0 && p.mask != nullptr;
  const bool mask_last = (p.flags & DSL_CONV_MASK_LAST) != 0 && p.mask != nullptr;
  if (p.scale) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      v[e] *= a.s0[e];
      v[4 + e] *= a.s1[e];
    }
  }
  if (p.bias) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      v[e] += a.b0[e];
      v[4 + e] += a.b1[e];
    }
  }
  float m[8] = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f};
  if (mask_first || mask_last) {
    const u32x4 mm = *reinterpret_cast<const u32x4*>(p.mask + dpix * p.ldm + co);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      m[2 * e] = bflo(mm[e]) > 0.f ? 1.f : 0.f;
      m[2 * e + 1] = bfhi(mm[e]) > 0.f ? 1.f : 0.f;
    }
  }
  if (mask_first) {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] *= m[e];
  }
  if (p.addend) {
    const u32x4 aa = *reinterpret_cast<const u32x4*>(p.addend + apix * p.lda + co);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      v[2 * e] += bflo(aa[e]);
      v[2 * e + 1] += bfhi(aa[e]);
    }
  }
  if (mask_last) {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] *= m[e];
  }
  if (p.flags & DSL_CONV_RELU_OUT) {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
  }
  if (p.flags & DSL_CONV_OUT_F32) {
    float* o = reinterpret_cast<float*>(p.dst) + dpix * p.ldd + co;
    *reinterpret_cast<f32x4*>(o) = f32x4{v[0], v[1], v[2], v[3]};
    *reinterpret_cast<f32x4*>(o + 4) = f32x4{v[4], v[5], v[6], v[7]};
  } else {
    u32x4 o = {pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7])};
    *reinterpret_cast<u32x4*>(reinterpret_cast<uint16_t*>(p.dst) + dpix * p.ldd + co) = o;
  }
}

// single-rounding multiply / add for epilogues that must round like the general path's separate "*= scale" and "+= bias" steps
#pragma clang fp contract(off)
__device__ __forceinline__ float mul_nc(float a, float b) { return a * b; }
__device__ __forceinline__ float add_nc(float a, float b) { return a + b; }
#pragma clang fp contract(fast)

// The common case of conv_epilogue8a with everything loop-invariant taken out: destination pixel == grid pixel (ident), 8 whole
// channels, bf16 output, the addend on the destination's own pixel grid.  Same operations in the same order (scale, bias,
// mask-first, addend, mask-last, ReLU, round) - only the flag tests, the 64-bit index arithmetic and the pixel decode of the
// general path are gone: the staged epilogue's item loop was ISSUE-bound on them (~190 VALU instructions and ~1 500 cycles per
// 16-byte store, tools/trace_conv.py: 5 700 - 7 100 cycles per 64-pixel slab against ~1 000 for the stores themselves).
struct EpiFast {
  bool on;                                 // uniform: the launch qualifies
  bool has_scale, has_bias, mask_first, mask_last, has_add, relu;
  uint16_t* dst;                           // + co already applied
  const uint16_t* add;
  const uint16_t* mask;
  int ldd, lda, ldm;
};
__device__ __forceinline__ EpiFast conv_epi_fast(const ConvK& p, int co) {
  EpiFast e;
  e.on = p.ident && !(p.flags & (DSL_CONV_OUT_F32 | DSL_CONV_ADD_UPSAMPLE));
  e.has_scale = p.scale != nullptr;
  e.has_bias = p.bias != nullptr;
  e.mask_first = (p.flags & DSL_CONV_MASK_FIRST) != 0 && p.mask != nullptr;
  e.mask_last = (p.flags & DSL_CONV_MASK_LAST) != 0 && p.mask != nullptr;
  e.has_add = p.addend != nullptr;
  e.relu = (p.flags & DSL_CONV_RELU_OUT) != 0;
  e.dst = reinterpret_cast<uint16_t*>(p.dst) + co;
  e.add = p.addend + co;
  e.mask = p.mask + co;
  e.ldd = p.ldd; e.lda = p.lda; e.ldm = p.ldm;
  return e;
}
__device__ __forceinline__ void conv_epilogue8_fast(const EpiFast& e, const Affine8& a, int gp, f32x4 lo, f32x4 hi) {
  float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  if (e.has_scale) {
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[i] *= a.s0[i]; v[4 + i] *= a.s1[i]; }
  }
  if (e.has_bias) {
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[i] += a.b0[i]; v[4 + i] += a.b1[i]; }
  }
  float m[8] = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f};
  if (e.mask_first || e.mask_last) {
    const u32x4 mm = *reinterpret_cast<const u32x4*>(e.mask + (long long)gp * e.ldm);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      m[2 * i] = bflo(mm[i]) > 0.f ? 1.f : 0.f;
      m[2 * i + 1] = bfhi(mm[i]) > 0.f ? 1.f : 0.f;
    }
  }
  if (e.mask_first) {
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] *= m[i];
  }
  if (e.has_add) {
    const u32x4 aa = *reinterpret_cast<const u32x4*>(e.add + (long long)gp * e.lda);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      v[2 * i] += bflo(aa[i]);
      v[2 * i + 1] += bfhi(aa[i]);
    }
  }
  if (e.mask_last) {
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] *= m[i];
  }
  if (e.relu) {
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = fmaxf(v[i], 0.f);
  }
  const u32x4 o = {pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7])};
  *reinterpret_cast<u32x4*>(e.dst + (long long)gp * e.ldd) = o;
}

__device__ __forceinline__ void conv_out_index(const ConvK& p, int gp, long long& dpix, long long& apix) {
  if (p.ident) {           // common case: no integer divisions in the epilogue
    dpix = gp;
    apix = gp;
    return;
  }
  int seg, img, y, x;
  decode_pixel(p, gp, seg, img, y, x);
  const int oy = y * p.os, ox = x * p.os;
  dpix = p.doff[seg] + ((long long)img * p.dh[seg] + oy) * p.dw[seg] + ox;
  apix = dpix;
  if (p.flags & DSL_CONV_ADD_UPSAMPLE) {
    const int ay = (oy * p.ah[seg]) / p.dh[seg], ax = (ox * p.aw[seg]) / p.dw[seg];
    apix = p.aoff[seg] + ((long long)img * p.ah[seg] + ay) * p.aw[seg] + ax;
  }
}

// split-K second pass: sum the fp32 partial tiles and run the epilogue
__global__ __launch_bounds__(256) void conv_splitk_epilogue_kernel(const ConvK p) {
  const int totpx = p.pxstart[p.nseg];
  const int c4 = p.cd_pad / 4;
  const long long total = (long long)totpx * c4;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int gp = (int)(i / c4);
    const int co = (int)(i - (long long)gp * c4) * 4;
    if (co >= p.cd) continue;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int sp = 0; sp < p.splits; ++sp)
      s += *reinterpret_cast<const f32x4*>(p.ws + ((long long)sp * totpx + gp) * p.cd_pad + co);
    long long dpix, apix;
    conv_out_index(p, gp, dpix, apix);
    float v[4] = {s[0], s[1], s[2], s[3]};
    conv_epilogue4(p, dpix, apix, co, v);
  }
}

template <int BCO, bool SMALLC>
__global__ __launch_bounds__(256) void conv_gemm_kernel(const ConvK p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int TILE_W = BCO * BK * 2;
  constexpr int TILE_X = BPX * BK * 2;
  constexpr int STAGE = TILE_W + TILE_X;
  constexpr int WM = BCO / 2;      // couts per wave
  constexpr int CT = WM / 32;      // 32-wide cout tiles per wave
  constexpr int WPASS = BCO / 32;  // weight rows handled per thread

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wave_co = wave >> 1, wave_px = wave & 1;
  const int co0 = blockIdx.x * BCO;
  const int px0 = blockIdx.y * BPX;
  const int totpx = p.pxstart[p.nseg];
  const int lrow = tid >> 3, chunk = tid & 7;

  // ---- per-thread pixel rows of the gather operand (constant over the K loop) ----
  PixRow pr[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int gp = px0 + lrow + 32 * i;
    pr[i].ok = gp < totpx;
    int seg = 0, img = 0, y = 0, x = 0;
    if (pr[i].ok) decode_pixel(p, gp, seg, img, y, x);
    pr[i].sh = p.sh[seg];
    pr[i].sw = p.sw[seg];
    pr[i].base = p.soff[seg] + (long long)img * pr[i].sh * pr[i].sw;
    pr[i].y = y;
    pr[i].x = x;
  }
  const uint16_t* wbase = p.wgt + (long long)(co0 + lrow) * p.wrow + chunk * 8;

  u32x4 rw[WPASS], rx[4];
  int tap_r = 0, tap_s = 0, cidx = 0;   // uniform K-walk state (non-SMALLC)

  auto gload = [&](int kt) {
    int r, s, coff;
    bool tapok = true;
    if (SMALLC) {
      const int tap = kt * 8 + chunk;       // one tap per 16-byte chunk (8 channels)
      r = tap / p.kw;
      s = tap - r * p.kw;
      tapok = tap < p.kh * p.kw;
      coff = 0;
    } else {
      r = tap_r;
      s = tap_s;
      coff = cidx * 64 + chunk * 8;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int sy, sx;
      bool ok = pr[i].ok && tapok;
      if (p.mode == 0) {
        sy = pr[i].y * p.stride + r - p.pad;
        sx = pr[i].x * p.stride + s - p.pad;
      } else {
        const int ty = pr[i].y + p.pad - r, tx = pr[i].x + p.pad - s;
        if (p.stride == 1) {
          sy = ty;
          sx = tx;
        } else {
          ok = ok && ty >= 0 && tx >= 0 && (ty % p.stride) == 0 && (tx % p.stride) == 0;
          sy = ty / p.stride;
          sx = tx / p.stride;
        }
      }
      ok = ok && (unsigned)sy < (unsigned)pr[i].sh && (unsigned)sx < (unsigned)pr[i].sw;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (ok) {
        const long long pix = pr[i].base + (long long)sy * pr[i].sw + sx;
        v = *reinterpret_cast<const u32x4*>(p.src + pix * p.lds + coff);
      }
      rx[i] = v;
    }
#pragma unroll
    for (int i = 0; i < WPASS; ++i)
      rw[i] = *reinterpret_cast<const u32x4*>(wbase + (long long)(32 * i) * p.wrow + (long long)kt * BK);
    if (!SMALLC) {   // advance the uniform K walk
      if (++cidx == p.kc) {
        cidx = 0;
        if (++tap_s == p.kw) {
          tap_s = 0;
          ++tap_r;
        }
      }
    }
  };

  const int swz_w = (chunk ^ ((lrow >> 1) & 7)) << 4;   // (row>>1)&7 is invariant under row += 32
  auto lds_store = [&](int buf) {
    unsigned char* base = smem + buf * STAGE;
    const bool relu_in = (p.flags & DSL_CONV_RELU_IN) != 0;
#pragma unroll
    for (int i = 0; i < WPASS; ++i)
      *reinterpret_cast<u32x4*>(base + (lrow + 32 * i) * 128 + swz_w) = rw[i];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      u32x4 v = rx[i];
      if (relu_in) v = relu_bf16x8(v);
      *reinterpret_cast<u32x4*>(base + TILE_W + (lrow + 32 * i) * 128 + swz_w) = v;
    }
  };

  f32x16 acc[CT][2];
#pragma unroll
  for (int a = 0; a < CT; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[a][b][j] = 0.f;

  const int frow = lane & 31, fhalf = lane >> 5;
  const int fswz = (frow >> 1) & 7;
  auto compute = [&](int buf) {
    const unsigned char* base = smem + buf * STAGE;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int coff = ((2 * kk + fhalf) ^ fswz) << 4;
      bf16x8 a[CT], b[2];
#pragma unroll
      for (int ct = 0; ct < CT; ++ct)
        a[ct] = *reinterpret_cast<const bf16x8*>(base + (wave_co * WM + ct * 32 + frow) * 128 + coff);
#pragma unroll
      for (int pt = 0; pt < 2; ++pt)
        b[pt] = *reinterpret_cast<const bf16x8*>(base + TILE_W + (wave_px * 64 + pt * 32 + frow) * 128 + coff);
#pragma unroll
      for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int pt = 0; pt < 2; ++pt)
          acc[ct][pt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ct], b[pt], acc[ct][pt], 0, 0, 0);
    }
  };

  // ---- main loop: register-staged double buffering, one barrier per K tile ----
  gload(0);
  lds_store(0);
  __syncthreads();
  for (int kt = 0; kt < p.ktiles; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < p.ktiles) gload(kt + 1);
    compute(cur);
    if (kt + 1 < p.ktiles) lds_store(cur ^ 1);
    __syncthreads();
  }

  // ---- epilogue ----
#pragma unroll
  for (int pt = 0; pt < 2; ++pt) {
    const int gp = px0 + wave_px * 64 + pt * 32 + frow;
    if (gp >= totpx) continue;
    long long dpix, apix;
    conv_out_index(p, gp, dpix, apix);
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int co = co0 + wave_co * WM + ct * 32 + 8 * g + 4 * fhalf;
        if (co >= p.cd) continue;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[ct][pt][4 * g + e];
        conv_epilogue4(p, dpix, apix, co, v);
      }
    }
  }
}

// ================================================================================================
// v2 forward / data-gradient kernel: bigger tiles, 8 waves, operands DMA'd straight into LDS
// (global_load_lds_dwordx4, 1 KB per wave-instruction = 8 rows x 128 B), the XOR swizzle applied through
// the per-lane SOURCE address (the LDS image of a DMA is lane-linear), out-of-image taps read a zero line.
// One barrier per K tile; the next tile's DMA is in flight while the current one feeds the MFMAs.
// ================================================================================================
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
__device__ __attribute__((aligned(16))) unsigned int g_zero_line[4] = {0u, 0u, 0u, 0u};

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int BCO, int BPX, int WCO, int WPX, int NST>
__global__ __launch_bounds__(64 * WCO * WPX) void conv_glds_kernel(const ConvK p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int T = 64 * WCO * WPX;
  constexpr int RPP = T / 8;                 // tile rows filled per pass (8 lanes per 128-byte row)
  constexpr int WPASS = BCO / RPP, XPASS = BPX / RPP;
  constexpr int TILE_W = BCO * 128;
  constexpr int STAGE = (BCO + BPX) * 128;
  constexpr int PT = BPX / WPX / 32;         // 32-pixel MFMA tiles per wave
  static_assert(BCO / WCO == 64, "each wave owns 64 couts");
  static_assert(BCO % RPP == 0 && BPX % RPP == 0 && (BPX / WPX) % 32 == 0, "tile/thread mismatch");

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wave_co = wave / WPX, wave_px = wave % WPX;
  const int co0 = blockIdx.x * BCO;
  const int px0 = blockIdx.y * BPX;
  const int totpx = p.pxstart[p.nseg];
  const int lrow = tid >> 3;
  const int chunk = (tid & 7) ^ ((tid >> 4) & 7);     // source chunk that belongs in LDS slot (tid & 7) of this row

  int r_base[XPASS], r_yx[XPASS], r_hw[XPASS];
#pragma unroll
  for (int i = 0; i < XPASS; ++i) {
    const int gp = px0 + lrow + RPP * i;
    int seg = 0, img = 0, y = 0, x = 0;
    const bool ok = gp < totpx;
    if (ok) decode_pixel(p, gp, seg, img, y, x);
    r_hw[i] = (p.sh[seg] << 16) | p.sw[seg];
    r_base[i] = (int)(p.soff[seg] + (long long)img * p.sh[seg] * p.sw[seg]);
    r_yx[i] = ok ? ((y << 16) | x) : -1;
  }
  const uint16_t* wbase = p.wgt + (long long)(co0 + lrow) * p.wrow + chunk * 8;
  const gptr_t zero = (gptr_t)g_zero_line;

  const int kt0 = blockIdx.z * p.kt_per_split;
  const int kt1 = min(kt0 + p.kt_per_split, p.ktiles);
  int cidx = kt0 % p.kc;
  int tap_r = (kt0 / p.kc) / p.kw, tap_s = (kt0 / p.kc) % p.kw;
  auto gload = [&](int kt, int buf) {
    unsigned char* stage = smem + buf * STAGE;
    const int coff = cidx * 64 + chunk * 8;
#pragma unroll
    for (int i = 0; i < XPASS; ++i) {
      const int y = r_yx[i] >> 16, x = r_yx[i] & 0xffff;
      const int sh = r_hw[i] >> 16, sw = r_hw[i] & 0xffff;
      int sy, sx;
      bool ok = r_yx[i] >= 0;
      if (p.mode == 0) {
        sy = y * p.stride + tap_r - p.pad;
        sx = x * p.stride + tap_s - p.pad;
      } else {
        const int ty = y + p.pad - tap_r, tx = x + p.pad - tap_s;
        if (p.stride == 1) {
          sy = ty;
          sx = tx;
        } else {
          ok = ok && ty >= 0 && tx >= 0 && (ty % p.stride) == 0 && (tx % p.stride) == 0;
          sy = ty / p.stride;
          sx = tx / p.stride;
        }
      }
      ok = ok && (unsigned)sy < (unsigned)sh && (unsigned)sx < (unsigned)sw;
      const long long off = ((long long)(r_base[i] + sy * sw + sx)) * p.lds + coff;
      const gptr_t g = ok ? (gptr_t)(p.src + off) : zero;
      __builtin_amdgcn_global_load_lds(g, (lptr_t)(stage + TILE_W + (i * RPP + wave * 8) * 128), 16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < WPASS; ++i)
      __builtin_amdgcn_global_load_lds((gptr_t)(wbase + (long long)(RPP * i) * p.wrow + (long long)kt * BK),
                                       (lptr_t)(stage + (i * RPP + wave * 8) * 128), 16, 0, 0);
    if (++cidx == p.kc) {
      cidx = 0;
      if (++tap_s == p.kw) {
        tap_s = 0;
        ++tap_r;
      }
    }
  };

  f32x16 acc[2][PT];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < PT; ++b)
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[a][b][j] = 0.f;

  const int frow = lane & 31, fhalf = lane >> 5;
  const int fswz = (frow >> 1) & 7;
  auto compute = [&](int buf) {
    const unsigned char* base = smem + buf * STAGE;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int coff = ((2 * kk + fhalf) ^ fswz) << 4;
      bf16x8 a[2], b[PT];
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
        a[ct] = *reinterpret_cast<const bf16x8*>(base + (wave_co * 64 + ct * 32 + frow) * 128 + coff);
#pragma unroll
      for (int pt = 0; pt < PT; ++pt)
        b[pt] = *reinterpret_cast<const bf16x8*>(base + TILE_W + (wave_px * (32 * PT) + pt * 32 + frow) * 128 + coff);
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int pt = 0; pt < PT; ++pt)
          acc[ct][pt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ct], b[pt], acc[ct][pt], 0, 0, 0);
    }
  };

  // NST-stage ring: tiles kt+1 .. kt+NST-1 are in flight while tile kt feeds the MFMAs.  Each thread issues
  // LPT DMA instructions per tile, so "tile kt has landed" == at most (tiles still allowed in flight) * LPT
  // outstanding (counted s_waitcnt, raw s_barrier: __syncthreads() would drain the whole queue).
  constexpr int LPT = WPASS + XPASS;
  static_assert((NST - 2) * LPT <= 63, "vmcnt range");
#pragma unroll
  for (int s = 0; s < NST - 1; ++s)
    if (kt0 + s < kt1) gload(kt0 + s, s);
  int slot = 0;                      // ring slot of tile kt
  for (int kt = kt0; kt < kt1; ++kt) {
    const int ahead = min(kt1 - 1 - kt, NST - 2);     // tiles after kt that may stay in flight
    if (NST >= 4 && ahead >= 2) wait_vmcnt<(NST >= 4 ? 2 : 0) * LPT>();
    else if (NST >= 3 && ahead >= 1) wait_vmcnt<(NST >= 3 ? 1 : 0) * LPT>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();          // everyone's DMA for tile kt landed; compute(kt-1) finished everywhere
    if (kt + NST - 1 < kt1) gload(kt + NST - 1, slot == 0 ? NST - 1 : slot - 1);
    compute(slot);
    slot = (slot + 1 == NST) ? 0 : slot + 1;
  }

  if (p.splits > 1) {                      // split-K: raw fp32 partial tile -> workspace [split][pixel][cd_pad]
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
      const int gp = px0 + wave_px * (32 * PT) + pt * 32 + (lane & 31);
      if (gp >= totpx) continue;
      float* row = p.ws + ((long long)blockIdx.z * totpx + gp) * p.cd_pad;
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int co = co0 + wave_co * 64 + ct * 32 + 8 * g + 4 * (lane >> 5);
          f32x4 o = {acc[ct][pt][4 * g], acc[ct][pt][4 * g + 1], acc[ct][pt][4 * g + 2], acc[ct][pt][4 * g + 3]};
          *reinterpret_cast<f32x4*>(row + co) = o;
        }
    }
    return;
  }

  // ---- epilogue, staged through LDS: the MFMA result layout gives each lane one pixel and scattered 4-channel
  // runs (8-byte stores 2 KB apart); transposing 32*WPX pixels at a time through the (now free) stage memory
  // turns every global access of the epilogue - output, residual addend, ReLU mask - into 16-byte lanes that
  // cover whole 512-byte pixel rows.
  constexpr int ROWB = BCO * 4 + 16;            // fp32 row + 16 B pad: conflict-free ds_write_b128 down a column
  constexpr int CPX = 32 * WPX;                 // pixels per chunk
  constexpr int GPR = BCO / 8;                  // 8-channel groups per pixel row
  static_assert(CPX * ROWB <= NST * STAGE, "epilogue staging must fit in the stage memory");
#pragma unroll
  for (int pt = 0; pt < PT; ++pt) {
    lds_barrier();                            // previous chunk fully read (first pass: all MFMA operands consumed)
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int col = wave_co * 64 + ct * 32 + 8 * g + 4 * fhalf;
        f32x4 o = {acc[ct][pt][4 * g], acc[ct][pt][4 * g + 1], acc[ct][pt][4 * g + 2], acc[ct][pt][4 * g + 3]};
        *reinterpret_cast<f32x4*>(smem + (wave_px * 32 + frow) * ROWB + col * 4) = o;
      }
    lds_barrier();
    for (int id = tid; id < CPX * GPR; id += T) {
      const int pl = id / GPR, cg = id - pl * GPR;
      const int gp = px0 + (pl >> 5) * (32 * PT) + pt * 32 + (pl & 31);
      const int co = co0 + cg * 8;
      if (gp >= totpx || co >= p.cd) continue;
      const f32x4 lo = *reinterpret_cast<const f32x4*>(smem + pl * ROWB + cg * 32);
      const f32x4 hi = *reinterpret_cast<const f32x4*>(smem + pl * ROWB + cg * 32 + 16);
      float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      long long dpix, apix;
      conv_out_index(p, gp, dpix, apix);
      conv_epilogue8(p, dpix, apix, co, v);
    }
  }
}

// ================================================================================================
// v3: software-pipelined DMA-to-LDS implicit GEMM
//   * `buffer_load_dwordx4 ... lds` with 32-bit per-lane offsets: the padding / ragged-tile zero fill is the
//     buffer out-of-range rule (offset 0x80000000 -> zeros land in LDS), the per-tap address is one running
//     per-lane row offset + a scalar (SGPR) column/channel offset, validity is one bit test per pass
//   * MFMA operand fragments double-buffered in registers: the ds_reads of K-step kk+1 are issued before the
//     MFMAs of step kk, and the reads of the NEXT tile's step 0 before the MFMAs of this tile's step 3 -
//     the s_barrier sits inside the MFMA stream instead of in front of an empty pipe
//   * the ring slot of tile kt is free after its step-3 fragments are in registers, so NST slots hold NST
//     tiles in flight / in use
// Not handled here (host routes them to the v1 kernel): data-gradient with stride > 1, kh*kw > 16 per axis
// limits (kh, kw <= 8), sources >= 2 GiB.
// ================================================================================================
template <int NMF, int NDS, int NVM>
__device__ __forceinline__ void sched_stage() {      // NMF x { 1 MFMA [, 1 DS read for the first NDS] [, 1 VMEM for the first NVM] }
#pragma unroll
  for (int i = 0; i < NMF; ++i) {
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    if (i < NDS) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    if (i < NVM) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
  }
}

// SMC: the source has 8 channels per pixel (the NHWC8 image of the 7x7 stem): one 16-byte DMA lane is one TAP, a K tile
// is 8 consecutive taps, so every lane gathers its own tap's pixel (K index = tap*8 + channel, as the stem weights
// are packed).
#ifdef DSL_TRACE_BUILD
constexpr int kTraceIters = 40;
__device__ unsigned long long g_conv_trace[8 * kTraceIters * 8 + 8 * 16 + 8];
#endif

// ================================================================================================
// Output tile of the DMA-pipelined kernels (bf16 and fp8): accumulators -> destination.  `stamp(i)` is the trace build's
// s_memtime hook (a no-op otherwise).  RING = bytes of LDS the K loop used (free once every wave is here).
// ================================================================================================
template <int BCO, int BPX, int WCO, int WPX, int CT, int PT, int RING, bool GNB = false, class Stamp>
__device__ __forceinline__ void conv_tile_epilogue(const ConvK& p, f32x16 (&acc)[CT][PT], unsigned char* smem, const int co0, const int px0,
                                                   const int totpx, Stamp&& stamp) {
  constexpr int T = 64 * WCO * WPX;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wave_co = wave / WPX, wave_px = wave % WPX;
  const int frow = lane & 31, fhalf = lane >> 5;
  // ---- epilogue staged through LDS (see conv_glds_kernel)
  constexpr int ROWB = BCO * 4 + 16;
  constexpr int CPX = 32 * WPX;
  constexpr int GPR = BCO / 8;
  static_assert(CPX * ROWB <= 160 * 1024, "epilogue staging must fit in LDS (the host sizes LDS as max(ring, staging))");
  static_assert(T % GPR == 0, "a thread keeps its channel group across the staged rows");
  const Affine8 aff = conv_affine8(p, co0 + (tid % GPR) * 8);     // (issued here: the loads fly during the first staging round)
  const EpiFast ef = conv_epi_fast(p, co0 + (tid % GPR) * 8);
  // ---- "pure" epilogue (no addend, no mask, bf16 out, ident): scale / bias / ReLU / rounding happen in the accumulator registers
  // (a lane owns 4 consecutive couts of one pixel per 8-cout group), the WHOLE tile is staged once as bf16 rows [pixel][BCO] and
  // leaves as 16-byte stores without a single VALU instruction in the store loop: two barriers instead of 2 * PT, half the LDS
  // bytes, same arithmetic in the same order as the staged fp32 path (mul_nc / add_nc: one rounding each).
  constexpr int ROWH = BCO * 2 + 16;
  static_assert((long long)BPX * ROWH <= (long long)RING, "the bf16 tile fits in the ring");
  // ---- addend epilogue (residual adds of the bottlenecks' forward and data-gradient chains, the FPN's and the head's gradient sums):
  // the addend TILE is brought into the then-free ring by DMA as dense bf16 rows [pixel][BCO] (16-byte chunk c of row r in slot
  // c ^ f(r): the swizzle is applied through the source address, the LDS image of a DMA is lane-linear), every lane adds its 4-cout
  // runs in the accumulator registers - scale, bias, + addend, ReLU, one rounding: the staged fp32 path's operations in its order -
  // and writes the rounded run back IN PLACE (a run's slot belongs to exactly one lane); the tile then leaves as 16-byte stores like
  // the pure path's.  A ReLU mask applied last commutes with the rounding (see below).  One DMA round trip, three barriers and
  // BPX x BCO x 2 bytes of LDS instead of PT slabs of {barrier, fp32 staging, barrier, per-item global loads of the addend}.
  // (addfast == 2: the addend is nearest-upsampled - the FPN's top-down path in the lateral convolution's epilogue, fpn.py:163-172 - the
  // DMA lanes compute the addend's pixel, everything else is the same launch-on-its-own-grid case)
  if ((ef.on || (p.addfast == 2 && !(p.flags & DSL_CONV_OUT_F32))) && ef.has_add && p.addfast && !ef.mask_first && !(ef.mask_last && ef.relu) &&
      (p.cd & 7) == 0) {
    constexpr int RB = BCO * 2;                        // bytes per staged row
    constexpr int NCH = RB / 16;                       // 16-byte chunks per row
    constexpr int FSH = RB >= 256 ? 0 : (RB == 128 ? 1 : 2);      // rows that share a bank phase before the chunk rotation repeats
    static_assert((long long)BPX * RB <= (long long)RING, "the dense bf16 tile fits in the ring");
    static_assert((BPX * RB) % (T * 16) == 0, "whole DMA instructions per wave");
    constexpr int NDMA = BPX * RB / (T * 16);          // DMA instructions per wave (1 KiB each)
    lds_barrier();                                     // every wave is done with the ring
    {
      const __amdgpu_buffer_rsrc_t rs_add = __builtin_amdgcn_make_buffer_rsrc((void*)p.addend, 0, 0x7fffffff, 0x00020000);
#pragma unroll
      for (int i = 0; i < NDMA; ++i) {
        const int off = (i * (T / 64) + wave) * 1024 + lane * 16;     // this lane's LDS byte
        const int row = off / RB, slot = (off % RB) >> 4;
        const int c = slot ^ ((row >> FSH) & (NCH - 1));              // source chunk that belongs in this slot
        const int gp = px0 + row;
        const bool ok = gp < totpx && co0 + c * 8 < p.cd;
        int arow = gp;
        if (p.addfast == 2 && ok) {
          int seg, img, y, x;
          decode_pixel(p, gp, seg, img, y, x);
          const int ay = (y * p.ah[seg]) / p.dh[seg], ax = (x * p.aw[seg]) / p.dw[seg];
          arow = (int)p.aoff[seg] + (img * p.ah[seg] + ay) * p.aw[seg] + ax;
        }
        const unsigned v = ok ? (unsigned)(arow * p.lda + co0 + c * 8) * 2u : 0x80000000u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_add, (lptr_t)(smem + (i * (T / 64) + wave) * 1024), 16, v, 0, 0, 0);
      }
    }
    wait_vmcnt<0>();
    lds_barrier();
    stamp(1);
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int col = wave_co * (32 * CT) + ct * 32 + 8 * g + 4 * fhalf;
        const bool in = co0 + col + 3 < p.cd;
        const f32x4 sc = (ef.has_scale && in) ? *reinterpret_cast<const f32x4*>(p.scale + co0 + col) : f32x4{1.f, 1.f, 1.f, 1.f};
        const f32x4 bi = (ef.has_bias && in) ? *reinterpret_cast<const f32x4*>(p.bias + co0 + col) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) {
          const int row = wave_px * (32 * PT) + pt * 32 + frow;
          unsigned char* cell = smem + row * RB + ((((col >> 3) ^ ((row >> FSH) & (NCH - 1))) << 4) | (fhalf << 3));
          const u32x2 aa = *reinterpret_cast<const u32x2*>(cell);
          const float ad[4] = {bflo(aa[0]), bfhi(aa[0]), bflo(aa[1]), bfhi(aa[1])};
          float v[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            v[j] = acc[ct][pt][4 * g + j];
            if (ef.has_scale) v[j] = mul_nc(v[j], sc[j]);
            if (ef.has_bias) v[j] = add_nc(v[j], bi[j]);
            v[j] = add_nc(v[j], ad[j]);
            if (ef.relu) v[j] = fmaxf(v[j], 0.f);
          }
          const u32x2 o = {pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
          *reinterpret_cast<u32x2*>(cell) = o;
        }
      }
    lds_barrier();
    stamp(2);
    {
      constexpr int NITP = BPX * GPR / T;
      static_assert(BPX * GPR % T == 0, "whole items per thread");
      constexpr int RPT = T / GPR;
      const int row0 = tid / GPR, cgp = tid % GPR;
      if (co0 + cgp * 8 < p.cd) {
        uint16_t* out = reinterpret_cast<uint16_t*>(p.dst) + (co0 + cgp * 8);
#pragma unroll 4
        for (int n_ = 0; n_ < NITP; ++n_) {
          const int row = row0 + n_ * RPT;
          const int gp = px0 + row;
          if (gp < totpx) {
            u32x4 r = *reinterpret_cast<const u32x4*>(smem + row * RB + ((cgp ^ ((row >> FSH) & (NCH - 1))) << 4));
            if (ef.mask_last) {
              const u32x4 mm = *reinterpret_cast<const u32x4*>(ef.mask + (long long)gp * ef.ldm);
#pragma unroll
              for (int e = 0; e < 4; ++e)
                r[e] &= (bflo(mm[e]) > 0.f ? 0xffffu : 0x8000u) | (bfhi(mm[e]) > 0.f ? 0xffff0000u : 0x80000000u);
            }
            *reinterpret_cast<u32x4*>(out + (long long)gp * p.ldd) = r;
          }
        }
      }
    }
    stamp(3);
    return;
  }
  // A ReLU mask applied LAST (the data gradients: round(v * m), m in {0, 1}) commutes with the rounding - m ? round(v) : +-0 with v's
  // sign - so it is applied to the staged bf16 words in the store loop, bit for bit what the fp32 path produces for finite v.
  if (ef.on && !(p.flags & DSL_CONV_EPI_STAGED) && !ef.has_add && !ef.mask_first && !(ef.mask_last && ef.relu) && (p.cd & 7) == 0) {
    lds_barrier();                           // every wave is done with the ring (its DMA has landed: wait_vmcnt<0> above)
    // (cout group outermost: a lane keeps ONE group's scale / bias at a time - all of them at once cost 64 registers and an
    // occupancy step on the small tiles)
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int col = wave_co * (32 * CT) + ct * 32 + 8 * g + 4 * fhalf;
        const bool in = co0 + col + 3 < p.cd;
        const f32x4 sc = (ef.has_scale && in) ? *reinterpret_cast<const f32x4*>(p.scale + co0 + col) : f32x4{1.f, 1.f, 1.f, 1.f};
        const f32x4 bi = (ef.has_bias && in) ? *reinterpret_cast<const f32x4*>(p.bias + co0 + col) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) {
          float v[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            v[j] = acc[ct][pt][4 * g + j];
            if (ef.has_scale) v[j] = mul_nc(v[j], sc[j]);
            if (ef.has_bias) v[j] = add_nc(v[j], bi[j]);
            if (ef.relu) v[j] = fmaxf(v[j], 0.f);
          }
          const int row = wave_px * (32 * PT) + pt * 32 + frow;
          u32x2 o = {pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
          *reinterpret_cast<u32x2*>(smem + row * ROWH + col * 2) = o;
        }
      }
    lds_barrier();
    stamp(2);
    {
      constexpr int NITP = BPX * GPR / T;
      static_assert(BPX * GPR % T == 0, "whole items per thread");
      const int row0 = tid / GPR, cgp = tid % GPR;
      if (co0 + cgp * 8 < p.cd) {
        const unsigned char* rd = smem + row0 * ROWH + cgp * 16;
        uint16_t* out = reinterpret_cast<uint16_t*>(p.dst) + (co0 + cgp * 8);
#pragma unroll 4
        for (int n_ = 0; n_ < NITP; ++n_) {
          const int gp = px0 + row0 + n_ * (T / GPR);
          if (gp < totpx) {
            u32x4 r = *reinterpret_cast<const u32x4*>(rd + n_ * (T / GPR) * ROWH);
            if (ef.mask_last) {
              const u32x4 mm = *reinterpret_cast<const u32x4*>(ef.mask + (long long)gp * ef.ldm);
#pragma unroll
              for (int e = 0; e < 4; ++e)
                r[e] &= (bflo(mm[e]) > 0.f ? 0xffffu : 0x8000u) | (bfhi(mm[e]) > 0.f ? 0xffff0000u : 0x80000000u);
            }
            *reinterpret_cast<u32x4*>(out + (long long)gp * p.ldd) = r;
          }
        }
      }
      // ---- GroupNorm statistics of the tile (dsl_conv_desc.gn_ws; 8 channels per group, so a 16-byte chunk of the staged tile is
      // one pixel of one group): per (segment, image) row of the level-major pixel axis that crosses this tile, sum and sum of
      // squares of the ROUNDED values - what gn_stats_kernel would read back - reduced over the tile's pixels in a fixed order and
      // written as one record per (row, pixel tile); gn_apply_kernel adds a row's records up in tile order.  Record layout:
      // [64 floats of header: word 0 = BPX][segment * n + image][maxhw / 64 + 2 tiles][cd / 8 groups][2].
      bool gn_bwd = false;
      if constexpr (GNB) gn_bwd = p.gnws != nullptr && p.gnx != nullptr;
      if (gn_bwd) {
        // ---- backward records (gn_bwd_reduce_kernel's, per pixel tile instead of per 128-pixel block): per channel dg = sum dz*xhat,
        // db = sum dz, sx = sum xhat; per group s1 = sum dz*gamma, s2 = sum dz*gamma*xhat, dz = dy * [gamma*xhat + beta > 0]; dy is the
        // staged tile, x comes from HBM.  Reduction: lanes of a wave that share a channel group by shuffles, then the waves through
        // the LDS left behind the staged tile, WPR waves per round (the 256 x 192 tile has room for four of its eight), slot k adding
        // waves k, k + WPR, ... in order; every sum in a fixed order.
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        constexpr int RPT = T / GPR, NW = T / 64, NV = 26;
        static_assert(GPR <= 64 && 64 % GPR == 0, "lanes of one channel group inside a wave");
        constexpr int FREE = RING - BPX * ROWH;
        constexpr int WFIT = FREE / (GPR * NV * 4);
        constexpr int WPR = WFIT >= NW ? NW : (WFIT >= NW / 2 ? NW / 2 : (WFIT >= NW / 4 ? NW / 4 : 1));
        static_assert(WPR >= 1 && WPR * GPR * NV * 4 <= FREE && NW % WPR == 0, "reduction scratch behind the staged tile");
        float* red = reinterpret_cast<float*>(smem + BPX * ROWH);
        const unsigned char* rd = smem + row0 * ROWH + cgp * 16;
        const bool live = co0 + cgp * 8 < p.cd;
        const int ngr = p.cd >> 3;
        const int R = 3 * p.cd + 2 * ngr;
        int maxhw = 0;
#pragma unroll
        for (int sg = 0; sg < DSL_MAX_SEG; ++sg)
          if (sg < p.nseg) maxhw = max(maxhw, p.gh[sg] * p.gw[sg]);
        const int nbk = maxhw / 64 + 2;
        if (px0 == 0 && co0 == 0 && tid == 0) *reinterpret_cast<int*>(p.gnws) = BPX;
        f32x2 ga[4], be[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int ch = co0 + cgp * 8 + 2 * e;
          ga[e] = live ? f32x2{p.gngamma[ch], p.gngamma[ch + 1]} : f32x2{0.f, 0.f};
          be[e] = live ? f32x2{p.gnbeta[ch], p.gnbeta[ch + 1]} : f32x2{0.f, 0.f};
        }
        const uint16_t* xsrc = p.gnx + (long long)(px0 + row0) * p.cd + (co0 + cgp * 8);
        const int pend = min(px0 + BPX, totpx);
        int cur = px0;
#pragma nounroll
        while (cur < pend) {
          int seg, img, y_, x_;
          decode_pixel(p, cur, seg, img, y_, x_);
          const int hw = p.gh[seg] * p.gw[seg];
          const int rs = p.pxstart[seg] + img * hw;
          const int lo = cur - px0, hi = min(rs + hw, pend) - px0;
          const int si = seg * p.n + img;
          f32x2 dg[4], db[4], sx[4], s1 = {0.f, 0.f}, s2 = {0.f, 0.f};
#pragma unroll
          for (int e = 0; e < 4; ++e) dg[e] = db[e] = sx[e] = f32x2{0.f, 0.f};
          if (live) {
            const float* stp = p.gnstats + ((long long)si * ngr + (co0 >> 3) + cgp) * 2;
            const float rstd = stp[1], nmr = -stp[0] * rstd;
            // x in batches of XB chunks, one batch ahead of the arithmetic (the loads are the epilogue's only HBM latency)
            constexpr int XB = NITP % 4 == 0 ? 4 : 1;
            u32x4 xn[XB];
            auto fetch = [&](int b_) {
#pragma unroll
              for (int i_ = 0; i_ < XB; ++i_) {
                const int r_ = row0 + (b_ * XB + i_) * RPT;
                xn[i_] = (r_ >= lo && r_ < hi) ? *reinterpret_cast<const u32x4*>(xsrc + (long long)(b_ * XB + i_) * RPT * p.cd) : u32x4{0, 0, 0, 0};
              }
            };
            fetch(0);
#pragma nounroll
            for (int b_ = 0; b_ < NITP / XB; ++b_) {
              u32x4 xc[XB];
#pragma unroll
              for (int i_ = 0; i_ < XB; ++i_) xc[i_] = xn[i_];
              if (b_ + 1 < NITP / XB) fetch(b_ + 1);
#pragma unroll
              for (int i_ = 0; i_ < XB; ++i_) {
                const int n_ = b_ * XB + i_;
                const int r_ = row0 + n_ * RPT;
                if (r_ >= lo && r_ < hi) {
                  const u32x4 gv = *reinterpret_cast<const u32x4*>(rd + n_ * RPT * ROWH);
                  const u32x4 xv = xc[i_];
#pragma unroll
                  for (int e = 0; e < 4; ++e) {
                    const f32x2 xx = {bflo(xv[e]), bfhi(xv[e])};
                    const f32x2 gg = {bflo(gv[e]), bfhi(gv[e])};
                    const f32x2 xh = xx * rstd + nmr;
                    const f32x2 t = xh * ga[e] + be[e];
                    const f32x2 dz = {t[0] > 0.f ? gg[0] : 0.f, t[1] > 0.f ? gg[1] : 0.f};
                    dg[e] += dz * xh;
                    db[e] += dz;
                    sx[e] += xh;
                    const f32x2 u = dz * ga[e];
                    s1 += u;
                    s2 += u * xh;
                  }
                }
              }
            }
          }
          float v[NV];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v[2 * e] = dg[e][0]; v[2 * e + 1] = dg[e][1];
            v[8 + 2 * e] = db[e][0]; v[8 + 2 * e + 1] = db[e][1];
            v[16 + 2 * e] = sx[e][0]; v[16 + 2 * e + 1] = sx[e][1];
          }
          v[24] = s1[0] + s1[1];
          v[25] = s2[0] + s2[1];
#pragma unroll
          for (int off = GPR; off < 64; off <<= 1)
#pragma unroll
            for (int k_ = 0; k_ < NV; ++k_) v[k_] += __shfl_xor(v[k_], off);
          // lanes [0, GPR) of every wave now hold the wave's sums for channel group `lane`
#pragma unroll
          for (int rnd = 0; rnd < NW / WPR; ++rnd) {
            if (wave / WPR == rnd && lane < GPR) {
              float* slot = red + ((wave % WPR) * GPR + lane) * NV;
#pragma unroll
              for (int k_ = 0; k_ < NV; ++k_) slot[k_] = rnd == 0 ? v[k_] : slot[k_] + v[k_];
            }
            lds_barrier();
          }
          {
            const int j = px0 / BPX - rs / BPX;
            float* rec = p.gnws + 64 + ((long long)si * nbk + j) * R;
#pragma nounroll
            for (int id = tid; id < GPR * NV; id += T) {
              const int cg = id / NV, k_ = id - cg * NV;
              float a = 0.f;
#pragma unroll
              for (int w_ = 0; w_ < WPR; ++w_) a += red[(w_ * GPR + cg) * NV + k_];
              if (co0 + cg * 8 < p.cd) {
                if (k_ < 24) rec[(k_ >> 3) * p.cd + co0 + cg * 8 + (k_ & 7)] = a;
                else rec[3 * p.cd + 2 * ((co0 >> 3) + cg) + (k_ - 24)] = a;
              }
            }
          }
          lds_barrier();
          cur = rs + hw;
        }
      } else if (p.gnws) {
        constexpr int RPT = T / GPR;
        static_assert((long long)BPX * ROWH + T * 8 <= (long long)RING, "reduction scratch behind the staged tile");
        static_assert(BPX >= 64, "the record count per row is sized for pixel tiles of at least 64");
        float* red = reinterpret_cast<float*>(smem + BPX * ROWH);
        const unsigned char* rd = smem + row0 * ROWH + cgp * 16;
        const bool live = co0 + cgp * 8 < p.cd;
        const int ngr = p.cd >> 3;
        int maxhw = 0;
#pragma unroll
        for (int sg = 0; sg < DSL_MAX_SEG; ++sg)
          if (sg < p.nseg) maxhw = max(maxhw, p.gh[sg] * p.gw[sg]);
        const int nbk = maxhw / 64 + 2;
        if (px0 == 0 && co0 == 0 && tid == 0) *reinterpret_cast<int*>(p.gnws) = BPX;
        const int pend = min(px0 + BPX, totpx);
        int cur = px0;
#pragma nounroll
        while (cur < pend) {
          int seg, img, y_, x_;
          decode_pixel(p, cur, seg, img, y_, x_);
          const int hw = p.gh[seg] * p.gw[seg];
          const int rs = p.pxstart[seg] + img * hw;
          const int lo = cur - px0, hi = min(rs + hw, pend) - px0;
          float s = 0.f, ss = 0.f;
          if (live) {
#pragma unroll 4
            for (int n_ = 0; n_ < NITP; ++n_) {
              const int r_ = row0 + n_ * RPT;
              if (r_ >= lo && r_ < hi) {
                const u32x4 v = *reinterpret_cast<const u32x4*>(rd + n_ * RPT * ROWH);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                  const float a = bflo(v[e]), b = bfhi(v[e]);
                  s += a + b;
                  ss += a * a + b * b;
                }
              }
            }
          }
          red[tid * 2] = s;
          red[tid * 2 + 1] = ss;
          lds_barrier();
          if (tid < GPR && co0 + tid * 8 < p.cd) {
            float a = 0.f, b = 0.f;
#pragma unroll
            for (int k_ = 0; k_ < RPT; ++k_) {
              a += red[(k_ * GPR + tid) * 2];
              b += red[(k_ * GPR + tid) * 2 + 1];
            }
            const int si = seg * p.n + img;
            const int j = px0 / BPX - rs / BPX;
            float* dst = p.gnws + 64 + (((long long)si * nbk + j) * ngr + (co0 >> 3) + tid) * 2;
            dst[0] = a;
            dst[1] = b;
          }
          lds_barrier();
          cur = rs + hw;
        }
      }
    }
    stamp(3);
    return;
  }
  // The slab loop is NOT unrolled and the item loops are rolled: this code runs once per workgroup, straight-line copies of it per
  // slab are cold in the instruction cache every time (tools/trace_conv.py: 5 000 - 7 000 cycles per slab whatever the body did,
  // against ~1 000 for the same stores from warm code) - only the accumulator -> LDS writes need the slab index at compile time.
  auto stage_slab = [&](auto pt_c) {
    constexpr int pt = decltype(pt_c)::value;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int col = wave_co * (32 * CT) + ct * 32 + 8 * g + 4 * fhalf;
        f32x4 o = {acc[ct][pt][4 * g], acc[ct][pt][4 * g + 1], acc[ct][pt][4 * g + 2], acc[ct][pt][4 * g + 3]};
        *reinterpret_cast<f32x4*>(smem + (wave_px * 32 + frow) * ROWB + col * 4) = o;
      }
  };
  static_assert(PT <= 4, "slab dispatch");
#pragma nounroll
  for (int pt = 0; pt < PT; ++pt) {
    lds_barrier();
    stamp(1 + 3 * pt);
    if (pt == 0) stage_slab(std::integral_constant<int, 0>{});
    else if (pt == 1) stage_slab(std::integral_constant<int, (PT > 1 ? 1 : 0)>{});
    else if (pt == 2) stage_slab(std::integral_constant<int, (PT > 2 ? 2 : 0)>{});
    else stage_slab(std::integral_constant<int, (PT > 3 ? 3 : 0)>{});
    lds_barrier();
    stamp(2 + 3 * pt);
    if (ef.on && aff.full) {                 // lean item loop (uniform test; aff.full is false only in a partial last channel group)
      constexpr int NIT = CPX * GPR / T;     // items per thread and slab: rows T / GPR apart
      static_assert(CPX * GPR % T == 0, "whole items per thread");
      const int pl0 = tid / GPR;
      const unsigned char* rd = smem + pl0 * ROWB + (tid % GPR) * 32;
#pragma nounroll
      for (int n_ = 0; n_ < NIT; ++n_) {
        const int pl = pl0 + n_ * (T / GPR);
        const int gp = px0 + (pl >> 5) * (32 * PT) + pt * 32 + (pl & 31);
        if (gp < totpx)
          conv_epilogue8_fast(ef, aff, gp, *reinterpret_cast<const f32x4*>(rd + n_ * (T / GPR) * ROWB),
                              *reinterpret_cast<const f32x4*>(rd + n_ * (T / GPR) * ROWB + 16));
      }
      stamp(3 + 3 * pt);
      continue;
    }
#pragma nounroll
    for (int id = tid; id < CPX * GPR; id += T) {
      const int pl = id / GPR, cg = id - pl * GPR;
      const int gp = px0 + (pl >> 5) * (32 * PT) + pt * 32 + (pl & 31);
      const int co = co0 + cg * 8;
      if (gp >= totpx || co >= p.cd) continue;
      const f32x4 lo = *reinterpret_cast<const f32x4*>(smem + pl * ROWB + cg * 32);
      const f32x4 hi = *reinterpret_cast<const f32x4*>(smem + pl * ROWB + cg * 32 + 16);
      float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#ifdef DSL_ABLATE_BUILD
      if ((p.dbg & 512) && v[0] != 12345.678f) continue;      // epilogue without the scale / bias / addend loads and the stores
#endif
      long long dpix, apix;
      conv_out_index(p, gp, dpix, apix);
      conv_epilogue8a(p, dpix, apix, co, v, aff);
    }
    stamp(3 + 3 * pt);
  }
}

template <int BCO, int BPX, int WCO, int WPX, int NST, bool SMC = false, bool GNB = false>
__global__ __launch_bounds__(64 * WCO * WPX) void conv_pipe_kernel(const ConvK p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int T = 64 * WCO * WPX;
  constexpr int RPP = T / 8;                 // tile rows filled per pass (8 lanes per 128-byte row)
  constexpr int WPASS = BCO / RPP, XPASS = BPX / RPP;
  constexpr int TILE_W = BCO * 128;
  constexpr int STAGE = (BCO + BPX) * 128;
  constexpr int PT = BPX / WPX / 32;         // 32-pixel MFMA tiles per wave
  constexpr int CT = BCO / WCO / 32;         // 32-cout MFMA tiles per wave
  constexpr int HB = 1;                      // cout tiles of the K tile's LAST k-step whose MFMAs are held back across the barrier (their
                                             // fragments are in registers): what the matrix pipe runs while the next tile's first reads fly
  static_assert(BCO == WCO * CT * 32 && CT >= 1, "cout tiles per wave");
  static_assert(BCO % RPP == 0 && BPX % RPP == 0 && (BPX / WPX) % 32 == 0, "tile/thread mismatch");

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wave_co = wave / WPX, wave_px = wave % WPX;
  // XCD-aware tile order (workgroups are dealt round-robin to the XCDs: equal b % 8 = same XCD): every XCD owns a contiguous run of tiles in (cout tile
  // fastest, then pixel tile, then K split) order, so neighbouring pixel tiles - which share their halo rows - and
  // the cout tiles of one pixel range hit the same L2 instead of being fetched into up to three of them.
#ifdef DSL_ABLATE_BUILD
  if (p.dbg & 128) return;                   // launch + dispatch floor
#endif
  const int wi = (int)(blockIdx.x & 7) * p.xcd_chunk + (int)(blockIdx.x >> 3);
  if ((int)(blockIdx.x >> 3) >= p.xcd_chunk || wi >= p.gx * p.gy * p.splits) return;
  const int bz = wi / (p.gx * p.gy);
  const int rem_t = wi - bz * (p.gx * p.gy);
  const int by = rem_t / p.gx;
  const int co0 = (rem_t - by * p.gx) * BCO;
  const int px0 = by * BPX;
  const int totpx = p.pxstart[p.nseg];
  const int dt = tid, dwave = wave;
  const int lrow = dt >> 3;
  const int chunk = (dt & 7) ^ ((dt >> 4) & 7);       // source chunk that belongs in LDS slot (dt & 7) of this row

  // buffer resources: base shifted back by `margin` so that every VALID tap has a non-negative per-lane offset
  // (the hardware range-checks the per-lane offset, not the scalar one)
  const unsigned margin = (unsigned)(p.kw * p.lds * 2);
  const __amdgpu_buffer_rsrc_t rs_src =
      __builtin_amdgcn_make_buffer_rsrc((void*)(reinterpret_cast<const unsigned char*>(p.src) - margin), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_wgt = __builtin_amdgcn_make_buffer_rsrc((void*)p.wgt, 0, 0x7fffffff, 0x00020000);

  const int kt0 = bz * p.kt_per_split;
#ifdef DSL_ABLATE_BUILD
  const int kt1 = (p.dbg & 8) ? kt0 + 1 : min(kt0 + p.kt_per_split, p.ktiles);
#else
  const int kt1 = min(kt0 + p.kt_per_split, p.ktiles);
#endif
  int cidx = kt0 % p.kc;
  int tap_r = (kt0 / p.kc) / p.kw, tap_s = (kt0 / p.kc) % p.kw;

  unsigned r_cur[XPASS], r_step[XPASS], r_mask[XPASS];
  int r_y[XPASS], r_x[XPASS], r_hw[XPASS];            // SMC: top-left source pixel of the window, source size
#pragma unroll
  for (int i = 0; i < XPASS; ++i) {
    const int gp = px0 + lrow + RPP * i;
    int seg = 0, img = 0, y = 0, x = 0;
    const bool ok = gp < totpx;
    if (ok) decode_pixel(p, gp, seg, img, y, x);
    const int sh = p.sh[seg], sw = p.sw[seg];
    if (SMC) {
      r_y[i] = ok ? y * p.stride - p.pad : -100000;    // not ok: every tap fails the bounds test
      r_x[i] = x * p.stride - p.pad;
      r_hw[i] = (sh << 16) | sw;
      r_cur[i] = (unsigned)(((int)(p.soff[seg]) + img * sh * sw + r_y[i] * sw + r_x[i]) * 16) + margin;
      r_step[i] = r_mask[i] = 0;
      continue;
    }
    const int row0 = p.mode == 0 ? y * p.stride - p.pad : y + p.pad;               // source row of tap r = 0
    const int col0 = p.mode == 0 ? x * p.stride - p.pad : x + p.pad - (p.kw - 1);  // leftmost source column
    unsigned m = 0;
    for (int r = 0; r < p.kh; ++r) {
      const int sy = p.mode == 0 ? row0 + r : row0 - r;
      if (ok && (unsigned)sy < (unsigned)sh) m |= 1u << r;
    }
    for (int s_ = 0; s_ < p.kw; ++s_) {
      const int sx = p.mode == 0 ? col0 + s_ : x + p.pad - s_;
      if (ok && (unsigned)sx < (unsigned)sw) m |= 0x100u << s_;
    }
    r_mask[i] = m;
    const unsigned pitch = (unsigned)(sw * p.lds * 2);
    r_step[i] = p.mode == 0 ? pitch : 0u - pitch;
    const unsigned base = (unsigned)(((int)(p.soff[seg]) + img * sh * sw + row0 * sw + col0) * p.lds + chunk * 8) * 2u + margin;
    r_cur[i] = base + (unsigned)tap_r * r_step[i];
  }
  const unsigned w_voff = (unsigned)(lrow * (int)p.wrow + chunk * 8) * 2u;
  const unsigned w_pass = (unsigned)(RPP * (int)p.wrow) * 2u;
  unsigned w_soff = (unsigned)(co0 * (int)p.wrow + kt0 * BK) * 2u;

  // ---- DMA of one K tile = LPT "pieces" per thread (XPASS pixel passes, then WPASS weight passes), issued a few at
  // a time between the MFMAs: a burst of all pieces right after the barrier fills the CU's address queue and
  // every wave then blocks on issue with an empty MFMA pipe.
  // Branch-free: past the last tile every lane goes out of range (zeros land in a slot nobody reads), so each
  // iteration issues exactly LPT DMA instructions and the vmcnt bookkeeping is a compile-time constant.
  constexpr int LPT = WPASS + XPASS;
  constexpr int P0 = (LPT + 1) / 3;                   // pieces issued right after the barrier (stage 3)
  constexpr int P1 = P0 + (LPT - P0 + 1) / 2;         // pieces [P0, P1) in stage 0, [P1, LPT) in stage 1
  int kt_next = kt0;               // K tile being fetched
  // The fetched tile's parameters are STATE, recomputed once per tile when the tile is fully issued (and the tap advance only on
  // the tile where the channel blocks wrap, behind a uniform branch): recomputing them branch-free in every pieces() call made 66
  // SALU instructions per K tile and wave, and these in-order waves with 8 MFMAs per K tile (the 128 x 128 / 128 x 64 tiles) are
  // bound by their own instruction count (probe: + 32 SALU or VALU per K tile = + 9 ... 12 % on the layer2-4 shapes).
  unsigned t_sel = (1u << tap_r) | (0x100u << tap_s);
  unsigned t_soff = (unsigned)((p.mode == 0 ? tap_s : p.kw - 1 - tap_s) * p.lds * 2 + cidx * 128);
  unsigned t_wv = w_voff;
  int c_left = p.kc - cidx;        // tiles until the channel blocks wrap (tap advance)
  unsigned r_v[XPASS];             // this lane's source offsets for the CURRENT tap (out of range where the tap leaves the image):
#pragma unroll                     // they change with the tap only, so the per-piece mask test moves into the tap advance
  for (int i = 0; i < XPASS; ++i) r_v[i] = (r_mask[i] & t_sel) == t_sel ? r_cur[i] : 0x80000000u;
  // ld_off: byte offset of the ring slot being filled (the K loop derives it from the slot it reads: no second ring counter)
  auto pieces = [&](auto lo_c, auto hi_c, const unsigned ld_off) {
    constexpr int LO = decltype(lo_c)::value, HI = decltype(hi_c)::value;
    unsigned char* stage = smem + ld_off;
    const bool live = kt_next < kt1;      // (used by the 8-channel-source variant only)
    const unsigned s_off = t_soff;
    const unsigned wv = t_wv;
    int s_tr = 0, s_ts = 0;
    bool s_ok = false;
    if (SMC) {                    // this lane's tap of the K tile
      const int tap = kt_next * 8 + chunk;
      s_tr = tap / p.kw;
      s_ts = tap - s_tr * p.kw;
      s_ok = live && tap < p.kh * p.kw;
    }
#pragma unroll
    for (int j = LO; j < HI; ++j) {
#ifdef DSL_ABLATE_BUILD
      if (p.dbg & (j < XPASS ? 1 : 2)) continue;
#endif
      if (j < XPASS) {
        unsigned v, so;
        if (SMC) {
          const int sh = r_hw[j] >> 16, sw = r_hw[j] & 0xffff;
          const bool in = s_ok && (unsigned)(r_y[j] + s_tr) < (unsigned)sh && (unsigned)(r_x[j] + s_ts) < (unsigned)sw;
          v = in ? r_cur[j] + (unsigned)((s_tr * sw + s_ts) * 16) : 0x80000000u;
          so = 0;
        } else {
          v = r_v[j];
          so = s_off;
        }
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_src, (lptr_t)(stage + TILE_W + (j * RPP + dwave * 8) * 128), 16, v, so, 0, 0);
      } else {
        const int i = j - XPASS;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_wgt, (lptr_t)(stage + (i * RPP + dwave * 8) * 128), 16, wv,
                                                 w_soff + i * w_pass, 0, 0);
      }
    }
    if (HI == LPT) {               // tile fully issued: advance to the next (r, s, channel-block) and ring slot
      ++kt_next;
      w_soff += BK * 2;
      if (--c_left != 0) {
        t_soff += 128;             // same tap, next 64-channel block
      } else {                     // tap advance: every kc-th tile (uniform branch)
        c_left = p.kc;
        const bool s_wrap = tap_s + 1 == p.kw;
        tap_s = s_wrap ? 0 : tap_s + 1;
        if (s_wrap) {
          ++tap_r;
#pragma unroll
          for (int i = 0; i < XPASS; ++i) r_cur[i] += r_step[i];
        }
        t_sel = (1u << tap_r) | (0x100u << tap_s);
        t_soff = (unsigned)((p.mode == 0 ? tap_s : p.kw - 1 - tap_s) * p.lds * 2);
#pragma unroll
        for (int i = 0; i < XPASS; ++i) r_v[i] = (r_mask[i] & t_sel) == t_sel ? r_cur[i] : 0x80000000u;
      }
      if (__builtin_expect(kt_next >= kt1, 0)) {   // past the last tile (the ring's tail): every lane out of range, zeros land in a slot
        asm volatile("" ::: "memory");            // nobody reads (a real branch: if-converted it costs five instructions per tile)
        t_sel = 0xffffffffu;
        t_wv = 0x80000000u;
#pragma unroll
        for (int i = 0; i < XPASS; ++i) r_v[i] = 0x80000000u;
      }
    }
  };
  using c0_t = std::integral_constant<int, 0>;
  using cp0_t = std::integral_constant<int, P0>;
  using cp1_t = std::integral_constant<int, P1>;
  using clpt_t = std::integral_constant<int, LPT>;

#ifdef DSL_ABLATE_BUILD
  if (p.dbg & 256) return;                   // + kernel-argument loads and the per-pixel decode
#endif
  f32x16 acc[CT][PT];
#pragma unroll
  for (int a = 0; a < CT; ++a)
#pragma unroll
    for (int b = 0; b < PT; ++b)
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[a][b][j] = 0.f;

  const int frow = lane & 31, fhalf = lane >> 5;
  const int fswz = (frow >> 1) & 7;
  const int a_off = (wave_co * (32 * CT) + frow) * 128;
  const int b_off = TILE_W + (wave_px * (32 * PT) + frow) * 128;
  bf16x8 fa[2][CT], fb[2][PT];
  auto lds_read = [&](const unsigned char* base, int kk, int f) {
#ifdef DSL_ABLATE_BUILD
    if (p.dbg & 64) return;
#endif
    const int coff = ((2 * kk + fhalf) ^ fswz) << 4;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) fa[f][ct] = *reinterpret_cast<const bf16x8*>(base + a_off + ct * 32 * 128 + coff);
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) fb[f][pt] = *reinterpret_cast<const bf16x8*>(base + b_off + pt * 32 * 128 + coff);
  };
  auto mma_half = [&](int f, int ct) {
#ifdef DSL_ABLATE_BUILD
    if (p.dbg & 4) return;
#endif
#pragma unroll
    for (int pt = 0; pt < PT; ++pt)
      acc[ct][pt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[f][ct], fb[f][pt], acc[ct][pt], 0, 0, 0);
  };
  auto mma_upto = [&](int f, int n) {       // cout tiles [0, n) of fragment set f
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
      if (ct < n) mma_half(f, ct);
  };
  auto mma = [&](int f) { mma_upto(f, CT); };

  static_assert((NST - 1) * LPT <= 63, "vmcnt range");
#ifdef DSL_TRACE_BUILD
  // s_memtime stamps of ONE workgroup's K loop (tools/trace_conv.py): [wave][iteration][point] in the LDS behind the ring, dumped
  // to g_conv_trace at the end.  Points: 0 loop top, 1 every pre-barrier MFMA issued, 2 own LDS reads landed, 3 next tile's DMA
  // landed, 4 barrier passed, 5 post-barrier MFMAs issued.
  const bool tr_on = wi == p.dbg;
  int tr_it = 0;
  unsigned long long* tr_lds = reinterpret_cast<unsigned long long*>(smem + NST * STAGE) + wave * (kTraceIters * 8);
#define TR(i)                                                                                  \
  do {                                                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                         \
    if (tr_on && tr_it < kTraceIters) {                                                        \
      const unsigned long long t_ = __builtin_amdgcn_s_memtime();                              \
      if (lane == 0) tr_lds[tr_it * 8 + (i)] = t_;                                             \
    }                                                                                          \
    __builtin_amdgcn_sched_barrier(0);                                                         \
  } while (0)
#else
#define TR(i) do {} while (0)
#endif
  // Static priority for the second-dispatched half of an 8-wave workgroup: on each SIMD the younger wave otherwise loses every
  // issue arbitration to its partner and reaches the K loop's barrier ~750 cycles late (tools/trace_conv.py); measured + 0.75 % on the step.
  if (WCO * WPX == 8 && wave >= 4) __builtin_amdgcn_s_setprio(1);
  // prologue: NST-1 whole tiles + the first pieces of the NST-th
#pragma unroll
  for (int s = 0; s < NST - 1; ++s) pieces(c0_t{}, clpt_t{}, (unsigned)(s * STAGE));
  pieces(c0_t{}, cp0_t{}, (unsigned)((NST - 1) * STAGE));
  wait_vmcnt<(NST - 2) * LPT + P0>();
  __builtin_amdgcn_s_barrier();
  lds_read(smem, 0, 0);
  unsigned slot = 0;                       // byte offset of the ring slot being read
  unsigned pslot = (NST - 1) * STAGE;      // ... of the slot before it: where tile kt + NST - 1 is still being fetched into
  for (int kt = kt0; kt < kt1 - 1; ++kt) {
    const unsigned char* base = smem + slot;
    const unsigned nslot = (slot + STAGE == NST * STAGE) ? 0u : slot + STAGE;
    TR(0);
#ifdef DSL_ABLATE_BUILD
    if (p.dbg & 1024) {                    // sensitivity probe: 32 extra dependent SALU instructions per K tile and wave
      unsigned d_ = (unsigned)kt;
#pragma unroll
      for (int q = 0; q < 32; ++q) asm volatile("s_add_u32 %0, %0, 1" : "+s"(d_));
      if (d_ == 0x7fffffffu) return;
    }
    if (p.dbg & 2048) {                    // ... and 32 extra VALU instructions
      unsigned d_ = (unsigned)lane;
#pragma unroll
      for (int q = 0; q < 32; ++q) asm volatile("v_add_u32 %0, %0, 1" : "+v"(d_));
      if (d_ == 0x7fffffffu) return;
    }
#endif
    lds_read(base, 1, 1);
    pieces(cp0_t{}, cp1_t{}, pslot);
    mma(0);
    lds_read(base, 2, 0);
    pieces(cp1_t{}, clpt_t{}, pslot);
    mma(1);
    lds_read(base, 3, 1);
    mma(0);
    mma_upto(1, CT - HB);
    sched_stage<CT * PT, CT + PT, P1 - P0>();
    sched_stage<CT * PT, CT + PT, LPT - P1>();
    sched_stage<CT * PT, CT + PT, 0>();
    sched_stage<(CT - HB) * PT, 0, 0>();
    __builtin_amdgcn_sched_barrier(0);     // keep these MFMAs in front of the waits below
    TR(1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's reads of tile kt are in registers (issued >= PT MFMAs ago)
    TR(2);
    wait_vmcnt<(NST - 2) * LPT>();     // tile kt+1 landed (tiles kt+2 .. kt+NST-1 may stay in flight)
    TR(3);
#ifdef DSL_ABLATE_BUILD
    if (!(p.dbg & 32))
#endif
    __builtin_amdgcn_s_barrier();          // ... and both hold for every wave
    TR(4);
    lds_read(smem + nslot, 0, 0);
    pieces(c0_t{}, cp0_t{}, slot);     // start refilling the slot tile kt just vacated with tile kt+NST
#pragma unroll
    for (int ct = CT - HB; ct < CT; ++ct) mma_half(1, ct);
    __builtin_amdgcn_sched_group_barrier(0x100, CT + PT, 0);
    sched_stage<HB * PT, 0, P0>();
    TR(5);
#ifdef DSL_TRACE_BUILD
    ++tr_it;
#endif
    pslot = slot;
    slot = nslot;
  }
  {                                        // last tile
    const unsigned char* base = smem + slot;
    lds_read(base, 1, 1);
    mma(0);
    lds_read(base, 2, 0);
    mma(1);
    lds_read(base, 3, 1);
    mma(0);
    mma(1);
    sched_stage<CT * PT, CT + PT, 0>();
    sched_stage<CT * PT, CT + PT, 0>();
    sched_stage<CT * PT, CT + PT, 0>();
  }
  wait_vmcnt<0>();  // the out-of-range tail DMAs still write (zeros) into the ring
#ifdef DSL_TRACE_BUILD
#undef TR
  // epilogue stamps: [wave][16] behind the K-loop stamps; 0 = epilogue entered, then per staged slab: 1 + 3 r = first barrier
  // passed, 2 + 3 r = slab written and second barrier passed, 3 + 3 r = the slab's stores issued
  unsigned long long* tre_lds = reinterpret_cast<unsigned long long*>(smem + NST * STAGE) + 8 * kTraceIters * 8 + wave * 16;
#define TRE(i)                                                                                 \
  do {                                                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                         \
    if (tr_on) {                                                                               \
      const unsigned long long t_ = __builtin_amdgcn_s_memtime();                              \
      if (lane == 0) tre_lds[(i)] = t_;                                                        \
    }                                                                                          \
    __builtin_amdgcn_sched_barrier(0);                                                         \
  } while (0)
  TRE(0);
  auto trace_dump = [&]() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    TRE(15);                                  // every store of this wave acknowledged
    if (tr_on) {
      __syncthreads();
      for (int i = tid; i < 8 * kTraceIters * 8 + 8 * 16; i += T)
        g_conv_trace[i] = reinterpret_cast<const unsigned long long*>(smem + NST * STAGE)[i];
      if (tid == 0) g_conv_trace[8 * kTraceIters * 8 + 8 * 16] = (unsigned long long)(WCO * WPX);
    }
  };
#else
#define TRE(i) do {} while (0)
#endif
#ifdef DSL_ABLATE_BUILD
  if (p.dbg & 16) return;
#endif

  if (p.splits > 1) {                      // split-K: raw fp32 partial tile -> workspace [split][pixel][cd_pad]
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
      const int gp = px0 + wave_px * (32 * PT) + pt * 32 + (lane & 31);
      if (gp >= totpx) continue;
      float* row = p.ws + ((long long)bz * totpx + gp) * p.cd_pad;
#pragma unroll
      for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int co = co0 + wave_co * (32 * CT) + ct * 32 + 8 * g + 4 * (lane >> 5);
          f32x4 o = {acc[ct][pt][4 * g], acc[ct][pt][4 * g + 1], acc[ct][pt][4 * g + 2], acc[ct][pt][4 * g + 3]};
          *reinterpret_cast<f32x4*>(row + co) = o;
        }
    }
    return;
  }

  // ---- epilogue (conv_tile_epilogue)
#ifdef DSL_TRACE_BUILD
  conv_tile_epilogue<BCO, BPX, WCO, WPX, CT, PT, NST * STAGE, GNB>(p, acc, smem, co0, px0, totpx, [&](int i_) { TRE(i_); });
#else
  conv_tile_epilogue<BCO, BPX, WCO, WPX, CT, PT, NST * STAGE, GNB>(p, acc, smem, co0, px0, totpx, [](int) {});
#endif
#ifdef DSL_TRACE_BUILD
  trace_dump();
#endif
#undef TRE
}

// ================================================================================================
// fp8 forward convolution (BASELINE.json configs[4], first slice): the v3 kernel with 1-byte operands - OCP e4m3 activations
// [pixel][C] and weights [CoutPad][kh][kw][Cin], quantised by dsl_quant_fp8 / dsl_quant_fp8_weights - on the MX-scaled MFMA
// v_mfma_scale_f32_32x32x64_f8f6f4 (block scales 2^0; the per-output-channel weight scale and the per-tensor activation scale are
// folded into the fp32 epilogue's `scale`).  Same LDS geometry as bf16 (128-byte rows, same swizzle, same DMA instruction count
// per K tile), but a K tile is 128 elements: half the DMA / LDS bytes per FLOP and twice the MFMA rate.  Forward mode, Cin % 128 == 0.
// The reference has no such path (its training is fp32); off by default behind the detector's `fp8=dict(...)` key.
// ================================================================================================
typedef int v8i32 __attribute__((ext_vector_type(8)));
template <int BCO, int BPX, int WCO, int WPX, int NST>
__global__ __launch_bounds__(64 * WCO * WPX) void conv_f8_kernel(const ConvK p) {
  constexpr bool SMC = false;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int T = 64 * WCO * WPX;
  constexpr int RPP = T / 8;                 // tile rows filled per pass (8 lanes per 128-byte row)
  constexpr int WPASS = BCO / RPP, XPASS = BPX / RPP;
  constexpr int TILE_W = BCO * 128;
  constexpr int STAGE = (BCO + BPX) * 128;
  constexpr int PT = BPX / WPX / 32;         // 32-pixel MFMA tiles per wave
  static_assert(BCO / WCO == 64, "each wave owns 64 couts");
  static_assert(BCO % RPP == 0 && BPX % RPP == 0 && (BPX / WPX) % 32 == 0, "tile/thread mismatch");

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wave_co = wave / WPX, wave_px = wave % WPX;
  // XCD-aware tile order (workgroups are dealt round-robin to the XCDs: equal b % 8 = same XCD): every XCD owns a contiguous run of tiles in (cout tile
  // fastest, then pixel tile, then K split) order, so neighbouring pixel tiles - which share their halo rows - and
  // the cout tiles of one pixel range hit the same L2 instead of being fetched into up to three of them.
  const int wi = (int)(blockIdx.x & 7) * p.xcd_chunk + (int)(blockIdx.x >> 3);
  if ((int)(blockIdx.x >> 3) >= p.xcd_chunk || wi >= p.gx * p.gy * p.splits) return;
  const int bz = wi / (p.gx * p.gy);
  const int rem_t = wi - bz * (p.gx * p.gy);
  const int by = rem_t / p.gx;
  const int co0 = (rem_t - by * p.gx) * BCO;
  const int px0 = by * BPX;
  const int totpx = p.pxstart[p.nseg];
  const int lrow = tid >> 3;
  const int chunk = (tid & 7) ^ ((tid >> 4) & 7);     // source chunk that belongs in LDS slot (tid & 7) of this row

  // buffer resources: base shifted back by `margin` so that every VALID tap has a non-negative per-lane offset
  // (the hardware range-checks the per-lane offset, not the scalar one)
  const unsigned margin = (unsigned)(p.kw * p.lds);            // (1-byte elements)
  const __amdgpu_buffer_rsrc_t rs_src =
      __builtin_amdgcn_make_buffer_rsrc((void*)(reinterpret_cast<const unsigned char*>(p.src) - margin), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_wgt = __builtin_amdgcn_make_buffer_rsrc((void*)p.wgt, 0, 0x7fffffff, 0x00020000);

  const int kt0 = bz * p.kt_per_split;
  const int kt1 = min(kt0 + p.kt_per_split, p.ktiles);
  int cidx = kt0 % p.kc;
  int tap_r = (kt0 / p.kc) / p.kw, tap_s = (kt0 / p.kc) % p.kw;

  unsigned r_cur[XPASS], r_step[XPASS], r_mask[XPASS];
  int r_y[XPASS], r_x[XPASS], r_hw[XPASS];            // SMC: top-left source pixel of the window, source size
#pragma unroll
  for (int i = 0; i < XPASS; ++i) {
    const int gp = px0 + lrow + RPP * i;
    int seg = 0, img = 0, y = 0, x = 0;
    const bool ok = gp < totpx;
    if (ok) decode_pixel(p, gp, seg, img, y, x);
    const int sh = p.sh[seg], sw = p.sw[seg];
    if (SMC) {
      r_y[i] = ok ? y * p.stride - p.pad : -100000;    // not ok: every tap fails the bounds test
      r_x[i] = x * p.stride - p.pad;
      r_hw[i] = (sh << 16) | sw;
      r_cur[i] = (unsigned)(((int)(p.soff[seg]) + img * sh * sw + r_y[i] * sw + r_x[i]) * 16) + margin;
      r_step[i] = r_mask[i] = 0;
      continue;
    }
    const int row0 = p.mode == 0 ? y * p.stride - p.pad : y + p.pad;               // source row of tap r = 0
    const int col0 = p.mode == 0 ? x * p.stride - p.pad : x + p.pad - (p.kw - 1);  // leftmost source column
    unsigned m = 0;
    for (int r = 0; r < p.kh; ++r) {
      const int sy = p.mode == 0 ? row0 + r : row0 - r;
      if (ok && (unsigned)sy < (unsigned)sh) m |= 1u << r;
    }
    for (int s_ = 0; s_ < p.kw; ++s_) {
      const int sx = p.mode == 0 ? col0 + s_ : x + p.pad - s_;
      if (ok && (unsigned)sx < (unsigned)sw) m |= 0x100u << s_;
    }
    r_mask[i] = m;
    const unsigned pitch = (unsigned)(sw * p.lds);
    r_step[i] = p.mode == 0 ? pitch : 0u - pitch;
    const unsigned base = (unsigned)(((int)(p.soff[seg]) + img * sh * sw + row0 * sw + col0) * p.lds + chunk * 16) + margin;
    r_cur[i] = base + (unsigned)tap_r * r_step[i];
  }
  const unsigned w_voff = (unsigned)(lrow * (int)p.wrow + chunk * 16);
  const unsigned w_pass = (unsigned)(RPP * (int)p.wrow);
  unsigned w_soff = (unsigned)(co0 * (int)p.wrow + kt0 * 128);

  // ---- DMA of one K tile = LPT "pieces" per thread (XPASS pixel passes, then WPASS weight passes), issued a few at
  // a time between the MFMAs: a burst of all pieces right after the barrier fills the CU's address queue and
  // every wave then blocks on issue with an empty MFMA pipe.
  // Branch-free: past the last tile every lane goes out of range (zeros land in a slot nobody reads), so each
  // iteration issues exactly LPT DMA instructions and the vmcnt bookkeeping is a compile-time constant.
  constexpr int LPT = WPASS + XPASS;
  constexpr int P0 = (LPT + 1) / 3;                   // pieces issued right after the barrier (stage 3)
  constexpr int P1 = P0 + (LPT - P0 + 1) / 2;         // pieces [P0, P1) in stage 0, [P1, LPT) in stage 1
  int kt_next = kt0;               // K tile being fetched
  int ld_slot = 0;                 // ... and the ring slot it goes to
  auto pieces = [&](auto lo_c, auto hi_c) {
    constexpr int LO = decltype(lo_c)::value, HI = decltype(hi_c)::value;
    unsigned char* stage = smem + ld_slot * STAGE;
    const bool live = kt_next < kt1;
    const unsigned sel = live ? ((1u << tap_r) | (0x100u << tap_s)) : 0xffffffffu;
    const unsigned s_off = (unsigned)((p.mode == 0 ? tap_s : p.kw - 1 - tap_s) * p.lds + cidx * 128);
    const unsigned wv = live ? w_voff : 0x80000000u;
    int s_tr = 0, s_ts = 0;
    bool s_ok = false;
    if (SMC) {                    // this lane's tap of the K tile
      const int tap = kt_next * 8 + chunk;
      s_tr = tap / p.kw;
      s_ts = tap - s_tr * p.kw;
      s_ok = live && tap < p.kh * p.kw;
    }
#pragma unroll
    for (int j = LO; j < HI; ++j) {
      if (j < XPASS) {
        unsigned v, so;
        if (SMC) {
          const int sh = r_hw[j] >> 16, sw = r_hw[j] & 0xffff;
          const bool in = s_ok && (unsigned)(r_y[j] + s_tr) < (unsigned)sh && (unsigned)(r_x[j] + s_ts) < (unsigned)sw;
          v = in ? r_cur[j] + (unsigned)((s_tr * sw + s_ts) * 16) : 0x80000000u;
          so = 0;
        } else {
          v = (r_mask[j] & sel) == sel ? r_cur[j] : 0x80000000u;
          so = s_off;
        }
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_src, (lptr_t)(stage + TILE_W + (j * RPP + wave * 8) * 128), 16, v, so, 0, 0);
      } else {
        const int i = j - XPASS;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_wgt, (lptr_t)(stage + (i * RPP + wave * 8) * 128), 16, wv,
                                                 w_soff + i * w_pass, 0, 0);
      }
    }
    if (HI == LPT) {               // tile fully issued: advance to the next (r, s, channel-block) and ring slot
      ++kt_next;
      ld_slot = (ld_slot + 1 == NST) ? 0 : ld_slot + 1;
      w_soff += BK * 2;
      const bool c_wrap = cidx + 1 == p.kc;
      const bool s_wrap = c_wrap && tap_s + 1 == p.kw;
      cidx = c_wrap ? 0 : cidx + 1;
      tap_s = s_wrap ? 0 : (c_wrap ? tap_s + 1 : tap_s);
      tap_r += s_wrap ? 1 : 0;
      const unsigned adv = s_wrap ? 0xffffffffu : 0u;
#pragma unroll
      for (int i = 0; i < XPASS; ++i) r_cur[i] += r_step[i] & adv;
    }
  };
  using c0_t = std::integral_constant<int, 0>;
  using cp0_t = std::integral_constant<int, P0>;
  using cp1_t = std::integral_constant<int, P1>;
  using clpt_t = std::integral_constant<int, LPT>;

  f32x16 acc[2][PT];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < PT; ++b)
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[a][b][j] = 0.f;

  const int frow = lane & 31, fhalf = lane >> 5;
  const int fswz = (frow >> 1) & 7;
  const int a_off = (wave_co * 64 + frow) * 128;
  const int b_off = TILE_W + (wave_px * (32 * PT) + frow) * 128;
  // a K tile is 128 fp8 elements = 2 k-steps of v_mfma_scale_f32_32x32x64_f8f6f4 (fp8 e4m3 x fp8 e4m3, block scales 2^0): a lane's
  // operand is 32 consecutive K elements of its row = two 16-byte chunks of the swizzled 128-byte LDS row
  v8i32 fa[2][2], fb[2][PT];
  auto rd32 = [&](const unsigned char* row, int kk) -> v8i32 {
    const int c0 = ((4 * kk + 2 * fhalf) ^ fswz) << 4, c1 = ((4 * kk + 2 * fhalf + 1) ^ fswz) << 4;
    const u32x4 lo = *reinterpret_cast<const u32x4*>(row + c0), hi = *reinterpret_cast<const u32x4*>(row + c1);
    v8i32 r = {(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
    return r;
  };
  auto lds_read = [&](const unsigned char* base, int kk, int f) {
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) fa[f][ct] = rd32(base + a_off + ct * 32 * 128, kk);
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) fb[f][pt] = rd32(base + b_off + pt * 32 * 128, kk);
  };
  auto mma_half = [&](int f, int ct) {
#pragma unroll
    for (int pt = 0; pt < PT; ++pt)
      acc[ct][pt] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fa[f][ct], fb[f][pt], acc[ct][pt], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
  };
  auto mma = [&](int f) {
    mma_half(f, 0);
    mma_half(f, 1);
  };

  static_assert((NST - 1) * LPT <= 63, "vmcnt range");
  // prologue: NST-1 whole tiles + the first pieces of the NST-th
#pragma unroll
  for (int s = 0; s < NST - 1; ++s) pieces(c0_t{}, clpt_t{});
  pieces(c0_t{}, cp0_t{});
  wait_vmcnt<(NST - 2) * LPT + P0>();
  __builtin_amdgcn_s_barrier();
  lds_read(smem, 0, 0);
  int slot = 0;
  for (int kt = kt0; kt < kt1 - 1; ++kt) {
    const unsigned char* base = smem + slot * STAGE;
    const int nslot = (slot + 1 == NST) ? 0 : slot + 1;
    lds_read(base, 1, 1);
    pieces(cp0_t{}, clpt_t{});
    mma(0);
    mma_half(1, 0);
    sched_stage<2 * PT, 2 * (2 + PT), LPT - P0>();
    sched_stage<PT, 0, 0>();
    __builtin_amdgcn_sched_barrier(0);     // keep these MFMAs in front of the waits below
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's reads of tile kt are in registers
    wait_vmcnt<(NST - 2) * LPT>();         // tile kt+1 landed (tiles kt+2 .. kt+NST-1 may stay in flight)
    __builtin_amdgcn_s_barrier();          // ... and both hold for every wave
    lds_read(smem + nslot * STAGE, 0, 0);
    pieces(c0_t{}, cp0_t{});               // start refilling the slot tile kt just vacated with tile kt+NST
    mma_half(1, 1);
    __builtin_amdgcn_sched_group_barrier(0x100, 2 * (2 + PT), 0);
    sched_stage<PT, 0, P0>();
    slot = nslot;
  }
  {                                        // last tile
    const unsigned char* base = smem + slot * STAGE;
    lds_read(base, 1, 1);
    mma(0);
    mma(1);
    sched_stage<2 * PT, 2 * (2 + PT), 0>();
  }
  wait_vmcnt<0>();                         // the out-of-range tail DMAs still write (zeros) into the ring

  if (p.splits > 1) {                      // split-K: raw fp32 partial tile -> workspace [split][pixel][cd_pad]
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
      const int gp = px0 + wave_px * (32 * PT) + pt * 32 + (lane & 31);
      if (gp >= totpx) continue;
      float* row = p.ws + ((long long)bz * totpx + gp) * p.cd_pad;
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int co = co0 + wave_co * 64 + ct * 32 + 8 * g + 4 * (lane >> 5);
          f32x4 o = {acc[ct][pt][4 * g], acc[ct][pt][4 * g + 1], acc[ct][pt][4 * g + 2], acc[ct][pt][4 * g + 3]};
          *reinterpret_cast<f32x4*>(row + co) = o;
        }
    }
    return;
  }

  // ---- epilogue (shared with the bf16 kernel)
  conv_tile_epilogue<BCO, BPX, WCO, WPX, 2, PT, NST * STAGE>(p, acc, smem, co0, px0, totpx, [](int) {});
}

}  // namespace

// ================================================================================================
// host side
// ================================================================================================
namespace {
// DMA-to-LDS tile configurations {BCO, BPX, workgroups per CU, ring depth}
struct TileCfg { int bco, bpx, occ, nst, wpx; };     // wpx: pixel-waves of the pipelined kernel (epilogue staging = 32*wpx pixels)
constexpr int kNumCfg = 8;
const TileCfg kCfgs[kNumCfg] = {{256, 192, 1, 2, 2}, {256, 128, 1, 3, 2}, {128, 256, 1, 3, 4}, {128, 128, 2, 2, 4}, {64, 256, 2, 2, 8},
                                {128, 64, 2, 3, 2},
                                // small tiles for the layers with few pixels (layer3/4: 8 400 / 2 100 pixels at N = 2): enough
                                // workgroups to use every CU without split-K partials, several resident per CU
                                {64, 64, 3, 3, 2}, {64, 128, 2, 3, 4}};

// strided data-gradients gather with per-tap divisibility tests: only the v2 kernel's general address path does that
inline bool conv_v2_only(const dsl_conv_desc* d) { return d->mode == 1 && d->stride > 1; }

// Launch-time cost model (microseconds) of one (tile config, split-K factor) choice.  Calibrated on MI355X
// (tools/ablate_pipe.py, tools/bench_conv.py; the constants were fitted to the forced-config sweep of
// `tools/bench_conv.py 2 0,1,2,3,4,5,6`: the model picks the measured-best tile on 12 of its 13 shapes, 1.9 us total
// regret): per K tile a workgroup needs bco*bpx/32 MFMA cycles of its CU and
// (bco+bpx)*128 B through the CU's 64 B/clk vector-memory path (~54 B/clk measured); co-resident workgroups share
// both; the two overlap imperfectly.  Output and split-K partial traffic are HBM-rate terms.
double conv_cost_us(int ci, long long px, int cd_pad, int ktiles, int sp, bool out_f32) {
  const TileCfg& c = kCfgs[ci];
  // per-config efficiency of the K loop (the 8-wave 128x128 tile keeps 2 waves per SIMD even alone on a CU)
  static const double kEff[kNumCfg] = {1.0, 1.0, 1.0, 0.7, 0.95, 0.9, 1.3, 1.25};     // 6, 7: measured best on one shape of tools/bench_conv.py only
  const long long wgs = (long long)(cd_pad / c.bco) * ((px + c.bpx - 1) / c.bpx) * sp;
  const long long slots = 256LL * c.occ;
  const long long rounds = (wgs + slots - 1) / slots;
  const long long per_cu = (wgs + 255) / 256;
  const double share = (double)(per_cu < c.occ ? per_cu : c.occ);     // workgroups sharing a CU in a round
  const double mfma = c.bco * c.bpx / 32.0, dma = (c.bco + c.bpx) * 128 / 54.0;
  const double tile = share * (1.15 * (mfma > dma ? mfma : dma) + 0.5 * (mfma > dma ? dma : mfma)) * kEff[ci];
  // per-workgroup fill + epilogue: grows with the tile, and co-resident workgroups overlap each other's
  const double fixed = (1000.0 + 0.05 * c.bco * c.bpx) * (share > 1.0 ? share / 2.0 : share);
  const int kt = (ktiles + sp - 1) / sp;
  double t = 4.0 + rounds * (kt * tile + fixed) / 2240.0;             // launch, then the rounds at ~2.24 GHz
  t += (double)px * cd_pad * (out_f32 ? 4 : 2) / 4.0e6;                // output write
  if (sp > 1) t += 3.0 + 2.0 * sp * px * cd_pad * 4 / 4.0e6;          // partial write + read, second launch
  return t;
}

// picks the tile configuration (-1 = v1 kernel) and the split-K factor for a conv
void conv_choose(const dsl_conv_desc* d, long long px, int ktiles, int* pick_out, int* splits_out) {
  const bool smallc = (d->flags & DSL_CONV_SMALL_C) != 0;
  const int force = (d->flags >> 8) & 15;          // test hook: 1..8 = tile config, 15 = v1 kernel
  const int force_split = (d->flags >> 12) & 15;   // test hook: split-K factor
  int pick = -1, splits = 1;
  long long src_px = 0;
  for (int sg = 0; sg < d->nseg; ++sg) src_px += (long long)d->n * d->sh[sg] * d->sw[sg];
  // the DMA kernels address the source with 32-bit buffer offsets and per-axis tap masks
  const long long lds_ = d->lds > 0 ? d->lds : d->cs;
  const bool dma_ok = conv_v2_only(d) || (d->kh <= 8 && d->kw <= 8 && src_px * lds_ * 2 + (long long)d->kw * lds_ * 2 < 0x7fff0000LL);
  const bool smallc_pipe = smallc && d->cd_pad % 64 == 0;    // stem: pipelined kernel, 64-cout tile
  const bool v1_only = ((smallc && !smallc_pipe) || (d->flags & DSL_CONV_RELU_IN) || !dma_ok) && !(d->flags & DSL_CONV_FP8);
  if (!v1_only && force != 15) {
    double best = 1e300;
    const bool out_f32 = (d->flags & DSL_CONV_OUT_F32) != 0;
    for (int c = 0; c < kNumCfg; ++c) {
      if (d->cd_pad % kCfgs[c].bco) continue;
      if (force >= 1 && force <= kNumCfg && force - 1 != c) continue;
      if (c >= 5 && conv_v2_only(d)) continue;       // the small tiles exist for the pipelined kernel only
      if (d->gn_x && c > 1) continue;                // backward GroupNorm records: the 256-cout tiles carry them (conv_gn_ok)
      if (smallc && c != 4) continue;                // the 8-channel-source variant is instantiated for the 64x256 tile
      if ((d->flags & DSL_CONV_FP8) && c != 0 && c != 1 && c != 3) continue;     // fp8: instantiated for 256x192, 256x128, 128x128
      for (int sp = 1; sp <= 16; ++sp) {
        if (sp > 1 && (smallc || (d->flags & DSL_CONV_FP8) || d->gn_ws)) break;
        if (sp > 1 && (!d->workspace || sp > ktiles / 2 || (size_t)sp * px * d->cd_pad * 4 > d->workspace_bytes)) break;
        if (force_split > 1 && sp != force_split) continue;
        const double t = conv_cost_us(c, px, d->cd_pad, ktiles, sp, out_f32);
        if (t < best) { best = t; pick = c; splits = sp; }
      }
    }
    if (pick < 0 && force_split > 1) {               // forced split not feasible: ignore it
      for (int c = 0; c < kNumCfg; ++c) {
        if (d->cd_pad % kCfgs[c].bco || (force >= 1 && force <= kNumCfg && force - 1 != c)) continue;
        if (c >= 5 && conv_v2_only(d)) continue;
        const double t = conv_cost_us(c, px, d->cd_pad, ktiles, 1, out_f32);
        if (t < best) { best = t; pick = c; splits = 1; }
      }
    }
  }
  *pick_out = pick;
  *splits_out = splits;
}

// algorithmic work of one launch: real channels (3 for the NHWC8 image, cs_real for padded gradient rows), every tensor
// read / written once
double conv_real_cin(const dsl_conv_desc* d) {
  return (d->flags & DSL_CONV_SMALL_C) ? 3.0 : (double)(d->cs_real > 0 ? d->cs_real : d->cs);
}
double conv_algo_flops(const dsl_conv_desc* d, long long px) {
  return 2.0 * px * (double)d->cd * d->kh * d->kw * conv_real_cin(d);
}
double conv_algo_bytes(const dsl_conv_desc* d, long long px) {
  double src_px = 0, dst_px = 0;
  for (int s = 0; s < d->nseg; ++s) {
    src_px += (double)d->n * d->sh[s] * d->sw[s];
    dst_px += (double)d->n * d->dh[s] * d->dw[s];
  }
  const double es = (d->flags & DSL_CONV_FP8) ? 1.0 : 2.0;
  double b = src_px * conv_real_cin(d) * es + (double)d->cd * d->kh * d->kw * conv_real_cin(d) * es +
             dst_px * d->cd * ((d->flags & DSL_CONV_OUT_F32) ? 4.0 : 2.0);
  if (d->addend) b += dst_px * d->cd * 2.0;
  if (d->mask) b += dst_px * d->cd * 2.0;
  return b;
}

long long conv_pixels(const dsl_conv_desc* d) {
  long long px = 0;
  for (int s = 0; s < d->nseg; ++s) px += (long long)d->n * d->gh[s] * d->gw[s];
  return px;
}

// Would a launch of `d` with gn_ws set leave the GroupNorm records?  Only the pipelined kernel's in-register ("pure") epilogue
// writes them: bf16 output on the compute grid's own pixels, nothing added, one launch (no split-K).
bool conv_gn_ok(const dsl_conv_desc* d) {
  if (d->nseg < 1 || d->nseg > DSL_MAX_SEG) return false;
  if (d->flags & (DSL_CONV_SMALL_C | DSL_CONV_OUT_F32 | DSL_CONV_ADD_UPSAMPLE | DSL_CONV_RELU_IN)) return false;
  if ((d->flags & DSL_CONV_FP8) && d->gn_x) return false;        // (the fp8 kernel shares the forward records' epilogue, not the backward ones)
  if (d->cs % 64 || d->cd % 8 || d->cd_pad % 64 || d->os != 1 || d->addend) return false;
  if (d->mask && ((d->flags & DSL_CONV_MASK_FIRST) || ((d->flags & DSL_CONV_MASK_LAST) && (d->flags & DSL_CONV_RELU_OUT)))) return false;
  for (int s = 0; s < d->nseg; ++s)
    if (d->gh[s] != d->dh[s] || d->gw[s] != d->dw[s]) return false;
  if (conv_v2_only(d)) return false;
  dsl_conv_desc t = *d;
  t.gn_ws = (void*)1;
  int pick, splits;
  conv_choose(&t, conv_pixels(d), d->kh * d->kw * (d->cs / 64), &pick, &splits);
  if (pick < 0 || splits != 1) return false;
  if (d->gn_x) {      // the backward records: instantiated for the 256-cout tiles only (256 x 192, 256 x 128).  Their workgroups own a
    // CU anyway (114 / 147 KB of LDS); the 128 x 128 tile with the ~60 extra registers (158) keeps ONE workgroup per CU instead of
    // two and the launch takes twice as long (N = 3 head: 100 -> 212 us, profiles/r04_rla_timeline.txt) - there the separate pass stays
    if (pick != 0 && pick != 1) return false;
    if (d->ldd != d->cd) return false;
  }
  return true;
}
}  // namespace

extern "C" int dsl_conv2d_gn_fusable(const dsl_conv_desc* d) { return d && conv_gn_ok(d) ? 1 : 0; }

extern "C" size_t dsl_conv2d_workspace_bytes(const dsl_conv_desc* d) {
  if (!d || d->nseg < 1 || d->nseg > DSL_MAX_SEG || (d->flags & DSL_CONV_SMALL_C) || d->cs % 64) return 0;
  dsl_conv_desc t = *d;
  t.workspace = (void*)1;
  t.workspace_bytes = (size_t)1 << 40;
  int pick, splits;
  const long long px = conv_pixels(d);
  conv_choose(&t, px, d->kh * d->kw * (d->cs / 64), &pick, &splits);
  return splits > 1 ? (size_t)splits * px * d->cd_pad * 4 : 0;
}

extern "C" int dsl_conv2d(const dsl_conv_desc* d, void* stream) {
  DSL_CHECK(d != nullptr, "dsl_conv2d: null descriptor");
  DSL_CHECK(d->nseg >= 1 && d->nseg <= DSL_MAX_SEG, "dsl_conv2d: nseg=%d out of range", d->nseg);
  DSL_CHECK(d->src && d->wgt && d->dst, "dsl_conv2d: null tensor pointer");
  const bool smallc = (d->flags & DSL_CONV_SMALL_C) != 0;
  const bool fp8 = (d->flags & DSL_CONV_FP8) != 0;
  if (smallc)
    DSL_CHECK(d->cs == 8 && d->mode == 0, "dsl_conv2d: SMALL_C needs cs == 8, forward mode");
  else if (fp8)
    DSL_CHECK(d->cs % 128 == 0 && d->cs > 0 && d->mode == 0 && !(d->flags & DSL_CONV_RELU_IN) && d->scale,
              "dsl_conv2d: FP8 needs forward mode, cs %% 128 == 0 (cs=%d) and the dequantisation `scale` vector", d->cs);
  else
    DSL_CHECK(d->cs % 64 == 0 && d->cs > 0, "dsl_conv2d: source channels %d not a multiple of 64", d->cs);
  DSL_CHECK(d->cd_pad % 64 == 0 && d->cd <= d->cd_pad && d->cd > 0, "dsl_conv2d: bad cd=%d cd_pad=%d", d->cd, d->cd_pad);
  DSL_CHECK(d->ldd % 4 == 0 && d->ldd >= d->cd, "dsl_conv2d: ldd=%d must be a multiple of 4 and >= cd", d->ldd);
  DSL_CHECK(d->stride >= 1 && d->os >= 1 && d->kh >= 1 && d->kw >= 1, "dsl_conv2d: bad stride/os/kernel");
  DSL_CHECK(!(d->addend) || d->lda % 4 == 0, "dsl_conv2d: lda must be a multiple of 4");
  DSL_CHECK(!(d->mask) || d->ldm % 4 == 0, "dsl_conv2d: ldm must be a multiple of 4");
  ConvK k;
  memset(&k, 0, sizeof(k));
  k.nseg = d->nseg;
  k.n = d->n;
  long long so = 0, dof = 0, ao = 0;
  int px = 0;
  for (int s = 0; s < d->nseg; ++s) {
    k.gh[s] = d->gh[s]; k.gw[s] = d->gw[s];
    k.sh[s] = d->sh[s]; k.sw[s] = d->sw[s];
    k.dh[s] = d->dh[s]; k.dw[s] = d->dw[s];
    const bool up = (d->flags & DSL_CONV_ADD_UPSAMPLE) != 0;
    k.ah[s] = up ? d->ah[s] : d->dh[s];
    k.aw[s] = up ? d->aw[s] : d->dw[s];
    DSL_CHECK(d->gh[s] > 0 && d->gw[s] > 0 && d->sh[s] > 0 && d->sw[s] > 0 && d->dh[s] > 0 && d->dw[s] > 0,
              "dsl_conv2d: empty segment %d", s);
    DSL_CHECK((d->gh[s] - 1) * d->os < d->dh[s] && (d->gw[s] - 1) * d->os < d->dw[s],
              "dsl_conv2d: compute grid x os exceeds destination in segment %d", s);
    k.pxstart[s] = px;
    k.soff[s] = so; k.doff[s] = dof; k.aoff[s] = ao;
    px += d->n * d->gh[s] * d->gw[s];
    so += (long long)d->n * d->sh[s] * d->sw[s];
    dof += (long long)d->n * d->dh[s] * d->dw[s];
    ao += (long long)d->n * k.ah[s] * k.aw[s];
  }
  k.pxstart[d->nseg] = px;
  k.ident = (d->os == 1 && !(d->flags & DSL_CONV_ADD_UPSAMPLE)) ? 1 : 0;
  for (int s = 0; s < d->nseg; ++s)
    if (d->gh[s] != d->dh[s] || d->gw[s] != d->dw[s]) k.ident = 0;
  k.cs = d->cs; k.cd = d->cd; k.ldd = d->ldd; k.lda = d->lda; k.ldm = d->ldm;
  // the in-register addend epilogue (conv_tile_epilogue): the addend tile goes to LDS by DMA with 32-bit buffer offsets
  bool identd = d->os == 1;               // destination pixel == compute-grid pixel (whatever the addend's grid is)
  for (int s = 0; s < d->nseg; ++s)
    if (d->gh[s] != d->dh[s] || d->gw[s] != d->dw[s]) identd = false;
  k.addfast = (d->addend && identd && !(d->flags & DSL_CONV_EPI_STAGED) && d->lda % 8 == 0 && d->ldd % 8 == 0 &&
               (!d->mask || d->ldm % 8 == 0) && ((uintptr_t)d->addend & 15) == 0 && ((uintptr_t)d->dst & 15) == 0 &&
               (dof > ao ? dof : ao) * (long long)d->lda * 2 < 0x7fff0000LL) ? ((d->flags & DSL_CONV_ADD_UPSAMPLE) ? 2 : 1) : 0;
  k.lds = d->lds > 0 ? d->lds : d->cs;
  DSL_CHECK(k.lds >= d->cs && k.lds % (fp8 ? 16 : 8) == 0, "dsl_conv2d: lds=%d must be >= cs=%d and a multiple of 16 bytes", k.lds, d->cs);
  k.kh = d->kh; k.kw = d->kw; k.stride = d->stride; k.pad = d->pad; k.mode = d->mode; k.os = d->os;
  k.flags = d->flags;
  if (smallc) {
    k.ktiles = (d->kh * d->kw + 7) / 8;
    k.kc = 1;
  } else if (fp8) {
    k.kc = d->cs / 128;                 // a K tile is 128 one-byte elements: the same 128-byte LDS rows
    k.ktiles = d->kh * d->kw * k.kc;
  } else {
    k.kc = d->cs / 64;
    k.ktiles = d->kh * d->kw * k.kc;
  }
  k.wrow = (long long)k.ktiles * (fp8 ? 128 : BK);
  k.src = (const uint16_t*)d->src; k.wgt = (const uint16_t*)d->wgt; k.dst = d->dst;
  k.scale = d->scale; k.bias = d->bias;
  k.addend = (const uint16_t*)d->addend; k.mask = (const uint16_t*)d->mask;
  DSL_CHECK(!d->gn_ws || conv_gn_ok(d), "dsl_conv2d: gn_ws is set but this launch cannot write GroupNorm records "
            "(ask dsl_conv2d_gn_fusable first)");
  k.gnws = (float*)d->gn_ws;
  if (d->gn_ws && d->gn_x) {
    DSL_CHECK(d->gn_gamma && d->gn_beta && d->gn_stats && d->ldd == d->cd, "dsl_conv2d: backward GroupNorm records need gn_gamma, "
              "gn_beta, gn_stats and a dense destination (ldd=%d cd=%d)", d->ldd, d->cd);
    k.gnx = (const uint16_t*)d->gn_x; k.gngamma = d->gn_gamma; k.gnbeta = d->gn_beta; k.gnstats = d->gn_stats;
  }

  hipStream_t st = (hipStream_t)stream;
  // ---- kernel / tile selection -------------------------------------------------------------------
  int pick, splits;
  conv_choose(d, px, k.ktiles, &pick, &splits);
  if (pick >= 0) {
    const TileCfg& c = kCfgs[pick];
    k.splits = splits;
    k.kt_per_split = (k.ktiles + splits - 1) / splits;
    k.cd_pad = d->cd_pad;
    k.ws = (float*)d->workspace;
#ifdef DSL_ABLATE_BUILD
    { const char* e = getenv("DSL_ABLATE"); k.dbg = e ? atoi(e) : 0; }
#endif
#ifdef DSL_TRACE_BUILD
    { const char* e = getenv("DSL_TRACE_WG"); k.dbg = e ? atoi(e) : -1; }
#endif
    dim3 grid(d->cd_pad / c.bco, (px + c.bpx - 1) / c.bpx, splits);
    const bool force_v2_kernel = conv_v2_only(d);
    k.gx = (int)grid.x;
    k.gy = (int)grid.y;
    k.xcd_chunk = (int)((grid.x * grid.y * grid.z + 7) / 8);
    size_t lds = (size_t)c.nst * (c.bco + c.bpx) * 128;
    if (!force_v2_kernel) {                // the pipelined kernel stages its epilogue in LDS: 32*wpx pixel rows of fp32
      const size_t stg = (size_t)32 * c.wpx * (c.bco * 4 + 16);
      if (stg > lds) lds = stg;
    }
#ifdef DSL_TRACE_BUILD
    lds += 8 * 40 * 8 * 8 + 8 * 16 * 8;    // the stamp area behind the ring
#endif
    int prof = -1;
    if (dsl_prof_active()) prof = dsl_prof_begin(pick == 3 ? 0 : (pick == 0 ? 1 : 2), conv_algo_flops(d, px), st, conv_algo_bytes(d, px));
#define LAUNCH2(A, B, C_, D, S_)                                                                               \
  do {                                                                                                        \
    static bool attr_set = false;                                                                             \
    if (!attr_set) {                                                                                          \
      hipFuncSetAttribute((const void*)conv_glds_kernel<A, B, C_, D, S_>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                          (int)lds);                                                                          \
      attr_set = true;                                                                                        \
    }                                                                                                         \
    hipLaunchKernelGGL((conv_glds_kernel<A, B, C_, D, S_>), grid, dim3(64 * C_ * D), lds, st, k);              \
  } while (0)
#define LAUNCH3(A, B, C_, D, S_)                                                                               \
  do {                                                                                                        \
    static bool attr_set3 = false;                                                                            \
    if (!attr_set3) {                                                                                         \
      hipFuncSetAttribute((const void*)conv_pipe_kernel<A, B, C_, D, S_>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                          (int)lds);                                                                          \
      attr_set3 = true;                                                                                       \
    }                                                                                                         \
    hipLaunchKernelGGL((conv_pipe_kernel<A, B, C_, D, S_>), dim3(8 * k.xcd_chunk), dim3(64 * C_ * D), lds, st, k); \
  } while (0)
#define LAUNCH3G(A, B, C_, D, S_)                                                                              \
  do {                                                                                                        \
    static bool attr_set3g = false;                                                                           \
    if (!attr_set3g) {                                                                                        \
      hipFuncSetAttribute((const void*)conv_pipe_kernel<A, B, C_, D, S_, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                          (int)lds);                                                                          \
      attr_set3g = true;                                                                                      \
    }                                                                                                         \
    hipLaunchKernelGGL((conv_pipe_kernel<A, B, C_, D, S_, false, true>), dim3(8 * k.xcd_chunk), dim3(64 * C_ * D), lds, st, k); \
  } while (0)
    if (fp8) {
      DSL_CHECK(!force_v2_kernel && (pick == 0 || pick == 1 || pick == 3), "dsl_conv2d: no fp8 kernel for this shape (tile config %d)", pick);
#define LAUNCH8(A, B, C_, D, S_)                                                                               \
  do {                                                                                                        \
    static bool attr_8 = false;                                                                               \
    if (!attr_8) {                                                                                            \
      hipFuncSetAttribute((const void*)conv_f8_kernel<A, B, C_, D, S_>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                          (int)lds);                                                                          \
      attr_8 = true;                                                                                          \
    }                                                                                                         \
    hipLaunchKernelGGL((conv_f8_kernel<A, B, C_, D, S_>), dim3(8 * k.xcd_chunk), dim3(64 * C_ * D), lds, st, k); \
  } while (0)
      switch (pick) {
        case 0: LAUNCH8(256, 192, 4, 2, 2); break;
        case 1: LAUNCH8(256, 128, 4, 2, 3); break;
        default: LAUNCH8(128, 128, 2, 4, 2); break;
      }
#undef LAUNCH8
    } else if (force_v2_kernel) {
      switch (pick) {
        case 0: LAUNCH2(256, 192, 4, 2, 2); break;
        case 1: LAUNCH2(256, 128, 4, 2, 3); break;
        case 2: LAUNCH2(128, 256, 2, 4, 3); break;
        case 3: LAUNCH2(128, 128, 2, 2, 2); break;
        default: LAUNCH2(64, 256, 1, 4, 2); break;
      }
    } else {
      if (k.gnx) {                   // (conv_gn_ok: one of the two tiles below)
        if (pick == 0) LAUNCH3G(256, 192, 4, 2, 2); else LAUNCH3G(256, 128, 4, 2, 3);
      } else
      switch (pick) {
        case 0: LAUNCH3(256, 192, 4, 2, 2); break;
        case 1: LAUNCH3(256, 128, 4, 2, 3); break;
        case 2: LAUNCH3(128, 256, 2, 4, 3); break;
        case 3: LAUNCH3(128, 128, 2, 4, 2); break;
        case 4:
          if (smallc) {
            static bool a4 = false;
            if (!a4) { hipFuncSetAttribute((const void*)conv_pipe_kernel<64, 256, 1, 8, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); a4 = true; }
            hipLaunchKernelGGL((conv_pipe_kernel<64, 256, 1, 8, 2, true>), dim3(8 * k.xcd_chunk), dim3(512), lds, st, k);
          } else {
            LAUNCH3(64, 256, 1, 8, 2);
          }
          break;
        case 5: LAUNCH3(128, 64, 2, 2, 3); break;
        case 6: LAUNCH3(64, 64, 1, 2, 3); break;
        default: LAUNCH3(64, 128, 1, 4, 3); break;
      }
    }
#undef LAUNCH2
#undef LAUNCH3
#undef LAUNCH3G
    dsl_prof_end(prof, st);
    if (splits > 1) {
      const long long total = (long long)px * (d->cd_pad / 4);
      int blocks = (int)((total + 255) / 256);
      if (blocks > 4096) blocks = 4096;
      hipLaunchKernelGGL(conv_splitk_epilogue_kernel, dim3(blocks), dim3(256), 0, st, k);
    }
    DSL_LAUNCH_CHECK("conv_glds_kernel");
    return 0;
  }
  const int bco = (d->cd_pad % 128 == 0) ? 128 : 64;
  dim3 grid(d->cd_pad / bco, (px + BPX - 1) / BPX);
  const size_t lds = 2 * (size_t)(bco + BPX) * BK * 2;
#define LAUNCH(BCO_, SC_)                                                                         \
  do {                                                                                            \
    static bool attr_set = false;                                                                 \
    if (!attr_set) {                                                                              \
      hipFuncSetAttribute((const void*)conv_gemm_kernel<BCO_, SC_>,                               \
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                  \
      attr_set = true;                                                                            \
    }                                                                                             \
    hipLaunchKernelGGL((conv_gemm_kernel<BCO_, SC_>), grid, dim3(256), lds, st, k);               \
  } while (0)
  int prof = -1;
  if (dsl_prof_active()) prof = dsl_prof_begin(2, conv_algo_flops(d, px), st, conv_algo_bytes(d, px));
  if (bco == 128) {
    if (smallc) LAUNCH(128, true); else LAUNCH(128, false);
  } else {
    if (smallc) LAUNCH(64, true); else LAUNCH(64, false);
  }
  dsl_prof_end(prof, st);
#undef LAUNCH
  DSL_LAUNCH_CHECK("conv_gemm_kernel");
  return 0;
}


#ifdef DSL_TRACE_BUILD
// tools/trace_conv.py: the s_memtime stamps of the traced workgroup (DSL_TRACE_WG) of the last conv_pipe launch
extern "C" int dsl_debug_conv_trace(unsigned long long* host_out) {
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_conv_trace), sizeof(unsigned long long) * (8 * 40 * 8 + 8 * 16 + 8)) == hipSuccess ? 0 : -1;
}
#endif
