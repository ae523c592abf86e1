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

// LW > 0: LW extra "loader" waves issue every LDS-DMA piece; the WCO x WPX MFMA waves only read fragments and multiply.  Why
// (tools/trace_conv.py, s_memtime stamps of one workgroup): an MFMA wave that issues a DMA piece stalls 60 - 185 cycles at issue
// while the CU's vector-memory queue drains the other waves' pieces, and the in-order wave cannot issue the MFMAs behind it -
// seven pieces per wave per K tile kept the matrix pipe 64 % busy in the K loop of the 256 x 192 tile.
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
  // A ReLU mask applied LAST (the data gradients: round(v * m), m in {0, 1}) commutes with the rounding - m ? round(v) : +-0 with v's
  // sign - so it is applied to the staged bf16 words in the store loop, bit for bit what the fp32 path produces for finite v.
  if (ef.on && !ef.has_add && !ef.mask_first && !(ef.mask_last && ef.relu) && (p.cd & 7) == 0) {
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

// ================================================================================================
// conv_h4_kernel: conv_pipe_kernel's tile, operands, arithmetic and summation order with the K loop re-cut into HALF K tiles
// (32 channels: two MFMA k-steps) in a ring of FOUR half stages.  Why: with two whole stages the first fragment reads of K tile
// kt + 1 can only be issued behind the barrier that ends tile kt - nothing guarantees earlier that every wave's DMA of that tile
// has landed - and the matrix pipe drains behind every barrier until they return (tools/trace_conv.py: 350 - 500 of a tile's
// ~2 450 cycles, the held-back MFMAs cover 190 of them).  With four half stages the half AFTER the next one is already complete at
// a barrier, so the next half's first fragments are read BEFORE the barrier and its MFMAs issue right behind it; the DMA keeps the
// same look-ahead in time (a half is fetched two half-stages before it is needed = one whole K tile).
// LDS: per half stage a weight plane [BCO][64 B] and a pixel plane [BPXP][64 B] (BPXP: BPX rounded up to whole DMA passes of T / 4
// rows; the surplus rows are out-of-range lanes: zeros nobody reads); 16-byte chunk c of row r sits in slot c ^ ((r >> 2) & 3) -
// sixteen consecutive rows of a ds_read_b128 pass then cover all 64 banks once.
// ================================================================================================
template <int BCO, int BPX, int WCO, int WPX, bool GNB = false>
__global__ __launch_bounds__(64 * WCO * WPX) void conv_h4_kernel(const ConvK p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int T = 64 * WCO * WPX;
  constexpr int RPI = T / 4;                  // tile rows one DMA pass fills (4 lanes per 64-byte row)
  constexpr int WPASS = BCO / RPI, XPASS = (BPX + RPI - 1) / RPI;
  constexpr int BPXP = XPASS * RPI;
  constexpr int PLANE_W = BCO * 64;
  constexpr int HSTAGE = (BCO + BPXP) * 64;
  constexpr int NH = 4;
  constexpr int PT = BPX / WPX / 32, CT = BCO / WCO / 32;
  constexpr int LPT = WPASS + XPASS;
  static_assert(BCO % RPI == 0 && BCO == WCO * CT * 32 && BPX == WPX * PT * 32, "tile / thread mismatch");
  static_assert(LPT % 2 == 0 && 2 * LPT <= 63, "pieces per half stage");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wave_co = wave / WPX, wave_px = wave % WPX;
  const int wi = (int)(blockIdx.x & 7) * p.xcd_chunk + (int)(blockIdx.x >> 3);
  if ((int)(blockIdx.x >> 3) >= p.xcd_chunk || wi >= p.gx * p.gy * p.splits) return;
  const int bz = wi / (p.gx * p.gy);
  const int rem_t = wi - bz * (p.gx * p.gy);
  const int by = rem_t / p.gx;
  const int co0 = (rem_t - by * p.gx) * BCO;
  const int px0 = by * BPX;
  const int totpx = p.pxstart[p.nseg];
  const int lrow = tid >> 2;
  const int chunk = (tid & 3) ^ ((tid >> 4) & 3);       // source chunk (of the half's four) that belongs in LDS slot (tid & 3) of this row

  const unsigned margin = (unsigned)(p.kw * p.lds * 2);
  const __amdgpu_buffer_rsrc_t rs_src =
      __builtin_amdgcn_make_buffer_rsrc((void*)(reinterpret_cast<const unsigned char*>(p.src) - margin), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_wgt = __builtin_amdgcn_make_buffer_rsrc((void*)p.wgt, 0, 0x7fffffff, 0x00020000);

  const int kt0 = bz * p.kt_per_split;
  const int kt1 = min(kt0 + p.kt_per_split, p.ktiles);
  const int h0 = 2 * kt0, hend = 2 * kt1;
  int cidx = kt0 % p.kc;
  int tap_r = (kt0 / p.kc) / p.kw, tap_s = (kt0 / p.kc) % p.kw;

  unsigned r_cur[XPASS], r_step[XPASS], r_mask[XPASS];
#pragma unroll
  for (int i = 0; i < XPASS; ++i) {
    const int trow = lrow + RPI * i;
    const int gp = px0 + trow;
    int seg = 0, img = 0, y = 0, x = 0;
    const bool ok = gp < totpx && trow < BPX;
    if (ok) decode_pixel(p, gp, seg, img, y, x);
    const int sh = p.sh[seg], sw = p.sw[seg];
    const int row0 = p.mode == 0 ? y * p.stride - p.pad : y + p.pad;
    const int col0 = p.mode == 0 ? x * p.stride - p.pad : x + p.pad - (p.kw - 1);
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
  const unsigned w_pass = (unsigned)(RPI * (int)p.wrow) * 2u;
  unsigned w_soff = (unsigned)(co0 * (int)p.wrow + kt0 * BK) * 2u;

  int h_next = h0;                  // half tile being fetched
  unsigned t_sel = (1u << tap_r) | (0x100u << tap_s);
  unsigned t_soff = (unsigned)((p.mode == 0 ? tap_s : p.kw - 1 - tap_s) * p.lds * 2 + cidx * 128);
  unsigned t_wv = w_voff;
  int c_left = 2 * (p.kc - cidx);   // halves until the channel blocks wrap (tap advance)
  unsigned r_v[XPASS];
#pragma unroll
  for (int i = 0; i < XPASS; ++i) r_v[i] = (r_mask[i] & t_sel) == t_sel ? r_cur[i] : 0x80000000u;
  auto pieces = [&](auto lo_c, auto hi_c, const unsigned ld_off) {
    constexpr int LO = decltype(lo_c)::value, HI = decltype(hi_c)::value;
    unsigned char* stage = smem + ld_off;
#pragma unroll
    for (int j = LO; j < HI; ++j) {
      if (j < XPASS) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_src, (lptr_t)(stage + PLANE_W + (j * RPI + wave * 16) * 64), 16, (unsigned)r_v[j], (unsigned)t_soff, 0, 0);
      } else {
        const int i = j - XPASS;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_wgt, (lptr_t)(stage + (i * RPI + wave * 16) * 64), 16, (unsigned)t_wv, (unsigned)(w_soff + i * w_pass), 0, 0);
      }
    }
    if (HI == LPT) {                // half fully issued: advance to the next (r, s, channel block, half)
      ++h_next;
      w_soff += 64;
      if (--c_left != 0) {
        t_soff += 64;
      } else {
        c_left = 2 * p.kc;
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
      if (__builtin_expect(h_next >= hend, 0)) {      // past the last half: every lane out of range (zeros into a slot nobody reads)
        asm volatile("" ::: "memory");
        t_sel = 0xffffffffu;
        t_wv = 0x80000000u;
#pragma unroll
        for (int i = 0; i < XPASS; ++i) r_v[i] = 0x80000000u;
      }
    }
  };
  using c0_t = std::integral_constant<int, 0>;
  using cmid_t = std::integral_constant<int, LPT / 2>;
  using clpt_t = std::integral_constant<int, LPT>;

  f32x16 acc[CT][PT];
#pragma unroll
  for (int a = 0; a < CT; ++a)
#pragma unroll
    for (int b = 0; b < PT; ++b)
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[a][b][j] = 0.f;
  const int frow = lane & 31, fhalf = lane >> 5;
  const int fswz = (frow >> 2) & 3;
  const int a_off = (wave_co * (32 * CT) + frow) * 64;
  const int b_off = PLANE_W + (wave_px * (32 * PT) + frow) * 64;
  bf16x8 fa[2][CT], fb[2][PT];
  auto lds_read = [&](const unsigned char* base, int kk, int f) {
    const int coff = ((2 * kk + fhalf) ^ fswz) << 4;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) fa[f][ct] = *reinterpret_cast<const bf16x8*>(base + a_off + ct * 32 * 64 + coff);
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) fb[f][pt] = *reinterpret_cast<const bf16x8*>(base + b_off + pt * 32 * 64 + coff);
  };
  auto mma = [&](int f) {
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int pt = 0; pt < PT; ++pt)
        acc[ct][pt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[f][ct], fb[f][pt], acc[ct][pt], 0, 0, 0);
  };
  if (WCO * WPX == 8 && wave >= 4) __builtin_amdgcn_s_setprio(1);
  // prologue: three whole halves in flight; the first two have landed (for every wave: barrier) when the loop starts
#pragma unroll
  for (int s_ = 0; s_ < NH - 1; ++s_) pieces(c0_t{}, clpt_t{}, (unsigned)(s_ * HSTAGE));
  wait_vmcnt<LPT>();
  __builtin_amdgcn_s_barrier();
  lds_read(smem, 0, 0);
  unsigned slot = 0;                        // byte offset of the half stage being read
  unsigned fslot = (NH - 1) * HSTAGE;       // ... of the one half h + 3 goes to (vacated by half h - 1 at the last barrier)
#pragma nounroll
  for (int h = h0; h < hend; ++h) {
    const unsigned char* base = smem + slot;
    const unsigned nslot = (slot + HSTAGE == NH * HSTAGE) ? 0u : slot + HSTAGE;
    lds_read(base, 1, 1);
    pieces(c0_t{}, cmid_t{}, fslot);
    mma(0);
    lds_read(smem + nslot, 0, 0);           // the NEXT half's first fragments: it landed two barriers ago
    pieces(cmid_t{}, clpt_t{}, fslot);
    mma(1);
    sched_stage<CT * PT, CT + PT, LPT / 2>();
    sched_stage<CT * PT, CT + PT, LPT - LPT / 2>();
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // own reads of half h done (its slot may be refilled), next fragments in registers
    wait_vmcnt<LPT>();                      // own pieces of half h + 2 landed (half h + 3 may stay in flight)
    __builtin_amdgcn_s_barrier();           // ... for every wave
    fslot = slot;
    slot = nslot;
  }
  wait_vmcnt<0>();                          // the out-of-range tail DMAs still write (zeros) into the ring

  if (p.splits > 1) {                       // split-K: raw fp32 partial tile -> workspace [split][pixel][cd_pad]
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
  conv_tile_epilogue<BCO, BPX, WCO, WPX, CT, PT, NH * HSTAGE, GNB>(p, acc, smem, co0, px0, totpx, [](int) {});
}

// LDS the epilogue may use: the K loop's ring, or - the 256 x 256 tile, whose bf16 rows do not fit in its two-stage ring - the staged
// tile plus the GroupNorm reduction scratch (the host sizes the launch's LDS the same way)
constexpr int conv_epi_ring(int bco, int bpx, int ring, int t) {
  return (bco == 256 && bpx == 256 && bpx * (bco * 2 + 16) + t * 8 > ring) ? bpx * (bco * 2 + 16) + t * 8 : ring;
}

template <int BCO, int BPX, int WCO, int WPX, int NST, bool SMC = false, int HB = 1, int LW = 0, bool GNB = false>
__global__ __launch_bounds__(64 * (WCO * WPX + LW)) void conv_pipe_kernel(const ConvK p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int T = 64 * WCO * WPX;          // MFMA ("consumer") threads
  constexpr int TD = LW ? 64 * LW : T;       // threads that issue the DMA
  constexpr int RPP = TD / 8;                // tile rows filled per pass (8 lanes per 128-byte row)
  constexpr int WPASS = BCO / RPP, XPASS = BPX / RPP;
  constexpr int TILE_W = BCO * 128;
  constexpr int STAGE = (BCO + BPX) * 128;
  constexpr int PT = BPX / WPX / 32;         // 32-pixel MFMA tiles per wave
  constexpr int CT = BCO / WCO / 32;         // 32-cout MFMA tiles per wave (2 for the 8-wave tiles; 4 = the "tall wave" variants:
                                             // LDS bytes read per MFMA are (CT + PT) / (CT * PT) KB, the bound of the large tiles)
  static_assert(BCO == WCO * CT * 32 && CT >= 1, "cout tiles per wave");
  // HB: cout tiles of the K tile's LAST k-step whose MFMAs are held back across the barrier (their fragments are in registers):
  // they are what the matrix pipe runs while the first reads of the next tile are in flight
  static_assert(HB >= 1 && HB <= CT, "held-back cout tiles");
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
  const bool loader = LW > 0 && wave >= WCO * WPX;      // wave-uniform
  const int dt = LW > 0 ? tid - T : tid;                // index among the DMA threads (negative in an MFMA wave when LW > 0: unused)
  const int dwave = LW > 0 ? wave - WCO * WPX : wave;
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
    if (LW > 0 && !loader) {               // MFMA waves of the loader variant carry no DMA state
      r_cur[i] = r_step[i] = r_mask[i] = 0;
      r_y[i] = r_x[i] = r_hw[i] = 0;
      continue;
    }
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
  if constexpr (LW > 0) {
    if (loader) {
      // ---- loader wave: tile t goes to ring slot (t - kt0) % NST as soon as the barrier that retires the slot's previous
      // tenant has passed; before barrier #(t - kt0) it waits until tile t has landed (counted vmcnt: later tiles stay in flight)
      int issued = kt0;                    // first tile not yet issued
      unsigned l_off = 0;                  // ... and the ring slot it goes to
      auto land = [&](int need) {          // every piece of tiles <= need has landed
        const int later = issued - 1 - need;
        if (later <= 0) wait_vmcnt<0>();
        else if (later == 1) wait_vmcnt<LPT>();
        else wait_vmcnt<(NST > 2 ? 2 : 1) * LPT>();
      };
      static_assert(NST <= 4, "land() distinguishes up to two tiles in flight behind the awaited one");
#pragma unroll
      for (int s_ = 0; s_ < NST - 1; ++s_)
        if (issued < kt1) { pieces(c0_t{}, clpt_t{}, l_off); ++issued; l_off = (l_off + STAGE == NST * STAGE) ? 0u : l_off + STAGE; }
      land(kt0);
      __builtin_amdgcn_s_barrier();
      for (int kt = kt0; kt < kt1 - 1; ++kt) {
        if (issued < kt1) { pieces(c0_t{}, clpt_t{}, l_off); ++issued; l_off = (l_off + STAGE == NST * STAGE) ? 0u : l_off + STAGE; }
        land(kt + 1);
        __builtin_amdgcn_s_barrier();
      }
      return;
    }
  }
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
  if constexpr (LW == 0) {
#pragma unroll
    for (int s = 0; s < NST - 1; ++s) pieces(c0_t{}, clpt_t{}, (unsigned)(s * STAGE));
    pieces(c0_t{}, cp0_t{}, (unsigned)((NST - 1) * STAGE));
    wait_vmcnt<(NST - 2) * LPT + P0>();
  }
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
    if constexpr (LW == 0) pieces(cp0_t{}, cp1_t{}, pslot);
    mma(0);
    lds_read(base, 2, 0);
    if constexpr (LW == 0) pieces(cp1_t{}, clpt_t{}, pslot);
    mma(1);
    lds_read(base, 3, 1);
    mma(0);
    mma_upto(1, CT - HB);
    sched_stage<CT * PT, CT + PT, LW ? 0 : P1 - P0>();
    sched_stage<CT * PT, CT + PT, LW ? 0 : LPT - P1>();
    sched_stage<CT * PT, CT + PT, 0>();
    sched_stage<(CT - HB) * PT, 0, 0>();
    __builtin_amdgcn_sched_barrier(0);     // keep these MFMAs in front of the waits below
    TR(1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's reads of tile kt are in registers (issued >= PT MFMAs ago)
    TR(2);
    if constexpr (LW == 0) wait_vmcnt<(NST - 2) * LPT>();     // tile kt+1 landed (tiles kt+2 .. kt+NST-1 may stay in flight)
    TR(3);
#ifdef DSL_ABLATE_BUILD
    if (!(p.dbg & 32))
#endif
    __builtin_amdgcn_s_barrier();          // ... and both hold for every wave
    TR(4);
    lds_read(smem + nslot, 0, 0);
    if constexpr (LW == 0) pieces(c0_t{}, cp0_t{}, slot);     // start refilling the slot tile kt just vacated with tile kt+NST
#pragma unroll
    for (int ct = CT - HB; ct < CT; ++ct) mma_half(1, ct);
    __builtin_amdgcn_sched_group_barrier(0x100, CT + PT, 0);
    sched_stage<HB * PT, 0, LW ? 0 : P0>();
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
  if constexpr (LW == 0) wait_vmcnt<0>();  // the out-of-range tail DMAs still write (zeros) into the ring
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
  conv_tile_epilogue<BCO, BPX, WCO, WPX, CT, PT, conv_epi_ring(BCO, BPX, NST * STAGE, T), GNB>(p, acc, smem, co0, px0, totpx, [&](int i_) { TRE(i_); });
#else
  conv_tile_epilogue<BCO, BPX, WCO, WPX, CT, PT, conv_epi_ring(BCO, BPX, NST * STAGE, T), GNB>(p, acc, smem, co0, px0, totpx, [](int) {});
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

// NMF x { 1 MFMA, its share of the NRD fragment reads, 1 DMA piece for the first NVM }
template <int I, int NMF, int NRD, int NVM>
__device__ __forceinline__ void kt_sched() {
  if constexpr (I < NMF) {
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    constexpr int R = ((I + 1) * NRD) / NMF - (I * NRD) / NMF;
    if constexpr (R > 0) __builtin_amdgcn_sched_group_barrier(0x100, R, 0);
    if constexpr (I < NVM) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
    kt_sched<I + 1, NMF, NRD, NVM>();
  }
}

// ================================================================================================
// v4 ("conv_kt"): the v3 kernel with the fragment pipeline at K-TILE granularity, for the tiles whose waves own one
// 32-pixel MFMA column (128x128 on 8 waves, 128x64, 64x128, 64x64: 2 MFMAs per k-step per wave).  Measured on the layer3
// shapes (tools/conv_cost.py): v3 spends ~1500 cycles per K tile where MFMA needs 512 and DMA issue ~600, and removing
// either changes nothing - the k-step loop is bound by LDS read LATENCY: the reads of k-step kk+1 are issued only 2 MFMAs
// (64 cycles) before their use.  Here a wave reads all 4 k-steps of tile kt+1 (12 ds_read_b128, 48 VGPRs) while it runs the 8
// MFMAs of tile kt: the latency is paid once per tile and hidden under a whole tile of MFMAs.  The ring slot of tile kt+1 is
// read during tile kt, so a tile must have landed one iteration earlier than in v3: NST = 3 for the same look-ahead.
// ================================================================================================
template <int BCO, int BPX, int WCO, int WPX, int NST>
__device__ __forceinline__ void conv_kt_body(const ConvK& p, unsigned char* smem) {
  // (a __device__ function, not the kernel itself: lambdas inside a __global__ function are implicitly __host__ __device__, and
  // one that contains AMDGPU builtins silently costs the kernel its host stub)
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
  const int lrow = tid >> 3;
  const int chunk = (tid & 7) ^ ((tid >> 4) & 7);     // source chunk that belongs in LDS slot (tid & 7) of this row

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
  // ---- per-lane gather state of the XPASS pixel rows this lane fills (constant over the K loop)
  unsigned r_base[XPASS], r_step[XPASS], r_mask[XPASS];
#pragma unroll
  for (int i = 0; i < XPASS; ++i) {
    const int gp = px0 + lrow + RPP * i;
    int seg = 0, img = 0, y = 0, x = 0;
    const bool ok = gp < totpx;
    if (ok) decode_pixel(p, gp, seg, img, y, x);
    const int sh = p.sh[seg], sw = p.sw[seg];
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
    r_base[i] = (unsigned)(((int)(p.soff[seg]) + img * sh * sw + row0 * sw + col0) * p.lds + chunk * 8) * 2u + margin;
  }
  const unsigned w_voff = (unsigned)(lrow * (int)p.wrow + chunk * 8) * 2u;
  const unsigned w_pass = (unsigned)(RPP * (int)p.wrow) * 2u;
  constexpr int LPT = WPASS + XPASS;

  // ---- DMA-side state.  Everything a tile's LPT DMA instructions need is ready-made: per-lane offsets xv[] (the pixel row
  // at the current tap, or out of range), wv (weights), scalar offsets s_pix / s_w, the ring slot's LDS base.  The per-tile
  // update is three scalar adds; a new tap (every kc tiles) and the end of the K range are rare uniform branches.  (v3 redid the
  // tap / mask / wrap arithmetic branch-free for every tile: ~60 SALU + ~25 VALU per 8 MFMAs - the K loop of the small tiles
  // was bound by instruction issue, tools/pmc_conv.sh.)
  const int kc_ = p.kc, kw_ = p.kw, mode_ = p.mode, lds2_ = p.lds * 2;      // (locals: the K loop must not touch the argument struct)
  int d_c = kt0 % kc_;                                          // channel slice of the tile being fetched
  int d_tr = (kt0 / kc_) / kw_, d_ts = (kt0 / kc_) % kw_;       // ... and its tap
  int d_left = kt1 - kt0;                                       // live tiles still to fetch
  unsigned xv[XPASS], wv = w_voff;
  unsigned s_pix = 0, s_w = (unsigned)(co0 * (int)p.wrow + kt0 * BK) * 2u;
  unsigned ld_base = 0;                                         // LDS offset of the slot being filled
  // (plain statements and one-level lambdas only: a lambda that calls another by-reference lambda keeps the captured scalars in
  // memory, and every value loaded back from scratch counts as divergent - the DMA offsets then go through waterfall loops)
#define KT_TAP_SETUP()                                                                                                   \
  do {                                                                                                                   \
    const unsigned sel_ = (1u << d_tr) | (0x100u << d_ts);                                                               \
    for (int j_ = 0; j_ < XPASS; ++j_)                                                                                   \
      xv[j_] = (r_mask[j_] & sel_) == sel_ ? r_base[j_] + (unsigned)d_tr * r_step[j_] : 0x80000000u;                      \
    s_pix = (unsigned)((mode_ == 0 ? d_ts : kw_ - 1 - d_ts) * lds2_ + d_c * 128);                                        \
  } while (0)
  KT_TAP_SETUP();
  if (d_left <= 0) {
#pragma unroll
    for (int j = 0; j < XPASS; ++j) xv[j] = 0x80000000u;
    wv = 0x80000000u;
  }
  const unsigned lw = (unsigned)(wave * 8 * 128);
  auto piece = [&](int j) {            // j is a constant after unrolling
#ifdef DSL_ABLATE_BUILD
    if (p.dbg & (j < XPASS ? 1 : 2)) return;
#endif
    if (j < XPASS)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_src, (lptr_t)(smem + ld_base + lw + TILE_W + j * RPP * 128), 16, xv[j], s_pix, 0, 0);
    else
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_wgt, (lptr_t)(smem + ld_base + lw + (j - XPASS) * RPP * 128), 16, wv,
                                               s_w + (unsigned)(j - XPASS) * w_pass, 0, 0);
  };
  auto advance = [&]() {               // the tile's pieces are all issued
    s_pix += 128;
    s_w += BK * 2;
    ld_base = ld_base + STAGE == NST * STAGE ? 0u : ld_base + STAGE;
    --d_left;
    ++d_c;
    if (d_left == 0) {                 // past the K range: zeros from here on (fixed DMA count per iteration, see v3)
#pragma unroll
      for (int j = 0; j < XPASS; ++j) xv[j] = 0x80000000u;
      wv = 0x80000000u;
    } else if (d_c == kc_) {           // next tap
      d_c = 0;
      if (++d_ts == kw_) {
        d_ts = 0;
        ++d_tr;
      }
      KT_TAP_SETUP();
    }
  };

#ifdef DSL_ABLATE_BUILD
  if (p.dbg & 256) return;                   // + kernel-argument loads and the per-pixel decode
#endif
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
  // fragments of a WHOLE K tile per buffer (4 k-steps x (2 A + PT B)), two buffers: the reads of tile kt+1 are issued
  // between the MFMAs of tile kt, one full tile (>= 256 MFMA cycles per wave) before their first use
  bf16x8 fa[2][4][2], fb[2][4][PT];
  auto read_all = [&](const unsigned char* base, int f) {
#ifdef DSL_ABLATE_BUILD
    if (p.dbg & 64) return;
#endif
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int coff = ((2 * kk + fhalf) ^ fswz) << 4;
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) fa[f][kk][ct] = *reinterpret_cast<const bf16x8*>(base + a_off + ct * 32 * 128 + coff);
#pragma unroll
      for (int pt = 0; pt < PT; ++pt) fb[f][kk][pt] = *reinterpret_cast<const bf16x8*>(base + b_off + pt * 32 * 128 + coff);
    }
  };
  auto mma_all = [&](int f) {
#ifdef DSL_ABLATE_BUILD
    if (p.dbg & 4) return;
#endif
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int pt = 0; pt < PT; ++pt)
          acc[ct][pt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[f][kk][ct], fb[f][kk][pt], acc[ct][pt], 0, 0, 0);
  };
  constexpr int NMF = 8 * PT, NRD = 4 * (2 + PT);
  static_assert(NST * LPT <= 63, "vmcnt range");
  // prologue: all NST slots filled (tiles kt0 .. kt0+NST-1), tile kt0's fragments on their way to buffer 0
#pragma unroll
  for (int s = 0; s < NST; ++s) {
#pragma unroll
    for (int j = 0; j < LPT; ++j) piece(j);
    advance();
  }
  wait_vmcnt<(NST - 1) * LPT>();
  __builtin_amdgcn_s_barrier();
  read_all(smem, 0);
  unsigned rd_base = 0;                  // LDS offset of the slot whose fragments are read NEXT (tile kt+1)
  // (no generic lambdas here: a lambda with AMDGPU builtins that is called from inside a generic lambda fails substitution in
  // hipcc's HOST pass, and the kernel silently loses its host stub - an undefined symbol at load time)
  auto iter = [&](int f) {             // f is a constant after inlining
    rd_base = rd_base + STAGE == NST * STAGE ? 0u : rd_base + STAGE;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // this wave's reads of tile kt are in registers
    wait_vmcnt<(NST - 2) * LPT>();                         // tile kt+1 landed (this wave's pieces)
#ifdef DSL_ABLATE_BUILD
    if (!(p.dbg & 32))
#endif
    __builtin_amdgcn_s_barrier();                          // ... for every wave; slot(kt) is free
    read_all(smem + rd_base, f ^ 1);
#pragma unroll
    for (int j = 0; j < LPT; ++j) piece(j);                // tile kt+NST -> the slot tile kt just vacated
    mma_all(f);
    kt_sched<0, NMF, NRD, LPT>();
    __builtin_amdgcn_sched_barrier(0);
    advance();
  };
  // an even number of iterations (the buffer index is a compile-time constant): a tile past kt1 is all zeros (out-of-range DMAs)
  for (int kt = kt0; kt < kt1; kt += 2) {
    iter(0);
    iter(1);
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  wait_vmcnt<0>();                         // the out-of-range tail DMAs still write (zeros) into the ring
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

  // ---- epilogue staged through LDS (see conv_glds_kernel)
  constexpr int ROWB = BCO * 4 + 16;
  constexpr int CPX = 32 * WPX;
  constexpr int GPR = BCO / 8;
  static_assert(CPX * ROWB <= 160 * 1024, "epilogue staging must fit in LDS (the host sizes LDS as max(ring, staging))");
#pragma unroll
  for (int pt = 0; pt < PT; ++pt) {
    lds_barrier();
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

template <int BCO, int BPX, int WCO, int WPX, int NST>
__global__ __launch_bounds__(64 * WCO * WPX) void conv_kt_kernel(const ConvK p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  conv_kt_body<BCO, BPX, WCO, WPX, NST>(p, smem);
}

// ================================================================================================
// weight gradient
// ================================================================================================
struct WgK {
  int nseg, n;
  int gh[DSL_MAX_SEG], gw[DSL_MAX_SEG], sh[DSL_MAX_SEG], sw[DSL_MAX_SEG];
  int pxstart[DSL_MAX_SEG + 1];
  long long xoff[DSL_MAX_SEG];
  FastDiv dhw[DSL_MAX_SEG], dwd[DSL_MAX_SEG];
  int cs, cy, kh, kw, stride, pad;
  int ktiles, tiles_per_split, ctiles_per_tap;
  int dbg;                  // ablation knobs (ablation build only): 1 = skip DMA after the first tile, 2 = skip MFMA, 4 = no epilogue
  int gx, gy, splits;       // v2: workgroup grid (cout tiles, column tiles) and split count for the XCD-aware 1-D launch
  int chunk;                // v2: consecutive work items (split-major) per XCD
  int group;                // v2: convolutions sharing this geometry in one launch (dsl_conv2d_wgrad_group)
  int ldx;                  // pixel stride of X in elements (>= cs)
  int cyp;                  // v2: cy rounded up to the cout tile (partial-tile rows in the workspace); dY columns >= cy read as zero
  const uint16_t* dyv[DSL_MAX_GROUP];
  const uint16_t* xv[DSL_MAX_GROUP];
  long long krow;
  const uint16_t* dy;
  const uint16_t* x;
  float* ws;
  // multi-launch (dsl_conv2d_wgrad_multi): with direct = 1 (splits == 1) the finished tile goes straight into dW (x scale)
  int direct, cd;
  int totpx;                // = pxstart[nseg] (a runtime-indexed read would keep a table copy of this struct in scratch)
  float* dwv[DSL_MAX_GROUP];
  const float* scalev[DSL_MAX_GROUP];
  // bias gradients db[co] = sum over pixels of dY[.][co], summed by the tap-0 / first-cin-tile workgroups from the dY stages
  // they stream anyway: dbmask bit g = member g has a db; partial sums go to dbws[(split * group + member) * cyp + co]
  // (direct launches: straight into dbv[member]).  Fixed summation order, no atomics.
  int dbmask;
  float* dbws;
  float* dbv[DSL_MAX_GROUP];
  // v3 (wgrad_pipe): per-pixel gather descriptors of this geometry (PixDesc, one per dY pixel, built once per geometry)
  const void* pixtab;
  unsigned pixtab_bytes;
  unsigned ybytes;          // extent of one member's dY in bytes (= totpx * cy * 2): rows past it read as zeros
};

template <int ROWBYTES>
__device__ __forceinline__ int tr_swz(int row) {
  return ROWBYTES == 256 ? (row & 3) : ((row >> 1) & 1);
}

// BCO couts x 128 cins per workgroup, 64 pixels per K stage; both operands are stored
// [pixel][channel] in LDS and read with ds_read_b64_tr_b16 (hardware transpose) into MFMA fragments.
template <int BCO>
__global__ __launch_bounds__(256) void wgrad_kernel(const WgK p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int YB = BCO * 2;             // bytes per pixel row of the dY tile
  constexpr int XB = 256;                 // 128 cin * 2
  constexpr int TILE_Y = 64 * YB, TILE_X = 64 * XB, STAGE = TILE_Y + TILE_X;
  constexpr int WM = BCO / 2, CT = WM / 32;
  constexpr int YCPR = BCO / 8;           // 16-byte chunks per dY row
  constexpr int YRPP = 256 / YCPR;        // rows per pass
  constexpr int YPASS = 64 / YRPP;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wave_co = wave >> 1, wave_ci = wave & 1;
  const int co0 = blockIdx.x * BCO;
  const int colt = blockIdx.y;
  const int tap = colt / p.ctiles_per_tap;
  const int ci0 = (colt - tap * p.ctiles_per_tap) * 128;
  const int tr = tap / p.kw, ts = tap - tr * p.kw;
  const int sp = blockIdx.z;
  const int kt0 = sp * p.tiles_per_split;
  const int kt1 = min(kt0 + p.tiles_per_split, p.ktiles);
  const int totpx = p.pxstart[p.nseg];

  const int yrow = tid / YCPR, ychunk = tid % YCPR;
  const int xrow = tid >> 4, xchunk = tid & 15;
  u32x4 ry[YPASS], rx[4];

  auto gload = [&](int kt) {
#pragma unroll
    for (int i = 0; i < YPASS; ++i) {
      const int gp = kt * 64 + yrow + YRPP * i;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (gp < totpx) v = *reinterpret_cast<const u32x4*>(p.dy + (long long)gp * p.cy + co0 + ychunk * 8);
      ry[i] = v;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int gp = kt * 64 + xrow + 16 * i;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (gp < totpx) {
        int seg = 0;
#pragma unroll
        for (int s = 1; s < DSL_MAX_SEG; ++s)
          if (s < p.nseg && gp >= p.pxstart[s]) seg = s;
        const uint32_t q = gp - p.pxstart[seg];
        const uint32_t img = fdiv(q, p.dhw[seg]);
        const uint32_t rem = q - img * p.dhw[seg].d;
        const uint32_t y = fdiv(rem, p.dwd[seg]);
        const uint32_t x = rem - y * p.dwd[seg].d;
        const int sy = (int)y * p.stride + tr - p.pad, sx = (int)x * p.stride + ts - p.pad;
        if ((unsigned)sy < (unsigned)p.sh[seg] && (unsigned)sx < (unsigned)p.sw[seg]) {
          const long long pix = p.xoff[seg] + ((long long)img * p.sh[seg] + sy) * p.sw[seg] + sx;
          v = *reinterpret_cast<const u32x4*>(p.x + pix * p.ldx + ci0 + xchunk * 8);
        }
      }
      rx[i] = v;
    }
  };
  auto lds_store = [&](int buf) {
    unsigned char* base = smem + buf * STAGE;
#pragma unroll
    for (int i = 0; i < YPASS; ++i) {
      const int row = yrow + YRPP * i;
      *reinterpret_cast<u32x4*>(base + row * YB + ((((ychunk >> 2) ^ tr_swz<YB>(row))) << 6) + ((ychunk & 3) << 4)) = ry[i];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = xrow + 16 * i;
      *reinterpret_cast<u32x4*>(base + TILE_Y + row * XB + ((((xchunk >> 2) ^ tr_swz<XB>(row))) << 6) + ((xchunk & 3) << 4)) = rx[i];
    }
  };

  f32x16 acc[CT][2];
#pragma unroll
  for (int a = 0; a < CT; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[a][b][j] = 0.f;

  // transpose-read geometry: 16-lane group g covers channel block (g&1)*16 and pixel block (g>>1)*8
  const int g16 = lane >> 4, l16 = lane & 15;
  const int iblk = (g16 & 1) * 16, kblk = (g16 >> 1) * 8;
  const int krow_l = kblk + (l16 >> 2);           // + kk*16 + hh*4
  const int ccol_l = iblk + 4 * (l16 & 3);        // channel (element) offset inside a 32-wide tile

  auto tr_read = [&](const unsigned char* tile, int rowbytes_sel, int krow, int col) -> s16x4 {
    int byte;
    if (rowbytes_sel == 256)
      byte = krow * 256 + ((((col * 2) >> 6) ^ tr_swz<256>(krow)) << 6) + ((col * 2) & 63);
    else
      byte = krow * 128 + ((((col * 2) >> 6) ^ tr_swz<128>(krow)) << 6) + ((col * 2) & 63);
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) s16x4*)(tile + byte));
  };

  auto compute = [&](int buf) {
    const unsigned char* base = smem + buf * STAGE;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      bf16x8 a[CT], b[2];
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        const int col = wave_co * WM + ct * 32 + ccol_l;
        const s16x4 lo = tr_read(base, YB, kk * 16 + krow_l, col);
        const s16x4 hi = tr_read(base, YB, kk * 16 + krow_l + 4, col);
        union { struct { s16x4 l, h; } s; bf16x8 v; } u;
        u.s.l = lo;
        u.s.h = hi;
        a[ct] = u.v;
      }
#pragma unroll
      for (int pt = 0; pt < 2; ++pt) {
        const int col = wave_ci * 64 + pt * 32 + ccol_l;
        const s16x4 lo = tr_read(base + TILE_Y, XB, kk * 16 + krow_l, col);
        const s16x4 hi = tr_read(base + TILE_Y, XB, kk * 16 + krow_l + 4, col);
        union { struct { s16x4 l, h; } s; bf16x8 v; } u;
        u.s.l = lo;
        u.s.h = hi;
        b[pt] = u.v;
      }
#pragma unroll
      for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int pt = 0; pt < 2; ++pt)
          acc[ct][pt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ct], b[pt], acc[ct][pt], 0, 0, 0);
    }
  };

  if (kt0 < kt1) {
    gload(kt0);
    lds_store(0);
    __syncthreads();
    for (int kt = kt0; kt < kt1; ++kt) {
      const int cur = (kt - kt0) & 1;
      if (kt + 1 < kt1) gload(kt + 1);
      compute(cur);
      if (kt + 1 < kt1) lds_store(cur ^ 1);
      __syncthreads();
    }
  }

  // partial tile -> workspace [split][cy][krow]
  const int frow = lane & 31, fhalf = lane >> 5;
#pragma unroll
  for (int ct = 0; ct < CT; ++ct)
#pragma unroll
    for (int pt = 0; pt < 2; ++pt) {
      const long long col = (long long)tap * p.cs + ci0 + wave_ci * 64 + pt * 32 + frow;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int co = co0 + wave_co * WM + ct * 32 + (j & 3) + 8 * (j >> 2) + 4 * fhalf;
        p.ws[((long long)sp * p.cy + co) * p.krow + col] = acc[ct][pt][j];
      }
    }
}

// inline-asm helpers must be explicit __device__ functions: a lambda inside a kernel is implicitly
// __host__ __device__, and its AMDGPU asm constraints break the (silently dropped) host instantiation
__device__ __forceinline__ u32x2 lds_tr_read_b64(unsigned addr) {
  u32x2 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(addr) : "memory");
  return v;
}
// ... with the constant part of the address in the instruction's 16-bit offset field (inline asm is opaque to the compiler: given
// the whole address in a register it spends one v_add per read - 24 of the 33 VALU instructions per stage of wgrad_pipe's K loop)
template <int OFF>
__device__ __forceinline__ u32x2 lds_tr_read_b64_o(unsigned addr) {
  static_assert(OFF >= 0 && OFF < 65536, "DS offset field");
  u32x2 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
  return v;
}
// read (k-step kk, half hi) of a fragment column whose (stage, column) address is `base`: kk and hi are constants after unrolling
template <int ROWB, int KK>
__device__ __forceinline__ u32x2 lds_tr_read_kh(unsigned base, int kk, int hi) {
  static_assert(KK <= 4, "k-steps per stage");
  switch (kk * 2 + hi) {
    case 0: return lds_tr_read_b64_o<0>(base);
    case 1: return lds_tr_read_b64_o<4 * ROWB>(base);
    case 2: return lds_tr_read_b64_o<16 * ROWB>(base);
    case 3: return lds_tr_read_b64_o<20 * ROWB>(base);
    case 4: return lds_tr_read_b64_o<32 * ROWB>(base);
    case 5: return lds_tr_read_b64_o<36 * ROWB>(base);
    case 6: return lds_tr_read_b64_o<48 * ROWB>(base);
    default: return lds_tr_read_b64_o<52 * ROWB>(base);
  }
}
struct Frag {
  u32x2 lo, hi;
};
__device__ __forceinline__ unsigned lds_read_b32_asm(unsigned addr) {
  unsigned v;
  asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(addr) : "memory");
  return v;
}
__device__ __forceinline__ void wait_lds4(unsigned& a, unsigned& b, unsigned& c, unsigned& d) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : : "memory");
}
// Wait for the outstanding transpose reads AND tell the compiler the fragment registers change here
// ("+v"): otherwise it may copy an asm-loaded register before the data has landed (the destination of an
// inline-asm load counts as written when the statement ends, not when the LDS returns).
template <int CT, int IT>
__device__ __forceinline__ void wait_frags(Frag (&fa)[CT], Frag (&fb)[IT]) {
  static_assert(IT == 2 && (CT == 2 || CT == 4), "fragment shapes used by the wgrad tiles");
  if constexpr (CT == 4) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(fa[0].lo), "+v"(fa[0].hi), "+v"(fa[1].lo), "+v"(fa[1].hi), "+v"(fa[2].lo), "+v"(fa[2].hi),
                   "+v"(fa[3].lo), "+v"(fa[3].hi), "+v"(fb[0].lo), "+v"(fb[0].hi), "+v"(fb[1].lo), "+v"(fb[1].hi)
                 :
                 : "memory");
  } else {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(fa[0].lo), "+v"(fa[0].hi), "+v"(fa[1].lo), "+v"(fa[1].hi), "+v"(fb[0].lo), "+v"(fb[0].hi),
                   "+v"(fb[1].lo), "+v"(fb[1].hi)
                 :
                 : "memory");
  }
  __builtin_amdgcn_sched_barrier(0);
}

// v2 weight gradient: DMA-to-LDS operands, 256-wide tiles, 8 waves, NST-deep ring of KS-pixel stages with
// counted vmcnt waits (the pixel streams come from HBM: one stage of lookahead does not cover the latency).
// Same math/outputs as wgrad_kernel.
struct SegSel {          // per-segment decode constants, selected with v_cndmask chains (no memory access:
  int px0;               // indexing kernel-argument arrays or LDS tables by a runtime segment id makes hipcc
  uint32_t m1lo, m1hi, d1, m2lo, m2hi, d2;   // drain the DMA queue with s_waitcnt vmcnt(0) inside the K loop)
  int sh, sw;
  long long xoff;
};
__device__ __forceinline__ SegSel seg_select(const WgK& p, int gp) {
  SegSel r;
  r.px0 = p.pxstart[0];
  r.m1lo = (uint32_t)p.dhw[0].m; r.m1hi = (uint32_t)(p.dhw[0].m >> 32); r.d1 = p.dhw[0].d;
  r.m2lo = (uint32_t)p.dwd[0].m; r.m2hi = (uint32_t)(p.dwd[0].m >> 32); r.d2 = p.dwd[0].d;
  r.sh = p.sh[0]; r.sw = p.sw[0]; r.xoff = p.xoff[0];
#pragma unroll
  for (int s = 1; s < DSL_MAX_SEG; ++s) {
    const bool in = s < p.nseg && gp >= p.pxstart[s];
    r.px0 = in ? p.pxstart[s] : r.px0;
    r.m1lo = in ? (uint32_t)p.dhw[s].m : r.m1lo; r.m1hi = in ? (uint32_t)(p.dhw[s].m >> 32) : r.m1hi;
    r.d1 = in ? p.dhw[s].d : r.d1;
    r.m2lo = in ? (uint32_t)p.dwd[s].m : r.m2lo; r.m2hi = in ? (uint32_t)(p.dwd[s].m >> 32) : r.m2hi;
    r.d2 = in ? p.dwd[s].d : r.d2;
    r.sh = in ? p.sh[s] : r.sh; r.sw = in ? p.sw[s] : r.sw;
    r.xoff = in ? p.xoff[s] : r.xoff;
  }
  return r;
}

template <int BCO, int BCI, int WCO, int WCI, int KS, int NST>
__device__ __forceinline__ void wgrad_glds_body(const WgK& p, const int bid, unsigned char* smem) {
  constexpr int NW = WCO * WCI;
  constexpr int YB = BCO * 2, XB = BCI * 2;          // bytes per pixel row of each tile
  constexpr int TILE_Y = KS * YB, TILE_X = KS * XB, STAGE = TILE_Y + TILE_X;
  constexpr int NY = TILE_Y / 1024, NX = TILE_X / 1024;      // 1 KB DMA instructions per tile
  constexpr int LY = NY / NW, LX = NX / NW;                   // per wave
  constexpr int LPT = LY + LX;
  constexpr int CT = BCO / WCO / 32, IT = BCI / WCI / 32;
  constexpr int KK = KS / 16;                                  // MFMA k-steps per stage
  static_assert(NY % NW == 0 && NX % NW == 0 && LY >= 1 && LX >= 1, "tile / wave mismatch");
  static_assert(YB >= 256 && XB >= 256, "64-byte-chunk swizzle needs >= 4 chunks per row");
  static_assert(KK == 2 || KK == 4, "stage depth");
  static_assert((NST - 2) * LPT <= 63 && NST >= 2 && NST <= 4, "vmcnt range");

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wave_co = wave / WCI, wave_ci = wave % WCI;
  // XCD-aware work mapping (blocks are dealt round-robin to the XCDs: equal b % 8 = same XCD): the (cout tile, tap, cin tile) workgroups of one pixel
  // split are neighbours on one XCD, so its dY / X pixel range is fetched into that XCD's L2 once instead of
  // once per tap.  Placement only affects speed.
  const int tiles_per_member = p.gx * p.gy;
  const int tiles_per_split_wg = tiles_per_member * p.group;
  const int xcd = bid & 7, jj = bid >> 3;
  const int witem = xcd * p.chunk + jj;              // work items are split-major: an XCD owns a contiguous range
  if (jj >= p.chunk || witem >= tiles_per_split_wg * p.splits) return;
  const int sp = witem / tiles_per_split_wg;
  const int rem_sp = witem - sp * tiles_per_split_wg;
  const int member = rem_sp / tiles_per_member;      // which convolution of the group
  const int rem_wg = rem_sp - member * tiles_per_member;
  // member pointers by select chain, once, outside the K loop (a runtime-indexed kernel-argument load inside the
  // loop would make hipcc drain the DMA queue)
  const uint16_t* dy_p = p.dyv[0];
  const uint16_t* x_p = p.xv[0];
  float* db_p = p.dbv[0];
#pragma unroll
  for (int g = 1; g < DSL_MAX_GROUP; ++g) {
    dy_p = member == g ? p.dyv[g] : dy_p;
    x_p = member == g ? p.xv[g] : x_p;
    db_p = member == g ? p.dbv[g] : db_p;
  }
  const int co0 = (rem_wg % p.gx) * BCO;
  const int colt = rem_wg / p.gx;
  // column sums of dY (the bias gradient) ride along in the workgroups of column tile 0
  const bool do_db = colt == 0 && ((p.dbmask >> member) & 1);
  constexpr int DB_PAIRS = BCO / 2, DB_RG = 64 * NW / DB_PAIRS, DB_ROWS = KS / DB_RG;
  static_assert(DB_ROWS % 4 == 0 && DB_RG * DB_PAIRS == 64 * NW, "bias-gradient thread mapping");
  const int db_cp = tid % DB_PAIRS, db_rg = tid / DB_PAIRS;
  float db_lo = 0.f, db_hi = 0.f;
  const int ctiles = p.cs / BCI;
  const int tap = colt / ctiles;
  const int ci0 = (colt - tap * ctiles) * BCI;
  const int tr = tap / p.kw, ts = tap - tr * p.kw;
  const int kt0 = sp * p.tiles_per_split;
  const int kt1 = min(kt0 + p.tiles_per_split, p.ktiles);
  const int totpx = p.totpx;
  const gptr_t zero = (gptr_t)g_zero_line;

  // per DMA instruction this lane's (row, source channel) inside the tile
  int yrow[LY], ych[LY], xrow[LX], xch[LX];
#pragma unroll
  for (int i = 0; i < LY; ++i) {
    const int off = (wave + NW * i) * 1024 + lane * 16;
    const int row = off / YB, inrow = off % YB;
    yrow[i] = row;
    ych[i] = (((inrow >> 6) ^ (row & 3)) << 5) + ((inrow & 63) >> 1);      // bf16 element offset in the row
  }
#pragma unroll
  for (int i = 0; i < LX; ++i) {
    const int off = (wave + NW * i) * 1024 + lane * 16;
    const int row = off / XB, inrow = off % XB;
    xrow[i] = row;
    xch[i] = (((inrow >> 6) ^ (row & 3)) << 5) + ((inrow & 63) >> 1);
  }

  f32x16 acc[CT][IT];
#pragma unroll
  for (int a = 0; a < CT; ++a)
#pragma unroll
    for (int b = 0; b < IT; ++b)
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[a][b][j] = 0.f;

  // Transpose reads are issued through inline asm: given the builtin (an addrspace(3) access) hipcc orders
  // every ds_read behind the in-flight LDS-DMA with s_waitcnt vmcnt(0), which would serialise DMA and MFMA.
  // The hazards are handled by hand: DMA data is read one barrier after its counted vmcnt wait; fragment
  // registers are consumed only after an explicit lgkmcnt(0) naming them.
  const int g16 = lane >> 4, l16 = lane & 15;
  const int iblk = (g16 & 1) * 16, kblk = (g16 >> 1) * 8;
  const int krow_l = kblk + (l16 >> 2);
  const int ccol_l = iblk + 4 * (l16 & 3);
  const unsigned lds_base = (unsigned)(size_t)smem;      // low 32 bits of a flat LDS address = the LDS offset
  unsigned a_off[CT], b_off[IT];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) {
    const int col = wave_co * (32 * CT) + ct * 32 + ccol_l;
    a_off[ct] = krow_l * YB + ((((col * 2) >> 6) ^ (krow_l & 3)) << 6) + ((col * 2) & 63);
  }
#pragma unroll
  for (int it = 0; it < IT; ++it) {
    const int col = wave_ci * (32 * IT) + it * 32 + ccol_l;
    b_off[it] = TILE_Y + krow_l * XB + ((((col * 2) >> 6) ^ (krow_l & 3)) << 6) + ((col * 2) & 63);
  }
  auto issue = [&](unsigned stage_addr, int kk, Frag (&fa)[CT], Frag (&fb)[IT]) {
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      const unsigned ad = stage_addr + a_off[ct] + kk * 16 * YB;
      fa[ct].lo = lds_tr_read_b64(ad);
      fa[ct].hi = lds_tr_read_b64_o<4 * YB>(ad);
    }
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      const unsigned ad = stage_addr + b_off[it] + kk * 16 * XB;
      fb[it].lo = lds_tr_read_b64(ad);
      fb[it].hi = lds_tr_read_b64_o<4 * XB>(ad);
    }
  };
  auto mma = [&](Frag (&fa)[CT], Frag (&fb)[IT]) {
    bf16x8 a[CT], b[IT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      union { struct { u32x2 l, h; } s; bf16x8 v; } u;
      u.s.l = fa[ct].lo;
      u.s.h = fa[ct].hi;
      a[ct] = u.v;
    }
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      union { struct { u32x2 l, h; } s; bf16x8 v; } u;
      u.s.l = fb[it].lo;
      u.s.h = fb[it].hi;
      b[it] = u.v;
    }
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int it = 0; it < IT; ++it)
        acc[ct][it] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ct], b[it], acc[ct][it], 0, 0, 0);
  };
  auto compute = [&](int slot) {
    const unsigned st = lds_base + slot * STAGE;
    Frag fa0[CT], fb0[IT], fa1[CT], fb1[IT];
    issue(st, 0, fa0, fb0);
    wait_frags<CT, IT>(fa0, fb0);
    issue(st, 1, fa1, fb1);           // next fragments fly while the MFMAs of this step run
    mma(fa0, fb0);
    wait_frags<CT, IT>(fa1, fb1);
    if constexpr (KK == 4) {
      issue(st, 2, fa0, fb0);
      mma(fa1, fb1);
      wait_frags<CT, IT>(fa0, fb0);
      issue(st, 3, fa1, fb1);
      mma(fa0, fb0);
      wait_frags<CT, IT>(fa1, fb1);
    }
    mma(fa1, fb1);
  };

  // DMA of one stage (tile index kt -> ring slot): address math first, then the burst of LPT instructions.
  // Written once and inlined at its single call site inside the loop.
  int slot_c = 0;                // ring slot of the tile being computed
  int slot_i = 0;                // ring slot the next DMA goes to
  for (int kc = kt0 - (NST - 1); kc < kt1; ++kc) {
    const int kl = kc + NST - 1;             // tile whose DMA is issued in this iteration
    if (kc >= kt0) {
      // tile kc must have landed: tiles kc+1 .. min(kc+NST-2, kt1-1) may still be in flight
      const int ahead = min(kt1 - 1 - kc, NST - 2);
      if (NST >= 4 && ahead >= 2) wait_vmcnt<(NST >= 4 ? 2 : 0) * LPT>();
      else if (NST >= 3 && ahead >= 1) wait_vmcnt<(NST >= 3 ? 1 : 0) * LPT>();
      else wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();          // everyone's part of tile kc landed; compute(kc-1) finished everywhere
    }
#ifdef DSL_ABLATE_BUILD
    if (kl < kt1 && !((p.dbg & 1) && kc >= kt0)) {
#else
    if (kl < kt1) {
#endif
      unsigned char* stage = smem + slot_i * STAGE;
      // every lane decodes ONE pixel row of the stage (row = lane) and the DMA instructions pick their rows'
      // source offsets up with a lane shuffle: one decode per stage instead of one per DMA instruction
      int my_off = -1;             // element offset of this row's source pixel (channel ci0), -1 = zero line
      {
        const int gp = kl * KS + lane;
        if (lane < KS && gp < totpx) {
          const SegSel t = seg_select(p, gp);
          const uint32_t q = gp - t.px0;
          const uint64_t m1 = ((uint64_t)t.m1hi << 32) | t.m1lo, m2 = ((uint64_t)t.m2hi << 32) | t.m2lo;
          const uint32_t img = (uint32_t)(((uint64_t)q * m1) >> 40);
          const uint32_t rem = q - img * t.d1;
          const uint32_t y = (uint32_t)(((uint64_t)rem * m2) >> 40);
          const uint32_t x = rem - y * t.d2;
          const int sy = (int)y * p.stride + tr - p.pad, sx = (int)x * p.stride + ts - p.pad;
          if ((unsigned)sy < (unsigned)t.sh && (unsigned)sx < (unsigned)t.sw)
            my_off = (int)((t.xoff + ((long long)img * t.sh + sy) * t.sw + sx) * p.ldx + ci0);
        }
      }
      int gx[LX];
#pragma unroll
      for (int i = 0; i < LX; ++i) gx[i] = __shfl(my_off, xrow[i], 64);
#pragma unroll
      for (int i = 0; i < LY; ++i) {
        const int gp = kl * KS + yrow[i];
        const gptr_t g = (gp < totpx && co0 + ych[i] < p.cy) ? (gptr_t)(dy_p + (long long)gp * p.cy + co0 + ych[i]) : zero;
        __builtin_amdgcn_global_load_lds(g, (lptr_t)(stage + (wave + NW * i) * 1024), 16, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < LX; ++i) {
        // keep the select in a named variable: passing the ?: expression straight into the builtin makes
        // hipcc silently drop this kernel's host stub
        const gptr_t g = gx[i] >= 0 ? (gptr_t)(x_p + gx[i] + xch[i]) : zero;
        __builtin_amdgcn_global_load_lds(g, (lptr_t)(stage + TILE_Y + (wave + NW * i) * 1024), 16, 0, 0);
      }
    }
    if (kl >= kt0) slot_i = (slot_i + 1 == NST) ? 0 : slot_i + 1;
    if (kc >= kt0) {
#ifdef DSL_ABLATE_BUILD
      if (!(p.dbg & 2))
#endif
      compute(slot_c);
      if (do_db) {              // this stage's dY tile: rows db_rg * DB_ROWS .. of column pair db_cp (same swizzle as the DMA wrote)
        const unsigned st = lds_base + slot_c * STAGE;
#pragma unroll
        for (int r4 = 0; r4 < DB_ROWS; r4 += 4) {
          unsigned v[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int row = db_rg * DB_ROWS + r4 + j;
            v[j] = lds_read_b32_asm(st + row * YB + ((((db_cp * 4) >> 6) ^ (row & 3)) << 6) + ((db_cp * 4) & 63));
          }
          wait_lds4(v[0], v[1], v[2], v[3]);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            db_lo += __uint_as_float(v[j] << 16);
            db_hi += __uint_as_float(v[j] & 0xffff0000u);
          }
        }
      }
      slot_c = (slot_c + 1 == NST) ? 0 : slot_c + 1;
    }
  }
#ifdef DSL_ABLATE_BUILD
  if (p.dbg & 4) return;
#endif

  if (do_db) {               // fold the row groups in a fixed order; one value per column leaves the workgroup
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);
    red[(db_rg * DB_PAIRS + db_cp) * 2] = db_lo;
    red[(db_rg * DB_PAIRS + db_cp) * 2 + 1] = db_hi;
    __syncthreads();
    if (tid < BCO) {
      float sacc = 0.f;
#pragma unroll
      for (int r = 0; r < DB_RG; ++r) sacc += red[(r * DB_PAIRS + (tid >> 1)) * 2 + (tid & 1)];
      const int co = co0 + tid;
      if (p.direct) {
        if (co < p.cd) db_p[co] = sacc;
      } else {
        p.dbws[((long long)sp * p.group + member) * p.cyp + co] = sacc;
      }
    }
  }
  const int frow = lane & 31, fhalf = lane >> 5;
  if (p.direct) {          // one split: this tile is the whole sum - scale and store it into dW, no partial / reduce pass
    float* dw_p = p.dwv[0];
    const float* sc_p = p.scalev[0];
#pragma unroll
    for (int g = 1; g < DSL_MAX_GROUP; ++g) {
      dw_p = member == g ? p.dwv[g] : dw_p;
      sc_p = member == g ? p.scalev[g] : sc_p;
    }
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int it = 0; it < IT; ++it) {
        const long long col = (long long)tap * p.cs + ci0 + wave_ci * (32 * IT) + it * 32 + frow;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          const int co = co0 + wave_co * (32 * CT) + ct * 32 + (j & 3) + 8 * (j >> 2) + 4 * fhalf;
          if (co < p.cd) dw_p[(long long)co * p.krow + col] = sc_p ? acc[ct][it][j] * sc_p[co] : acc[ct][it][j];
        }
      }
    return;
  }
#pragma unroll
  for (int ct = 0; ct < CT; ++ct)
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      const long long col = (long long)tap * p.cs + ci0 + wave_ci * (32 * IT) + it * 32 + frow;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int co = co0 + wave_co * (32 * CT) + ct * 32 + (j & 3) + 8 * (j >> 2) + 4 * fhalf;
        p.ws[(((long long)sp * p.group + member) * p.cyp + co) * p.krow + col] = acc[ct][it][j];
      }
    }
}

template <int BCO, int BCI, int WCO, int WCI, int KS, int NST>
__global__ __launch_bounds__(64 * WCO * WCI) void wgrad_glds_kernel(const WgK p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  wgrad_glds_body<BCO, BCI, WCO, WCI, KS, NST>(p, (int)blockIdx.x, smem);
}

// ================================================================================================
// v3 weight gradient ("wgrad_pipe"): the same tiles, operands and outputs as wgrad_glds_body, re-scheduled so that nothing
// but the MFMA stream is on the critical path (v2 measured on the head shape, tools/ablate_wgrad.py: DMA-only 50 us +
// MFMA-only 53 us = 87 us together - the two did not overlap at all: every wave issued its whole DMA burst, then decoded
// the next stage's pixels, then waited for its first fragments with an idle matrix pipe):
//   * no per-stage pixel decode: a per-geometry table of 8-byte pixel descriptors {source pixel of tap (0,0), row pitch,
//     per-axis tap validity bits} (PixDesc, built once per geometry on the host, cached by the library) is itself DMA'd
//     into a small LDS ring a few stages ahead; a gather row's offset is then 5 VALU instructions, a dY row's offset is a
//     running counter (dY rows are contiguous in the pixel index); padding, ragged tails and dead stages are the buffer
//     out-of-range rule (zeros land in LDS), so every stage issues the same number of DMA instructions;
//   * KS-pixel stages in an NST-deep ring (32-pixel stages: 4 x 32 KB for the 256x256 tile), stage s+NST-1 is fetched while
//     stage s feeds the MFMAs: its DMA instructions are issued ONE AT A TIME between the MFMAs of the stage (a burst blocks
//     the wave on issue for ~1000 cycles with an empty matrix pipe);
//   * fragment reads of k-step kk+1 are in flight during the MFMAs of step kk, the next stage's first fragments are issued
//     right behind the stage's single barrier, in front of its last MFMA block: the barrier sits inside the MFMA stream.
// vmcnt bookkeeping (P DMA instructions per wave per stage, returned in order): at the barrier that ends stage s, stage
// s+1 must have landed; it was issued during stage s-NST+2, so (NST-3) whole stages plus the Pa pieces of stage s issued so
// far may stay in flight.  Everything issued during stage s-NST+2 or earlier has then landed, including the descriptors
// fetched then: descriptors of stage t are fetched during stage t-(2*NST-2) and read (into registers) right behind the
// barrier that ends stage t-NST, for the gather DMAs issued during stage t-NST+1.
// ================================================================================================
struct PixDesc {
  int32_t base;        // source pixel index of tap (0,0) of this output pixel (may be "virtual": outside the image)
  uint32_t info;       // (source row pitch in pixels) << 16 | x-tap validity bits << 8 | y-tap validity bits
};

__device__ __forceinline__ u32x2 lds_read_b64_asm(unsigned addr) {
  u32x2 v;
  asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(addr) : "memory");
  return v;
}
__device__ __forceinline__ void pin2(u32x2& a) { asm volatile("" : "+v"(a)); }
__device__ __forceinline__ void pin1(unsigned& a) { asm volatile("" : "+v"(a)); }

template <int BCO, int BCI, int WCO, int WCI, int KS, int NST>
__device__ __forceinline__ void wgrad_pipe_body(const WgK& p, const int bid, unsigned char* smem) {
  constexpr int NW = WCO * WCI;
  constexpr int YB = BCO * 2, XB = BCI * 2;
  constexpr int TILE_Y = KS * YB, TILE_X = KS * XB, STAGE = TILE_Y + TILE_X;
  constexpr int NY = TILE_Y / 1024, NX = TILE_X / 1024;
  constexpr int LY = NY / NW, LX = NX / NW;
  constexpr int NDSC = KS * 8 / 256;                            // 256-byte descriptor DMAs per stage
  constexpr int P = NDSC + LY + LX;                              // DMA instructions per wave per stage
  constexpr int CT = BCO / WCO / 32, IT = BCI / WCI / 32, NM = CT * IT;
  constexpr int KK = KS / 16;
  constexpr int DR = 16;                                         // descriptor ring depth (stages)
  constexpr int DESC_BASE = NST * STAGE;
  constexpr int DLEAD = 2 * NST - 2;                             // descriptors run this many stages ahead of the stage computed
  static_assert(NY % NW == 0 && NX % NW == 0 && LY >= 1 && LX >= 1, "tile / wave mismatch");
  static_assert(YB >= 256 && XB >= 256, "64-byte-chunk swizzle needs >= 4 chunks per row");
  static_assert(KK == 2 || KK == 4, "stage depth");
  static_assert(NST >= 3 && DLEAD < DR && NDSC >= 1, "ring depths");
  static_assert((NST - 1) * P + DLEAD * NDSC <= 63, "vmcnt range");

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wave_co = wave / WCI, wave_ci = wave % WCI;
  const int tiles_per_member = p.gx * p.gy;
  const int tiles_per_split_wg = tiles_per_member * p.group;
  const int xcd = bid & 7, jj = bid >> 3;
  const int witem = xcd * p.chunk + jj;
  if (jj >= p.chunk || witem >= tiles_per_split_wg * p.splits) return;
  const int sp = witem / tiles_per_split_wg;
  const int rem_sp = witem - sp * tiles_per_split_wg;
  const int member = rem_sp / tiles_per_member;
  const int rem_wg = rem_sp - member * tiles_per_member;
  const uint16_t* dy_p = p.dyv[0];
  const uint16_t* x_p = p.xv[0];
  float* db_p = p.dbv[0];
#pragma unroll
  for (int g = 1; g < DSL_MAX_GROUP; ++g) {
    dy_p = member == g ? p.dyv[g] : dy_p;
    x_p = member == g ? p.xv[g] : x_p;
    db_p = member == g ? p.dbv[g] : db_p;
  }
  const int co0 = (rem_wg % p.gx) * BCO;
  const int colt = rem_wg / p.gx;
  const bool do_db = colt == 0 && ((p.dbmask >> member) & 1);
  constexpr int DB_PAIRS = BCO / 2, DB_RG = 64 * NW / DB_PAIRS, DB_ROWS = KS / DB_RG;
  static_assert(DB_ROWS >= 1 && DB_ROWS <= 8 && DB_RG * DB_PAIRS == 64 * NW, "bias-gradient thread mapping");
  const int db_cp = tid % DB_PAIRS, db_rg = tid / DB_PAIRS;
  float db_lo = 0.f, db_hi = 0.f;
  const int ctiles = p.cs / BCI;
  const int tap = colt / ctiles;
  const int ci0 = (colt - tap * ctiles) * BCI;
  const int tr = tap / p.kw, ts = tap - tr * p.kw;
  const int kt0 = sp * p.tiles_per_split;
  const int kt1 = min(kt0 + p.tiles_per_split, p.ktiles);
  if (kt0 >= kt1) return;      // (cannot happen with the host's split factors; a workgroup without stages has nothing to add)

  const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc((void*)dy_p, 0, (int)p.ybytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)x_p, 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_t = __builtin_amdgcn_make_buffer_rsrc((void*)p.pixtab, 0, (int)p.pixtab_bytes, 0x00020000);

  // per DMA instruction this lane's (row, source channel) inside the tile (as in wgrad_glds_body)
  unsigned yv[LY];             // running byte offset of this lane's dY element (row of the stage being fetched)
  unsigned xcol[LX], xdaddr[LX];
#pragma unroll
  for (int i = 0; i < LY; ++i) {
    const int off = (wave + NW * i) * 1024 + lane * 16;
    const int row = off / YB, inrow = off % YB;
    const int ch = (((inrow >> 6) ^ (row & 3)) << 5) + ((inrow & 63) >> 1);
    yv[i] = co0 + ch < p.cy ? (unsigned)(((kt0 * KS + row) * p.cy + co0 + ch) * 2) : 0x80000000u;
  }
#pragma unroll
  for (int i = 0; i < LX; ++i) {
    const int off = (wave + NW * i) * 1024 + lane * 16;
    const int row = off / XB, inrow = off % XB;
    const int ch = (((inrow >> 6) ^ (row & 3)) << 5) + ((inrow & 63) >> 1);
    xcol[i] = (unsigned)(ci0 + ch) * 2u;
    xdaddr[i] = (unsigned)(DESC_BASE + row * 8);
  }
  const unsigned y_step = (unsigned)(KS * p.cy * 2);
  const unsigned ldx2 = (unsigned)(p.ldx * 2);
  const unsigned sel = (1u << tr) | (0x100u << ts);

  f32x16 acc[CT][IT];
#pragma unroll
  for (int a = 0; a < CT; ++a)
#pragma unroll
    for (int b = 0; b < IT; ++b)
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[a][b][j] = 0.f;

  const int g16 = lane >> 4, l16 = lane & 15;
  const int iblk = (g16 & 1) * 16, kblk = (g16 >> 1) * 8;
  const int krow_l = kblk + (l16 >> 2);
  const int ccol_l = iblk + 4 * (l16 & 3);
  const unsigned lds_base = (unsigned)(size_t)smem;
  unsigned a_off[CT], b_off[IT];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) {
    const int col = wave_co * (32 * CT) + ct * 32 + ccol_l;
    a_off[ct] = krow_l * YB + ((((col * 2) >> 6) ^ (krow_l & 3)) << 6) + ((col * 2) & 63);
  }
#pragma unroll
  for (int it = 0; it < IT; ++it) {
    const int col = wave_ci * (32 * IT) + it * 32 + ccol_l;
    b_off[it] = TILE_Y + krow_l * XB + ((((col * 2) >> 6) ^ (krow_l & 3)) << 6) + ((col * 2) & 63);
  }
  Frag fa[2][CT], fb[2][IT];
  constexpr int NR = 2 * (CT + IT);         // fragment reads per k-step
  // fragment column addresses of the stage being READ (one v_add per column and stage; the k-step / half offsets are immediates)
  unsigned ra[CT], rb[IT];
  auto set_read_stage = [&](unsigned stage_addr) {
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) ra[ct] = stage_addr + a_off[ct];
#pragma unroll
    for (int it = 0; it < IT; ++it) rb[it] = stage_addr + b_off[it];
  };
  auto rd = [&](int kk, int f, int j) {      // read j of the k-step's NR (kk, j are constants after unrolling)
#ifdef DSL_ABLATE_BUILD
    if (p.dbg & 8) return;
#endif
    if (j < 2 * CT) {
      const int ct = j >> 1;
      if (j & 1) fa[f][ct].hi = lds_tr_read_kh<YB, KK>(ra[ct], kk, 1); else fa[f][ct].lo = lds_tr_read_kh<YB, KK>(ra[ct], kk, 0);
    } else {
      const int it = (j - 2 * CT) >> 1;
      if (j & 1) fb[f][it].hi = lds_tr_read_kh<XB, KK>(rb[it], kk, 1); else fb[f][it].lo = lds_tr_read_kh<XB, KK>(rb[it], kk, 0);
    }
  };
  auto issue = [&](int kk, int f) {
#pragma unroll
    for (int j = 0; j < NR; ++j) rd(kk, f, j);
  };
  auto wait_lds = [&](int f) {       // every outstanding LDS read of this wave has landed; the registers it wrote change HERE
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) { pin2(fa[f][ct].lo); pin2(fa[f][ct].hi); }
#pragma unroll
    for (int it = 0; it < IT; ++it) { pin2(fb[f][it].lo); pin2(fb[f][it].hi); }
  };

  // ---- DMA pieces of one stage: [0, NDSC) descriptors of stage t_desc, [NDSC, NDSC+LY) dY rows, then the gather rows of
  // stage t_data into ring slot ld_slot
  int t_data = kt0, t_desc = kt0, ld_slot = 0;
  unsigned xv[LX];                   // gather offsets of stage t_data (from its descriptors)
  u32x2 dreg[LX];
  auto piece = [&](int k) {          // k is a constant after unrolling
#ifdef DSL_ABLATE_BUILD
    if ((p.dbg & 1) && t_data >= kt0 + NST) return;      // no DMA after the ring's first fill
#endif
    if (k < NDSC) {
      const unsigned v = (unsigned)(t_desc * (KS * 8) + k * 256 + lane * 4);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_t, (lptr_t)(smem + DESC_BASE + (t_desc & (DR - 1)) * (KS * 8) + k * 256), 4, v, 0, 0, 0);
    } else if (k < NDSC + LY) {
      const int i = k - NDSC;
      const unsigned v = t_data < kt1 ? yv[i] : 0x80000000u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_y, (lptr_t)(smem + ld_slot * STAGE + (wave + NW * i) * 1024), 16, v, 0, 0, 0);
    } else {
      const int i = k - NDSC - LY;
      const unsigned v = t_data < kt1 ? xv[i] : 0x80000000u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lptr_t)(smem + ld_slot * STAGE + TILE_Y + (wave + NW * i) * 1024), 16, v, 0, 0, 0);
    }
  };
  auto desc_read = [&]() {           // descriptors of stage t_data (landed and barrier-published) -> registers
#pragma unroll
    for (int i = 0; i < LX; ++i) dreg[i] = lds_read_b64_asm(lds_base + xdaddr[i] + (unsigned)((t_data & (DR - 1)) * (KS * 8)));
  };
  auto desc_use = [&]() {            // ... -> this lane's gather offsets (after the wait that covers desc_read)
#pragma unroll
    for (int i = 0; i < LX; ++i) {
      pin2(dreg[i]);
      const unsigned info = dreg[i][1];
      const unsigned px = (unsigned)((int)dreg[i][0] + tr * (int)(info >> 16) + ts);
      xv[i] = (info & sel) == sel ? px * ldx2 + xcol[i] : 0x80000000u;
    }
  };
  auto advance = [&]() {             // the stage's pieces are all issued
    ++t_data;
    ++t_desc;
    ld_slot = (ld_slot + 1 == NST) ? 0 : ld_slot + 1;
#pragma unroll
    for (int i = 0; i < LY; ++i) yv[i] += y_step;
  };
  // one k-step's MFMAs with the DMA pieces [lo, hi) of the stage issued between them: one piece behind every second MFMA
  // (everything here is pinned in source order)
  auto block = [&](int f, int lo, int hi, int rd_kk) {
    bf16x8 a[CT], b[IT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      union { struct { u32x2 l, h; } s; bf16x8 v; } u;
      u.s.l = fa[f][ct].lo;
      u.s.h = fa[f][ct].hi;
      a[ct] = u.v;
    }
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      union { struct { u32x2 l, h; } s; bf16x8 v; } u;
      u.s.l = fb[f][it].lo;
      u.s.h = fb[f][it].hi;
      b[it] = u.v;
    }
    __builtin_amdgcn_sched_barrier(0);
    int k = lo;
#pragma unroll
    for (int m = 0; m < NM; ++m) {
#ifdef DSL_ABLATE_BUILD
      if (!(p.dbg & 2))
#endif
      acc[m / IT][m % IT] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m / IT], b[m % IT], acc[m / IT][m % IT], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      // the next k-step's fragment reads ride between the MFMAs (a burst of NR reads in front of the block keeps the wave
      // on LDS issue for as long as the block's MFMAs take: measured, the two simply added up)
#pragma unroll
      for (int j = 0; j < NR; ++j)
        if (j >= m * NR / NM && j < (m + 1) * NR / NM) rd(rd_kk, f ^ 1, j);
      if ((m & 1) == 1 && k < hi) {
        piece(k);
        ++k;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int j = 0; j < P; ++j)
      if (k + j < hi) piece(k + j);
    __builtin_amdgcn_sched_barrier(0);
  };
  constexpr int PA = (KK - 1) * P / KK;          // pieces issued before the stage's barrier (k-steps 0 .. KK-2)

  // ---- prologue: the descriptors of the first NST-1 stages, then NST-1 whole stages (each with the descriptor pieces of a
  // later stage, so that every stage - prologue or not - is exactly P DMA instructions: the waits below count in stages)
#pragma unroll
  for (int j = 0; j < NST - 1; ++j) {
#pragma unroll
    for (int k = 0; k < NDSC; ++k) piece(k);
    ++t_desc;
  }
  wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int j = 0; j < NST - 1; ++j) {
    desc_read();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    desc_use();
#pragma unroll
    for (int k = 0; k < P; ++k) piece(k);
    advance();
  }
  // steady state from here: stage s fetches the descriptors of stage s + DLEAD (t_desc) and the data of stage s + NST - 1 (t_data)
  wait_vmcnt<(NST - 2) * P>();                   // stage kt0 landed, and the descriptors fetched with it (stage kt0 + NST - 1's)
  __builtin_amdgcn_s_barrier();
  desc_read();
  set_read_stage(lds_base);
  issue(0, 0);

  unsigned dbr[8];
  int slot_c = 0;
  for (int s = kt0; s < kt1; ++s) {
    const unsigned st = lds_base + slot_c * STAGE;
    const int nslot = (slot_c + 1 == NST) ? 0 : slot_c + 1;
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) {
      const int f = kk & 1;
      wait_lds(f);
      if (kk == 0) desc_use();
      if (kk == 1 && do_db) {          // the dY column sums of this stage (issued in k-step 0)
#pragma unroll
        for (int j = 0; j < DB_ROWS; ++j) {
          pin1(dbr[j]);
          db_lo += __uint_as_float(dbr[j] << 16);
          db_hi += __uint_as_float(dbr[j] & 0xffff0000u);
        }
      }
      if (kk == 0 && do_db) {
#pragma unroll
        for (int j = 0; j < DB_ROWS; ++j) {
          const int row = db_rg * DB_ROWS + j;
          dbr[j] = lds_read_b32_asm(st + row * YB + ((((db_cp * 4) >> 6) ^ (row & 3)) << 6) + ((db_cp * 4) & 63));
        }
      }
      if (kk == KK - 1) {
        wait_vmcnt<(NST - 3) * P + PA>();      // stage s+1 landed (and every older DMA of this wave)
#ifdef DSL_ABLATE_BUILD
        if (!(p.dbg & 16))
#endif
        __builtin_amdgcn_s_barrier();          // ... for every wave; every wave is done reading stage s
      }
      __builtin_amdgcn_sched_barrier(0);
      // this k-step's MFMAs, between them the fragment reads of the next k-step (the last k-step: of the next stage's first,
      // behind the barrier above) and this k-step's share of the stage's DMA pieces
      if (kk < KK - 1) {
        block(f, kk * P / KK, (kk + 1) * P / KK, kk + 1);
      } else {
        set_read_stage(lds_base + nslot * STAGE);
        block(f, kk * P / KK, P, 0);
      }
      if (kk == KK - 1) {
        advance();
        desc_read();                           // descriptors of the stage fetched next (published by the barrier above)
      }
    }
    slot_c = nslot;
  }
  wait_vmcnt<0>();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#ifdef DSL_ABLATE_BUILD
  if (p.dbg & 4) return;
#endif

  if (do_db) {               // fold the row groups in a fixed order; one value per column leaves the workgroup
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);
    red[(db_rg * DB_PAIRS + db_cp) * 2] = db_lo;
    red[(db_rg * DB_PAIRS + db_cp) * 2 + 1] = db_hi;
    __syncthreads();
    if (tid < BCO) {
      float sacc = 0.f;
#pragma unroll
      for (int r = 0; r < DB_RG; ++r) sacc += red[(r * DB_PAIRS + (tid >> 1)) * 2 + (tid & 1)];
      const int co = co0 + tid;
      if (p.direct) {
        if (co < p.cd) db_p[co] = sacc;
      } else {
        p.dbws[((long long)sp * p.group + member) * p.cyp + co] = sacc;
      }
    }
  }
  const int frow = lane & 31, fhalf = lane >> 5;
  if (p.direct) {
    float* dw_p = p.dwv[0];
    const float* sc_p = p.scalev[0];
#pragma unroll
    for (int g = 1; g < DSL_MAX_GROUP; ++g) {
      dw_p = member == g ? p.dwv[g] : dw_p;
      sc_p = member == g ? p.scalev[g] : sc_p;
    }
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int it = 0; it < IT; ++it) {
        const long long col = (long long)tap * p.cs + ci0 + wave_ci * (32 * IT) + it * 32 + frow;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          const int co = co0 + wave_co * (32 * CT) + ct * 32 + (j & 3) + 8 * (j >> 2) + 4 * fhalf;
          if (co < p.cd) dw_p[(long long)co * p.krow + col] = sc_p ? acc[ct][it][j] * sc_p[co] : acc[ct][it][j];
        }
      }
    return;
  }
#pragma unroll
  for (int ct = 0; ct < CT; ++ct)
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      const long long col = (long long)tap * p.cs + ci0 + wave_ci * (32 * IT) + it * 32 + frow;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int co = co0 + wave_co * (32 * CT) + ct * 32 + (j & 3) + 8 * (j >> 2) + 4 * fhalf;
        p.ws[(((long long)sp * p.group + member) * p.cyp + co) * p.krow + col] = acc[ct][it][j];
      }
    }
}

template <int BCO, int BCI, int WCO, int WCI, int KS, int NST>
__global__ __launch_bounds__(64 * WCO * WCI) void wgrad_pipe_kernel(const WgK p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // persistent form (grid < work items, a multiple of 8 so that a block keeps its XCD): the launch never holds more CUs than
  // its workgroup budget, whatever the number of tiles and splits - the caller's chain of small launches keeps the rest
  const int total = p.chunk * 8;
  for (int vb = (int)blockIdx.x; vb < total; vb += (int)gridDim.x) {
    wgrad_pipe_body<BCO, BCI, WCO, WCI, KS, NST>(p, vb, smem);
    __syncthreads();
  }
}

// Several weight-gradient launches of ONE tile configuration as one grid (dsl_conv2d_wgrad_multi): sub-launch s owns the
// blocks [wg_end[s-1], wg_end[s]) (multiples of 8, so a block's XCD is the same as in a launch of its own); its WgK comes
// from a table in device memory, read once with scalar loads before the K loop.  The host orders the sub-launches by
// decreasing work per workgroup: the hardware dispatches blocks in index order, so the short ones fill the tail.
constexpr int kMaxMulti = DSL_MAX_MULTI;
struct WgMultiHdr {
  int nsub;
  int wg_end[kMaxMulti];
};
template <int BCO, int BCI, int WCO, int WCI, int KS, int NST>
__global__ __launch_bounds__(64 * WCO * WCI) void wgrad_glds_multi_kernel(const WgMultiHdr h, const WgK* __restrict__ tab) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  int sub = 0, start = 0;
#pragma unroll
  for (int s = 1; s < kMaxMulti; ++s) {
    const bool in = s < h.nsub && (int)blockIdx.x >= h.wg_end[s - 1];
    sub = in ? s : sub;
    start = in ? h.wg_end[s - 1] : start;
  }
  const WgK p = tab[sub];
  wgrad_glds_body<BCO, BCI, WCO, WCI, KS, NST>(p, (int)blockIdx.x - start, smem);
}

template <int BCO, int BCI, int WCO, int WCI, int KS, int NST>
__global__ __launch_bounds__(64 * WCO * WCI) void wgrad_pipe_multi_kernel(const WgMultiHdr h, const WgK* __restrict__ tab) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  int sub = 0, start = 0;
#pragma unroll
  for (int s = 1; s < kMaxMulti; ++s) {
    const bool in = s < h.nsub && (int)blockIdx.x >= h.wg_end[s - 1];
    sub = in ? s : sub;
    start = in ? h.wg_end[s - 1] : start;
  }
  const WgK p = tab[sub];
  wgrad_pipe_body<BCO, BCI, WCO, WCI, KS, NST>(p, (int)blockIdx.x - start, smem);
}

// persistent form of the multi launch: `grid` (a multiple of 8) workgroups walk the block list
template <int BCO, int BCI, int WCO, int WCI, int KS, int NST>
__global__ __launch_bounds__(64 * WCO * WCI) void wgrad_pipe_multi_persist_kernel(const WgMultiHdr h, const WgK* __restrict__ tab, int total) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  for (int vb = (int)blockIdx.x; vb < total; vb += (int)gridDim.x) {
    int sub = 0, start = 0;
#pragma unroll
    for (int s = 1; s < kMaxMulti; ++s) {
      const bool in = s < h.nsub && vb >= h.wg_end[s - 1];
      sub = in ? s : sub;
      start = in ? h.wg_end[s - 1] : start;
    }
    const WgK p = tab[sub];
    wgrad_pipe_body<BCO, BCI, WCO, WCI, KS, NST>(p, vb - start, smem);
    __syncthreads();
  }
}

// scheduled form (round 4, wgrad_plan_*): the host assigns every valid virtual block to a workgroup (longest-processing-time
// first inside the block's XCD class), sched[r * gridDim.x + b] = the r-th block of workgroup b or -1; what a workgroup computes
// for a block, and hence every result, is the same as in the stride form
template <int BCO, int BCI, int WCO, int WCI, int KS, int NST>
__global__ __launch_bounds__(64 * WCO * WCI) void wgrad_pipe_multi_sched_kernel(const WgMultiHdr h, const WgK* __restrict__ tab,
                                                                               const short* __restrict__ sched, int rounds) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  for (int r = 0; r < rounds; ++r) {
    const int vb = __builtin_amdgcn_readfirstlane((int)sched[r * (int)gridDim.x + (int)blockIdx.x]);
    if (vb < 0) break;
    int sub = 0, start = 0;
#pragma unroll
    for (int s = 1; s < kMaxMulti; ++s) {
      const bool in = s < h.nsub && vb >= h.wg_end[s - 1];
      sub = in ? s : sub;
      start = in ? h.wg_end[s - 1] : start;
    }
    const WgK p = tab[sub];
    wgrad_pipe_body<BCO, BCI, WCO, WCI, KS, NST>(p, vb - start, smem);
    __syncthreads();
  }
}

// the reduce passes of a multi launch: entry e (one member of one sub-launch with more than one split) owns the blocks
// [blk_start, blk_start + nblk)
struct RedEnt {
  const float* ws;         // this member's first partial: ws + member * cyp * krow
  float* dw;
  const float* scale;
  float* db;               // bias gradient: the splits' column-sum partials dbws[sp * dbstride + c] folded in order (or NULL)
  const float* dbws;
  long long dbstride;
  long long krow, sstride;
  int splits, cd, blk_start, nblk;
};
__global__ void wgrad_reduce_multi_kernel(const RedEnt* __restrict__ tab, int n) {
  int e = 0;
  for (int i = 1; i < n; ++i) e = (int)blockIdx.x >= tab[i].blk_start ? i : e;
  const RedEnt r = tab[e];
  const int lb = (int)blockIdx.x - r.blk_start;
  if (lb == 0 && r.db)
    for (int c = threadIdx.x; c < r.cd; c += blockDim.x) {
      float sacc = 0.f;
      for (int sp = 0; sp < r.splits; ++sp) sacc += r.dbws[sp * r.dbstride + c];
      r.db[c] = sacc;
    }
  const long long total4 = (long long)r.cd * r.krow / 4;
  for (long long i = (long long)lb * blockDim.x + threadIdx.x; i < total4; i += (long long)r.nblk * blockDim.x) {
    const long long el = i * 4;
    const int co = (int)(el / r.krow);
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    const float* base = r.ws + el;
    int sp = 0;
    for (; sp + 4 <= r.splits; sp += 4) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(base + sp * r.sstride);
      const f32x4 b = *reinterpret_cast<const f32x4*>(base + (sp + 1) * r.sstride);
      const f32x4 c = *reinterpret_cast<const f32x4*>(base + (sp + 2) * r.sstride);
      const f32x4 d = *reinterpret_cast<const f32x4*>(base + (sp + 3) * r.sstride);
      s += (a + b) + (c + d);
    }
    for (; sp < r.splits; ++sp) s += *reinterpret_cast<const f32x4*>(base + sp * r.sstride);
    if (r.scale) s *= r.scale[co];
    *reinterpret_cast<f32x4*>(r.dw + el) = s;
  }
}

struct RedK {
  float* dw[DSL_MAX_GROUP];
  const float* scale[DSL_MAX_GROUP];
  float* db[DSL_MAX_GROUP];        // bias-gradient vectors: summed from dbws, or cleared for the column-sum kernel that follows (or NULL)
  const float* dbws;               // [split][member][cy] column-sum partials of the DMA kernels (NULL: v1 kernel)
};

// sums the split partials ws[split][member][cy][krow] of member blockIdx.y into its dW (x scale)
__global__ void wgrad_reduce_kernel(const float* __restrict__ ws, const RedK r, int splits, int group, int cy, int cd,
                                    long long krow) {
  const int member = blockIdx.y;
  float* __restrict__ dw = r.dw[0];
  const float* __restrict__ scale = r.scale[0];
#pragma unroll
  for (int g = 1; g < DSL_MAX_GROUP; ++g) {
    dw = member == g ? r.dw[g] : dw;
    scale = member == g ? r.scale[g] : scale;
  }
  if (blockIdx.x == 0) {
    float* db = r.db[0];
#pragma unroll
    for (int g = 1; g < DSL_MAX_GROUP; ++g) db = member == g ? r.db[g] : db;
    if (db)
      for (int c = threadIdx.x; c < cd; c += blockDim.x) {
        float sacc = 0.f;
        if (r.dbws)                 // in-kernel column sums: fold the splits in order; else cleared for the column-sum pass
          for (int sp = 0; sp < splits; ++sp) sacc += r.dbws[((long long)sp * group + member) * cy + c];
        db[c] = sacc;
      }
  }
  const long long total4 = (long long)cd * krow / 4;
  const long long sstride = (long long)group * cy * krow;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total4;
       i += (long long)gridDim.x * blockDim.x) {
    const long long e = i * 4;
    const int co = (int)(e / krow);
    const long long k = e - (long long)co * krow;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    const float* base = ws + ((long long)member * cy + co) * krow + k;
    int sp = 0;
    for (; sp + 4 <= splits; sp += 4) {      // 4 independent loads in flight per thread
      const f32x4 a = *reinterpret_cast<const f32x4*>(base + sp * sstride);
      const f32x4 b = *reinterpret_cast<const f32x4*>(base + (sp + 1) * sstride);
      const f32x4 c = *reinterpret_cast<const f32x4*>(base + (sp + 2) * sstride);
      const f32x4 d = *reinterpret_cast<const f32x4*>(base + (sp + 3) * sstride);
      s += (a + b) + (c + d);
    }
    for (; sp < splits; ++sp) s += *reinterpret_cast<const f32x4*>(base + sp * sstride);
    if (scale) s *= scale[co];
    *reinterpret_cast<f32x4*>(dw + e) = s;
  }
}

}  // namespace

// ================================================================================================
// host side
// ================================================================================================
namespace {
// DMA-to-LDS tile configurations {BCO, BPX, workgroups per CU, ring depth}
struct TileCfg { int bco, bpx, occ, nst, wpx; };     // wpx: pixel-waves of the pipelined kernel (epilogue staging = 32*wpx pixels)
constexpr int kNumCfg = 9;
const TileCfg kCfgs[kNumCfg] = {{256, 192, 1, 2, 2}, {256, 128, 1, 3, 2}, {128, 256, 1, 3, 4}, {128, 128, 2, 2, 4}, {64, 256, 2, 2, 8},
                                {128, 64, 2, 3, 2},
                                // small tiles for the layers with few pixels (layer3/4: 8 400 / 2 100 pixels at N = 2): enough
                                // workgroups to use every CU without split-K partials, several resident per CU
                                {64, 64, 3, 3, 2}, {64, 128, 2, 3, 4},
                                // 256 x 256 (round 4 experiment, DSL_CONV_256=1 or forced): 14 % fewer DMA bytes and 10 % fewer fragment
                                // reads per MFMA than 256 x 192, 175 instead of 234 workgroups on the head shape
                                {256, 256, 1, 2, 2}};

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
  static const double kEff[kNumCfg] = {1.0, 1.0, 1.0, 0.7, 0.95, 0.9, 1.3, 1.25, 1.0};     // 6, 7: measured best on one shape of tools/bench_conv.py only
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
  int force = (d->flags >> 8) & 15;                // test hook: 1..9 = tile config, 10 = 256 x 192 with the half-stage K loop, 15 = v1 kernel
  if (force == 10) force = 1;
  const int force_split = (d->flags >> 12) & 15;   // test hook: split-K factor
  int pick = -1, splits = 1;
  long long src_px = 0;
  for (int sg = 0; sg < d->nseg; ++sg) src_px += (long long)d->n * d->sh[sg] * d->sw[sg];
  // the DMA kernels address the source with 32-bit buffer offsets and per-axis tap masks
  const long long lds_ = d->lds > 0 ? d->lds : d->cs;
  const bool dma_ok = conv_v2_only(d) || (d->kh <= 8 && d->kw <= 8 && src_px * lds_ * 2 + (long long)d->kw * lds_ * 2 < 0x7fff0000LL);
  const bool smallc_pipe = smallc && d->cd_pad % 64 == 0 && !getenv("DSL_STEM_V1");    // stem: pipelined kernel, 64-cout tile
  const bool v1_only = ((smallc && !smallc_pipe) || (d->flags & DSL_CONV_RELU_IN) || !dma_ok) && !(d->flags & DSL_CONV_FP8);
  static const int mode_256 = [] { const char* e = getenv("DSL_CONV_256"); return e ? atoi(e) : 0; }();
  const bool use_256 = mode_256 != 0;
  // (2: every 256-cout convolution over >= 40 000 pixels takes the 256 x 256 tile whatever the model says - the CU-time experiment)
  if (mode_256 == 2 && !v1_only && force == 0 && !smallc && !(d->flags & DSL_CONV_FP8) && !conv_v2_only(d) && d->cd_pad % 256 == 0 &&
      px >= 40000 && !d->gn_x && !getenv("DSL_CONV_V2") && !getenv("DSL_CONV_KT")) {
    *pick_out = 8;
    *splits_out = 1;
    return;
  }
  if (!v1_only && force != 15) {
    double best = 1e300;
    const bool out_f32 = (d->flags & DSL_CONV_OUT_F32) != 0;
    for (int c = 0; c < kNumCfg; ++c) {
      if (d->cd_pad % kCfgs[c].bco) continue;
      if (force >= 1 && force <= kNumCfg && force - 1 != c) continue;
      if (c >= 5 && conv_v2_only(d)) continue;       // the small tiles exist for the pipelined kernel only
      if (c == 8 && force - 1 != 8 && !use_256) continue;
      if (c == 8 && (smallc || getenv("DSL_CONV_V2") || getenv("DSL_CONV_KT"))) continue;
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
        if (c == 8 && force - 1 != 8 && !use_256) continue;
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
  if (d->flags & (DSL_CONV_SMALL_C | DSL_CONV_FP8 | DSL_CONV_OUT_F32 | DSL_CONV_ADD_UPSAMPLE | DSL_CONV_RELU_IN)) return false;
  if (d->cs % 64 || d->cd % 8 || d->cd_pad % 64 || d->os != 1 || d->addend) return false;
  if (d->mask && ((d->flags & DSL_CONV_MASK_FIRST) || ((d->flags & DSL_CONV_MASK_LAST) && (d->flags & DSL_CONV_RELU_OUT)))) return false;
  for (int s = 0; s < d->nseg; ++s)
    if (d->gh[s] != d->dh[s] || d->gw[s] != d->dw[s]) return false;
  if (conv_v2_only(d) || getenv("DSL_CONV_V2") || getenv("DSL_CONV_KT") || getenv("DSL_CONV_LOADER")) return false;
  dsl_conv_desc t = *d;
  t.gn_ws = (void*)1;
  int pick, splits;
  conv_choose(&t, conv_pixels(d), d->kh * d->kw * (d->cs / 64), &pick, &splits);
  if (pick < 0 || splits != 1) return false;
  if (d->gn_x) {      // the backward records: instantiated for the 256-cout tiles only (256 x 192, 256 x 128).  Their workgroups own a
    // CU anyway (114 / 147 KB of LDS); the 128 x 128 tile with the ~60 extra registers (158) keeps ONE workgroup per CU instead of
    // two and the launch takes twice as long (N = 3 head: 100 -> 212 us, profiles/r04_rla_timeline.txt) - there the separate pass stays
    if (pick != 0 && pick != 1) return false;
    if (getenv("DSL_CONV_TALL") || getenv("DSL_CONV_HOLD") || d->ldd != d->cd) return false;
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
  // ---- 3x3 / 1, 64 -> 64, BatchNorm + ReLU epilogue (layer1's middle convolutions): the activation-stationary kernel of patch3.hip
  // (pixel tile + halo staged once, nine taps out of LDS, weights in registers) instead of nine K tiles of implicit GEMM
  if (!smallc && !fp8 && d->mode == 0 && d->nseg == 1 && d->cs == 64 && d->cd == 64 && d->cd_pad == 64 && d->kh == 3 && d->kw == 3 &&
      d->stride == 1 && d->pad == 1 && d->os == 1 && k.ident && (d->flags & 0xff & ~DSL_CONV_RELU_OUT) == 0 && (d->flags >> 8) == 0 &&
      d->scale && d->bias && !d->addend && !d->mask && d->ldd % 8 == 0 && d->sh[0] == d->gh[0] && d->sw[0] == d->gw[0]) {
    // Opt-in (DSL_PATCH3=1): standalone it is 1.46 x the implicit GEMM (31.1 -> 21.2 us, N = 2, same bits), in the training step it is
    // not faster (three alternations: 421.5 img/s without, 418.7 with - one 89 KB / 474-register workgroup per CU shares a CU with
    // nothing, and the frozen prefix runs beside the previous step's backward tail)
    const char* e = getenv("DSL_PATCH3");
    if (e && atoi(e) != 0) {
      int prof = -1;
      if (dsl_prof_active()) prof = dsl_prof_begin(2, conv_algo_flops(d, px), st, conv_algo_bytes(d, px));
      const int rc = dsl_conv3x3_c64_patch(d->src, k.lds, d->wgt, d->scale, d->bias, d->dst, d->ldd, d->n, d->gh[0], d->gw[0],
                                           (d->flags & DSL_CONV_RELU_OUT) ? 1 : 0, stream);
      dsl_prof_end(prof, st);
      return rc;
    }
  }
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
    static const bool force_v2 = getenv("DSL_CONV_V2") != nullptr;
    const bool force_v2_kernel = force_v2 || conv_v2_only(d);
    k.gx = (int)grid.x;
    k.gy = (int)grid.y;
    k.xcd_chunk = (int)((grid.x * grid.y * grid.z + 7) / 8);
    size_t lds = (size_t)c.nst * (c.bco + c.bpx) * 128;
    if (!force_v2_kernel) {                // the pipelined kernel stages its epilogue in LDS: 32*wpx pixel rows of fp32
      const size_t stg = (size_t)32 * c.wpx * (c.bco * 4 + 16);
      if (stg > lds) lds = stg;
      if (pick == 8) {                     // ... or the whole tile as bf16 rows + the GroupNorm scratch (conv_epi_ring)
        const size_t pure = (size_t)c.bpx * (c.bco * 2 + 16) + 512 * 8;
        if (pure > lds) lds = pure;
      }
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
    // v4 (K-tile-granular fragment pipeline, lean per-tile bookkeeping) for the one-MFMA-column tiles: DSL_CONV_KT = 0 off
    // (default), 1 where the launch has at most one workgroup per CU, 2 always.  Measured and rejected (tools/conv_cost.py,
    // tools/pmc_conv.sh): it halves the instructions per K tile and takes the LDS latency off the k-step chain (SQ_WAIT_ANY
    // 47 % -> 32 % of wave cycles), but the layer3 / layer4 shapes run within 4 % of v3 either way and the step is slower (372
    // vs 386 img/s: one workgroup per CU) - these launches are bound by the 64 B/clk/CU L2 -> LDS path ((BCO + BPX) x 128 B per
    // K tile: 32 KB per 512 MFMA cycles for 128 x 128), not by issue or latency; only operand reuse across taps would cut that.
    static const int kt_mode = [] { const char* e = getenv("DSL_CONV_KT"); return e ? atoi(e) : 0; }();
    static const int tall = [] { const char* e = getenv("DSL_CONV_TALL"); return e ? atoi(e) : 0; }();
    static const int hold = [] { const char* e = getenv("DSL_CONV_HOLD"); return e ? atoi(e) : 0; }();
    static const int ldw = [] { const char* e = getenv("DSL_CONV_LOADER"); return e ? atoi(e) : 0; }();
#define LAUNCH3L(A, B, C_, D, S_, L_)                                                                          \
  do {                                                                                                        \
    static bool attr_set3l = false;                                                                           \
    if (!attr_set3l) {                                                                                        \
      hipFuncSetAttribute((const void*)conv_pipe_kernel<A, B, C_, D, S_, false, 1, L_>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                          (int)lds);                                                                          \
      attr_set3l = true;                                                                                      \
    }                                                                                                         \
    hipLaunchKernelGGL((conv_pipe_kernel<A, B, C_, D, S_, false, 1, L_>), dim3(8 * k.xcd_chunk), dim3(64 * (C_ * D + L_)), lds, st, k); \
  } while (0)
#define LAUNCH3H(A, B, C_, D, S_, H_)                                                                          \
  do {                                                                                                        \
    static bool attr_set3h = false;                                                                           \
    if (!attr_set3h) {                                                                                        \
      hipFuncSetAttribute((const void*)conv_pipe_kernel<A, B, C_, D, S_, false, H_>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                          (int)lds);                                                                          \
      attr_set3h = true;                                                                                      \
    }                                                                                                         \
    hipLaunchKernelGGL((conv_pipe_kernel<A, B, C_, D, S_, false, H_>), dim3(8 * k.xcd_chunk), dim3(64 * C_ * D), lds, st, k); \
  } while (0)
#define LAUNCH3G(A, B, C_, D, S_)                                                                              \
  do {                                                                                                        \
    static bool attr_set3g = false;                                                                           \
    if (!attr_set3g) {                                                                                        \
      hipFuncSetAttribute((const void*)conv_pipe_kernel<A, B, C_, D, S_, false, 1, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                          (int)lds);                                                                          \
      attr_set3g = true;                                                                                      \
    }                                                                                                         \
    hipLaunchKernelGGL((conv_pipe_kernel<A, B, C_, D, S_, false, 1, 0, true>), dim3(8 * k.xcd_chunk), dim3(64 * C_ * D), lds, st, k); \
  } while (0)
    const long long n_wg = (long long)grid.x * grid.y * grid.z;
    const bool use_kt = !force_v2_kernel && !smallc && (pick == 3 || pick == 5 || pick == 6 || pick == 7) &&
                        (kt_mode == 2 || (kt_mode == 1 && n_wg <= 256LL * c.occ));
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
    } else if (use_kt) {
      constexpr int KT_NST = 3;
      size_t ldk = (size_t)KT_NST * (c.bco + c.bpx) * 128;
      const size_t stg = (size_t)32 * c.wpx * (c.bco * 4 + 16);
      if (stg > ldk) ldk = stg;
#define LAUNCHK(A, B, C_, D)                                                                                  \
  do {                                                                                                        \
    static bool attr_k = false;                                                                               \
    if (!attr_k) {                                                                                            \
      hipFuncSetAttribute((const void*)conv_kt_kernel<A, B, C_, D, KT_NST>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                          (int)ldk);                                                                          \
      attr_k = true;                                                                                          \
    }                                                                                                         \
    hipLaunchKernelGGL((conv_kt_kernel<A, B, C_, D, KT_NST>), dim3(8 * k.xcd_chunk), dim3(64 * C_ * D), ldk, st, k); \
  } while (0)
      switch (pick) {
        case 3: LAUNCHK(128, 128, 2, 4); break;
        case 5: LAUNCHK(128, 64, 2, 2); break;
        case 6: LAUNCHK(64, 64, 1, 2); break;
        default: LAUNCHK(64, 128, 1, 4); break;
      }
#undef LAUNCHK
    } else if (force_v2_kernel) {
      switch (pick) {
        case 0: LAUNCH2(256, 192, 4, 2, 2); break;
        case 1: LAUNCH2(256, 128, 4, 2, 3); break;
        case 2: LAUNCH2(128, 256, 2, 4, 3); break;
        case 3: LAUNCH2(128, 128, 2, 2, 2); break;
        default: LAUNCH2(64, 256, 1, 4, 2); break;
      }
    } else {
      // the half-stage K loop (conv_h4_kernel) for the 256 x 192 tile: DSL_CONV_H4=1 (off by default: neutral in the step,
      // LAB_NOTES.md), or tile hook 10 of the tests
      static const int h4_env = [] { const char* e = getenv("DSL_CONV_H4"); return e ? atoi(e) : 0; }();
      const bool h4 = pick == 0 && !smallc && !(tall & 1) && !hold && !ldw && (h4_env != 0 || ((d->flags >> 8) & 15) == 10);
      if (h4) {
        lds = (size_t)4 * (256 + 256) * 64;
#define LAUNCHH(G_)                                                                                           \
  do {                                                                                                        \
    static bool attr_h = false;                                                                               \
    if (!attr_h) {                                                                                            \
      hipFuncSetAttribute((const void*)conv_h4_kernel<256, 192, 4, 2, G_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
      attr_h = true;                                                                                          \
    }                                                                                                         \
    hipLaunchKernelGGL((conv_h4_kernel<256, 192, 4, 2, G_>), dim3(8 * k.xcd_chunk), dim3(512), lds, st, k);   \
  } while (0)
        if (k.gnx) LAUNCHH(true); else LAUNCHH(false);
#undef LAUNCHH
      } else if (k.gnx) {            // (conv_gn_ok: one of the two tiles below, none of the variant knobs)
        if (pick == 0) LAUNCH3G(256, 192, 4, 2, 2); else LAUNCH3G(256, 128, 4, 2, 3);
      } else
      switch (pick) {
        case 0:
          if (tall & 1) LAUNCH3(256, 192, 2, 2, 2); else if (hold) LAUNCH3H(256, 192, 4, 2, 2, 2); else if (ldw) LAUNCH3L(256, 192, 4, 2, 2, 4); else LAUNCH3(256, 192, 4, 2, 2);
          break;
        case 1:
          if (tall & 2) LAUNCH3(256, 128, 2, 2, 3); else if (hold) LAUNCH3H(256, 128, 4, 2, 3, 2); else if (ldw) LAUNCH3L(256, 128, 4, 2, 3, 4); else LAUNCH3(256, 128, 4, 2, 3);
          break;
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
        case 8: LAUNCH3(256, 256, 4, 2, 2); break;
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

// ---- v3 weight gradient: per-geometry pixel descriptor tables (PixDesc), built on the host once per geometry and kept
// in device memory for the life of the process (a few hundred KB per geometry; a training run has ~20 geometries)
namespace {
constexpr int kWgV3KS = 32, kWgV3DR = 16;
struct PixTabEntry {
  int dev, nseg, n, stride, pad, kh, kw;
  int gh[DSL_MAX_SEG], gw[DSL_MAX_SEG], sh[DSL_MAX_SEG], sw[DSL_MAX_SEG];
  void* ptr;
  unsigned bytes;
};
std::mutex g_pixtab_mu;
std::vector<PixTabEntry> g_pixtabs;

int wgrad_slots();
bool wgrad_persist() {
  // measured (tools/exp_env.sh, bench.py N = 2): persistent grids of 128 workgroups +2.3 % (96 .. 160 within 0.3 %, 64: -1 %)
  static const bool on = [] { const char* e = getenv("DSL_WGRAD_PERSIST"); return !e || atoi(e) != 0; }();
  return on;
}
bool wgrad_v3_enabled() {
  static const bool on = [] { const char* e = getenv("DSL_WGRAD_V3"); return !e || atoi(e) != 0; }();
  return on;
}
// ring depth of the v3 kernel per tile configuration (1: 256x256 -> 4 x 32 KB; 2, 3: 24 KB stages)
int wgrad_v3_nst(int cfg) { return cfg == 1 ? 4 : 5; }
bool wgrad_v3_ok(const dsl_wgrad_desc* d, int cfg) {
  if (!wgrad_v3_enabled() || cfg < 1 || cfg > 3 || d->kh > 8 || d->kw > 8) return false;
  long long px = 0, xo = 0;
  for (int s = 0; s < d->nseg; ++s) {
    px += (long long)d->n * d->gh[s] * d->gw[s];
    xo += (long long)d->n * d->sh[s] * d->sw[s];
    if (d->sw[s] >= 65536) return false;
  }
  const long long ldx = d->ldx > 0 ? d->ldx : d->cs;
  return px * d->cy * 2 < 0x7fff0000LL && xo * ldx * 2 < 0x7fff0000LL && px < (1 << 20);
}
// returns the device table of d's geometry (building it on first use), or nullptr on failure
const void* wgrad_pixtab(const dsl_wgrad_desc* d, unsigned* bytes) {
  int dev = 0;
  hipGetDevice(&dev);
  std::lock_guard<std::mutex> lk(g_pixtab_mu);
  for (const PixTabEntry& e : g_pixtabs) {
    if (e.dev != dev || e.nseg != d->nseg || e.n != d->n || e.stride != d->stride || e.pad != d->pad || e.kh != d->kh || e.kw != d->kw) continue;
    bool same = true;
    for (int s = 0; s < d->nseg; ++s)
      same = same && e.gh[s] == d->gh[s] && e.gw[s] == d->gw[s] && e.sh[s] == d->sh[s] && e.sw[s] == d->sw[s];
    if (same) { *bytes = e.bytes; return e.ptr; }
  }
  long long px = 0;
  for (int s = 0; s < d->nseg; ++s) px += (long long)d->n * d->gh[s] * d->gw[s];
  std::vector<PixDesc> h((size_t)px);
  long long xoff = 0;
  size_t i = 0;
  for (int s = 0; s < d->nseg; ++s) {
    const int sh = d->sh[s], sw = d->sw[s];
    for (int img = 0; img < d->n; ++img)
      for (int y = 0; y < d->gh[s]; ++y)
        for (int x = 0; x < d->gw[s]; ++x) {
          const int y0 = y * d->stride - d->pad, x0 = x * d->stride - d->pad;
          unsigned ym = 0, xm = 0;
          for (int r = 0; r < d->kh; ++r) if ((unsigned)(y0 + r) < (unsigned)sh) ym |= 1u << r;
          for (int c = 0; c < d->kw; ++c) if ((unsigned)(x0 + c) < (unsigned)sw) xm |= 1u << c;
          h[i].base = (int32_t)(xoff + ((long long)img * sh + y0) * sw + x0);
          h[i].info = ((unsigned)sw << 16) | (xm << 8) | ym;
          ++i;
        }
    xoff += (long long)d->n * sh * sw;
  }
  PixTabEntry e;
  memset(&e, 0, sizeof(e));
  e.dev = dev; e.nseg = d->nseg; e.n = d->n; e.stride = d->stride; e.pad = d->pad; e.kh = d->kh; e.kw = d->kw;
  for (int s = 0; s < d->nseg; ++s) { e.gh[s] = d->gh[s]; e.gw[s] = d->gw[s]; e.sh[s] = d->sh[s]; e.sw[s] = d->sw[s]; }
  e.bytes = (unsigned)(px * sizeof(PixDesc));
  if (hipMalloc(&e.ptr, e.bytes + 256) != hipSuccess) return nullptr;
  if (hipMemcpy(e.ptr, h.data(), e.bytes, hipMemcpyHostToDevice) != hipSuccess) { hipFree(e.ptr); return nullptr; }
  g_pixtabs.push_back(e);
  *bytes = e.bytes;
  return e.ptr;
}
int wgrad_v3_fill(const dsl_wgrad_desc* d, WgK& k, long long px) {
  unsigned tb = 0;
  k.pixtab = wgrad_pixtab(d, &tb);
  DSL_CHECK(k.pixtab != nullptr, "dsl_conv2d_wgrad: could not build the pixel descriptor table");
  k.pixtab_bytes = tb;
  k.ybytes = (unsigned)(px * d->cy * 2);
  return 0;
}
size_t wgrad_v3_lds(int cfg) {
  const int bcos[5] = {0, 256, 256, 128, 128}, bcis[5] = {0, 256, 128, 256, 128};
  return (size_t)wgrad_v3_nst(cfg) * kWgV3KS * 2 * (bcos[cfg] + bcis[cfg]) + (size_t)kWgV3DR * kWgV3KS * 8;
}
}  // namespace

// wgrad tile configurations: 0 = v1 (BCO 128|64 x 128, register staged), 1 = 256x256, 2 = 256co x 128ci,
// 3 = 128co x 256ci, 4 = 128x128 (v2)
static int wgrad_pick(const dsl_wgrad_desc* d) {
  const int force = d->splits < 0 ? -d->splits : 0;       // test hook: splits = -(cfg+1) forces a config
  if (force) return force - 1;
  if (d->cy % 128) return d->cs % 256 == 0 ? 3 : 4;      // cy = 64 (mod 128): the 128-cout tiles, upper half reads zeros
  if (d->cy % 256 == 0 && d->cs % 256 == 0) return 1;
  if (d->cy % 256 == 0) return 2;
  if (d->cs % 256 == 0) return 3;
  return 4;
}
static int wgrad_geometry(const dsl_wgrad_desc* d, int* ktiles, int* tiles, int* bco) {
  long long px = 0;
  for (int s = 0; s < d->nseg; ++s) px += (long long)d->n * d->gh[s] * d->gw[s];
  *ktiles = (int)((px + 63) / 64);
  const int cfg = wgrad_pick(d);
  const int bcos[5] = {(d->cy % 128 == 0) ? 128 : 64, 256, 256, 128, 128};
  const int bcis[5] = {128, 256, 128, 256, 128};
  *bco = bcos[cfg];
  *tiles = ((d->cy + *bco - 1) / *bco) * (d->kh * d->kw * d->cs / bcis[cfg]);
  return cfg;
}

static int wgrad_splits_for(const dsl_wgrad_desc* d, int count) {
  int ktiles, tiles, bco;
  const int cfg = wgrad_geometry(d, &ktiles, &tiles, &bco);
  tiles *= count;
  const int max_by_k = ktiles / 4 > 0 ? ktiles / 4 : 1;    // at least 4 K stages per split
  int splits;
  if (cfg == 0) {
    splits = (768 + tiles - 1) / tiles;                    // v1: 2-3 small workgroups per CU
  } else {
    const int per_cu = cfg == 4 ? 2 : 1;                   // 128x128 tiles: two workgroups per CU
    // one full round, never a nearly-empty second one.  (Accumulating the split partials with XCD-local L2 float
    // atomics instead of writing them out was measured: 117 vs 85 us on the head shape - L2 atomics retire about
    // two lanes per clock per channel.)
    // DSL_WGRAD_SLOTS < 256 leaves CUs free: the weight gradients run on the side stream under the caller's chain of
    // small convolutions, and a full round of 128 KB-LDS workgroups that live for 100-250 us would leave those
    // kernels only the handful of CUs the round did not cover
    // (measured, bench.py N = 2: 256 -> 305, 224 -> 306, 192 -> 309, 160 -> 313, 128 -> 310 img/s)
    static const int slots_env = [] { const char* e = getenv("DSL_WGRAD_SLOTS"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 128; }();
    const int slots = d->slots > 0 ? d->slots : slots_env;
    splits = slots * per_cu / tiles;
  }
  static const int forced_splits = [] { const char* e = getenv("DSL_WGRAD_SPLITS"); return e ? atoi(e) : 0; }();   // tuning knob
  if (forced_splits > 0 && cfg != 0) splits = forced_splits;
  if (splits > max_by_k) splits = max_by_k;
  if (splits < 1) splits = 1;
  if (splits > 256) splits = 256;
  {
    // No EMPTY split (round 4): split i covers stages [i * tps, (i + 1) * tps), tps = ceil(stages / splits); with 9 or more splits
    // and few stages the last ones start past the end - (splits - 1) * tps >= stages, e.g. 129 stages in 16 splits of 9 - their
    // workgroups return without writing their partial tile and the reduce pass adds whatever the scratch buffer held.  The comment
    // "cannot happen with the host's split factors" in the kernels was wrong for this corner; the planner of the multi launches
    // normalises the same way (plan_norm_splits).
    long long px = 0;
    for (int s = 0; s < d->nseg; ++s) px += (long long)d->n * d->gh[s] * d->gw[s];
    const int ks = wgrad_v3_ok(d, cfg) ? kWgV3KS : 64;
    const int stages = (int)((px + ks - 1) / ks);
    const int tps = (stages + splits - 1) / splits;
    splits = (stages + tps - 1) / tps;
  }
  return splits;
}

extern "C" int dsl_wgrad_splits(const dsl_wgrad_desc* d) { return wgrad_splits_for(d, 1); }

static size_t wgrad_cy_pad(const dsl_wgrad_desc* d) {       // rows of one partial tile set in the workspace
  int ktiles, tiles, bco;
  wgrad_geometry(d, &ktiles, &tiles, &bco);
  return (size_t)(d->cy + bco - 1) / bco * bco;
}

extern "C" size_t dsl_wgrad_workspace_bytes(const dsl_wgrad_desc* d) {
  const int splits = d->splits > 0 ? d->splits : dsl_wgrad_splits(d);
  return (size_t)splits * wgrad_cy_pad(d) * ((size_t)d->kh * d->kw * d->cs + 1) * sizeof(float);    // + one row of column sums
}

extern "C" size_t dsl_wgrad_group_workspace_bytes(const dsl_wgrad_desc* descs, int count) {
  if (!descs || count < 1) return 0;
  if (count == 1) return dsl_wgrad_workspace_bytes(descs);
  return (size_t)wgrad_splits_for(descs, count) * count * wgrad_cy_pad(descs) * ((size_t)descs->kh * descs->kw * descs->cs + 1) * sizeof(float);
}

extern "C" int dsl_colsum(const void* x, float* out, long rows, int c, int ld, void* stream);
int dsl_colsum_acc(const void* x, float* out, long rows, int c, int ld, void* stream);   // no memset: out += column sums

static bool wgrad_same_geometry(const dsl_wgrad_desc* a, const dsl_wgrad_desc* b) {
  if (a->ldx != b->ldx || a->shared != b->shared) return false;
  if (a->nseg != b->nseg || a->n != b->n || a->cs != b->cs || a->cy != b->cy || a->cd != b->cd || a->kh != b->kh ||
      a->kw != b->kw || a->stride != b->stride || a->pad != b->pad)
    return false;
  for (int s = 0; s < a->nseg; ++s)
    if (a->gh[s] != b->gh[s] || a->gw[s] != b->gw[s] || a->sh[s] != b->sh[s] || a->sw[s] != b->sw[s]) return false;
  return true;
}

// `count` convolutions of one geometry as one launch (count == 1: the plain weight gradient)
static int wgrad_launch(const dsl_wgrad_desc* descs, int count, void* stream) {
  const dsl_wgrad_desc* d = descs;
  DSL_CHECK(d != nullptr && count >= 1 && count <= DSL_MAX_GROUP, "dsl_conv2d_wgrad: bad group (count=%d)", count);
  DSL_CHECK(d->nseg >= 1 && d->nseg <= DSL_MAX_SEG, "dsl_conv2d_wgrad: nseg=%d", d->nseg);
  DSL_CHECK(d->cs % 128 == 0, "dsl_conv2d_wgrad: Cin=%d must be a multiple of 128", d->cs);
  DSL_CHECK(d->cy % 64 == 0 && d->cd <= d->cy, "dsl_conv2d_wgrad: bad cy=%d cd=%d", d->cy, d->cd);
  for (int g = 0; g < count; ++g) {
    DSL_CHECK(descs[g].dy && descs[g].x && descs[g].dw, "dsl_conv2d_wgrad: null pointer (member %d)", g);
    DSL_CHECK(wgrad_same_geometry(d, &descs[g]), "dsl_conv2d_wgrad_group: member %d has a different geometry", g);
  }
  DSL_CHECK(d->workspace, "dsl_conv2d_wgrad: null workspace");
  int ktiles, tiles, bco;
  const int cfg = wgrad_geometry(d, &ktiles, &tiles, &bco);
  if (cfg == 0 && count > 1) {          // the register-staged kernel has no group form: run the members one by one
    for (int g = 0; g < count; ++g) {
      dsl_wgrad_desc t = descs[g];
      t.workspace = d->workspace;
      t.workspace_bytes = d->workspace_bytes;
      t.splits = 0;
      const int rc = wgrad_launch(&t, 1, stream);
      if (rc) return rc;
    }
    return 0;
  }
  const int splits = count == 1 ? (d->splits > 0 ? d->splits : dsl_wgrad_splits(d)) : wgrad_splits_for(d, count);
  const int cyp = (int)wgrad_cy_pad(d);
  const size_t need = (size_t)splits * count * cyp * ((size_t)d->kh * d->kw * d->cs + 1) * sizeof(float);
  DSL_CHECK(d->workspace_bytes >= need, "dsl_conv2d_wgrad: workspace too small (%zu < %zu)", d->workspace_bytes, need);
  WgK k;
  memset(&k, 0, sizeof(k));
  k.nseg = d->nseg; k.n = d->n;
  int px = 0;
  long long xo = 0;
  for (int s = 0; s < d->nseg; ++s) {
    k.gh[s] = d->gh[s]; k.gw[s] = d->gw[s]; k.sh[s] = d->sh[s]; k.sw[s] = d->sw[s];
    k.pxstart[s] = px;
    k.xoff[s] = xo;
    k.dhw[s] = make_fastdiv((uint32_t)(d->gh[s] * d->gw[s]));
    k.dwd[s] = make_fastdiv((uint32_t)d->gw[s]);
    px += d->n * d->gh[s] * d->gw[s];
    xo += (long long)d->n * d->sh[s] * d->sw[s];
  }
  DSL_CHECK(px < (1 << 20), "dsl_conv2d_wgrad: %d pixels exceed the 2^20 fast-division range", px);
  const int ldx = d->ldx > 0 ? d->ldx : d->cs;
  DSL_CHECK(ldx >= d->cs && ldx % 8 == 0, "dsl_conv2d_wgrad: ldx=%d must be >= cs=%d and a multiple of 8", ldx, d->cs);
  DSL_CHECK(xo * ldx < (1LL << 31), "dsl_conv2d_wgrad: X has more than 2^31 elements");
  k.pxstart[d->nseg] = px;
  k.totpx = px;
  k.cs = d->cs; k.cy = d->cy; k.kh = d->kh; k.kw = d->kw; k.stride = d->stride; k.pad = d->pad;
  k.ktiles = ktiles;
  k.tiles_per_split = (ktiles + splits - 1) / splits;
  k.ctiles_per_tap = d->cs / 128;
  k.krow = (long long)d->kh * d->kw * d->cs;
  k.dy = (const uint16_t*)d->dy; k.x = (const uint16_t*)d->x; k.ws = (float*)d->workspace;
  k.group = count;
  k.ldx = ldx;
  k.cyp = cyp;
  for (int g = 0; g < DSL_MAX_GROUP; ++g) {
    k.dyv[g] = (const uint16_t*)descs[g < count ? g : 0].dy;
    k.xv[g] = (const uint16_t*)descs[g < count ? g : 0].x;
  }
  hipStream_t st = (hipStream_t)stream;
  // weight gradient: dY and X read once, dW written once (fp32)
  const int prof = dsl_prof_active()
                       ? dsl_prof_begin(3, 2.0 * count * px * (double)d->cd * d->kh * d->kw * d->cs, st,
                                        count * ((double)px * d->cd * 2.0 + (double)xo * d->cs * 2.0 + (double)d->cd * d->kh * d->kw * d->cs * 4.0))
                       : -1;
  if (cfg >= 1) {
    const int bcis[5] = {128, 256, 128, 256, 128};
    const int bci = bcis[cfg];
    DSL_CHECK(cyp % bco == 0 && d->cs % bci == 0, "dsl_conv2d_wgrad: tile config %d does not divide cy=%d / cs=%d", cfg, d->cy, d->cs);
    k.gx = cyp / bco;
    k.gy = d->kh * d->kw * d->cs / bci;
    k.splits = splits;
    k.dbws = (float*)d->workspace + (size_t)splits * count * cyp * k.krow;      // behind the dW partials
    for (int g = 0; g < count; ++g)
      if (descs[g].db) k.dbmask |= 1 << g;
#ifdef DSL_ABLATE_BUILD
    { const char* e = getenv("DSL_ABLATE"); k.dbg = e ? atoi(e) : 0; }
#endif
    k.chunk = (k.gx * k.gy * count * splits + 7) / 8;
    dim3 grid2(k.chunk * 8);
    const bool v3 = wgrad_v3_ok(d, cfg);
    const int kss[5] = {64, 64, 64, 64, 64}, nsts[5] = {2, 2, 3, 3, 2};
    const int ks = v3 ? kWgV3KS : kss[cfg];
    // the stage length of this tile configuration defines the K-tile unit
    k.ktiles = (px + ks - 1) / ks;
    k.tiles_per_split = (k.ktiles + splits - 1) / splits;
    if (v3)
      if (int rc = wgrad_v3_fill(d, k, px)) return rc;
    const size_t lds2 = v3 ? wgrad_v3_lds(cfg) : (size_t)nsts[cfg] * ks * 2 * (bco + bci);
#define LAUNCHW(KERNEL, A, B, C_, D, KS_, S_)                                                                         \
  do {                                                                                                               \
    static bool a_ = false;                                                                                          \
    if (!a_) {                                                                                                       \
      hipFuncSetAttribute((const void*)KERNEL<A, B, C_, D, KS_, S_>,                                                 \
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);                                    \
      a_ = true;                                                                                                     \
    }                                                                                                                \
    hipLaunchKernelGGL((KERNEL<A, B, C_, D, KS_, S_>), grid2, dim3(64 * C_ * D), lds2, st, k);                        \
  } while (0)
    if (v3) {
      if (wgrad_persist()) {
        const int cap = ((d->slots > 0 ? d->slots : wgrad_slots()) + 7) / 8 * 8;
        if ((int)grid2.x > cap) grid2.x = cap;
      }
      switch (cfg) {
        case 1: LAUNCHW(wgrad_pipe_kernel, 256, 256, 2, 4, 32, 4); break;
        case 2: LAUNCHW(wgrad_pipe_kernel, 256, 128, 4, 2, 32, 5); break;
        default: LAUNCHW(wgrad_pipe_kernel, 128, 256, 2, 4, 32, 5); break;
      }
    } else {
      switch (cfg) {
        case 1: LAUNCHW(wgrad_glds_kernel, 256, 256, 2, 4, 64, 2); break;
        case 2: LAUNCHW(wgrad_glds_kernel, 256, 128, 4, 2, 64, 3); break;
        case 3: LAUNCHW(wgrad_glds_kernel, 128, 256, 2, 4, 64, 3); break;
        default: LAUNCHW(wgrad_glds_kernel, 128, 128, 2, 2, 64, 2); break;
      }
    }
#undef LAUNCHW
  } else if (bco == 128) {
    dim3 grid(d->cy / bco, d->kh * d->kw * d->cs / 128, splits);
    const size_t lds = 2 * (size_t)(64 * bco * 2 + 64 * 256);
    static bool a = false;
    if (!a) { hipFuncSetAttribute((const void*)wgrad_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); a = true; }
    hipLaunchKernelGGL((wgrad_kernel<128>), grid, dim3(256), lds, st, k);
  } else {
    dim3 grid(d->cy / bco, d->kh * d->kw * d->cs / 128, splits);
    const size_t lds = 2 * (size_t)(64 * bco * 2 + 64 * 256);
    static bool a = false;
    if (!a) { hipFuncSetAttribute((const void*)wgrad_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); a = true; }
    hipLaunchKernelGGL((wgrad_kernel<64>), grid, dim3(256), lds, st, k);
  }
  dsl_prof_end(prof, st);
  DSL_LAUNCH_CHECK("wgrad_kernel");
  const long long total4 = (long long)d->cd * k.krow / 4;
  int rb = (int)((total4 + 255) / 256);
  if (rb > 4096 / count) rb = 4096 / count;
  RedK r;
  for (int g = 0; g < DSL_MAX_GROUP; ++g) {
    r.dw[g] = descs[g < count ? g : 0].dw;
    r.scale[g] = descs[g < count ? g : 0].scale;
    r.db[g] = g < count ? descs[g].db : nullptr;
  }
  r.dbws = cfg >= 1 ? k.dbws : nullptr;
  if (d->shared && count > 1) {
    // the members are applications of ONE convolution (weights shared along a recurrence): their partial tiles are just
    // more splits of the same dW - [split][member] pairs are contiguous in the workspace
    for (int g = 1; g < count; ++g)
      DSL_CHECK(descs[g].dw == d->dw && descs[g].scale == d->scale && !descs[g].db, "dsl_conv2d_wgrad_group: shared members must share dw / scale and have no db");
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(rb * count, 1), dim3(256), 0, st, (const float*)d->workspace, r, splits * count, 1,
                       cyp, d->cd, k.krow);
    DSL_LAUNCH_CHECK("wgrad_reduce_kernel");
    return 0;
  }
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(rb, count), dim3(256), 0, st, (const float*)d->workspace, r, splits, count,
                     cyp, d->cd, k.krow);
  DSL_LAUNCH_CHECK("wgrad_reduce_kernel");
  if (cfg == 0)              // register-staged kernel: separate column-sum pass (db was cleared by the reduce kernel above)
    for (int g = 0; g < count; ++g)
      if (descs[g].db) {
        const int rc = dsl_colsum_acc(descs[g].dy, descs[g].db, (long)px, d->cd, d->cy, stream);
        if (rc) return rc;
      }
  return 0;
}

// ------------------------------------------------------------------------------------------------
// multi launch: the weight gradients of several geometries (one tile configuration) as ONE grid + ONE reduce grid
// ------------------------------------------------------------------------------------------------
namespace {
constexpr int kMaxRed = 128, kMaxColsum = 64, kSchedMax = 8192;
struct ColsumItem { const void* x; float* out; long long rows; int c, ld, clear; };
struct WgMultiTable {
  int magic, cfg, nsub, total_blocks;
  WgMultiHdr hdr;
  int n_red, red_blocks, n_colsum, v3;
  double flops, bytes;
  ColsumItem colsum[kMaxColsum];
  WgK k[kMaxMulti];
  RedEnt red[kMaxRed];
  // scheduled persistent launch (wgrad_plan): sched_grid workgroups, sched_rounds blocks each at most; 0 rounds = stride form
  int sched_grid, sched_rounds;
  int plan_makespan, plan_items;      // (stages incl. the per-item overhead; valid blocks) - what dsl_wgrad_multi_info reports
  short sched[kSchedMax];
};
constexpr int kMultiMagic = 0x574d5431;

int wgrad_fill_k(const dsl_wgrad_desc* descs, int count, int splits, int cfg, bool v3, WgK& k, long long* px_out, long long* xo_out) {
  const dsl_wgrad_desc* d = descs;
  memset(&k, 0, sizeof(k));
  k.nseg = d->nseg; k.n = d->n;
  int px = 0;
  long long xo = 0;
  for (int s = 0; s < d->nseg; ++s) {
    k.gh[s] = d->gh[s]; k.gw[s] = d->gw[s]; k.sh[s] = d->sh[s]; k.sw[s] = d->sw[s];
    k.pxstart[s] = px;
    k.xoff[s] = xo;
    k.dhw[s] = make_fastdiv((uint32_t)(d->gh[s] * d->gw[s]));
    k.dwd[s] = make_fastdiv((uint32_t)d->gw[s]);
    px += d->n * d->gh[s] * d->gw[s];
    xo += (long long)d->n * d->sh[s] * d->sw[s];
  }
  DSL_CHECK(px < (1 << 20), "dsl_conv2d_wgrad: %d pixels exceed the 2^20 fast-division range", px);
  const int ldx = d->ldx > 0 ? d->ldx : d->cs;
  DSL_CHECK(ldx >= d->cs && ldx % 8 == 0, "dsl_conv2d_wgrad: ldx=%d must be >= cs=%d and a multiple of 8", ldx, d->cs);
  DSL_CHECK(xo * ldx < (1LL << 31), "dsl_conv2d_wgrad: X has more than 2^31 elements");
  k.pxstart[d->nseg] = px;
  k.totpx = px;
  k.cs = d->cs; k.cy = d->cy; k.kh = d->kh; k.kw = d->kw; k.stride = d->stride; k.pad = d->pad;
  k.ctiles_per_tap = d->cs / 128;
  k.krow = (long long)d->kh * d->kw * d->cs;
  k.group = count;
  k.ldx = ldx;
  k.cyp = (int)wgrad_cy_pad(d);
  k.cd = d->cd;
  for (int g = 0; g < DSL_MAX_GROUP; ++g) {
    const dsl_wgrad_desc& m = descs[g < count ? g : 0];
    k.dyv[g] = (const uint16_t*)m.dy;
    k.xv[g] = (const uint16_t*)m.x;
    k.dwv[g] = m.dw;
    k.scalev[g] = m.scale;
    k.dbv[g] = m.db;
    if (g < count && m.db) k.dbmask |= 1 << g;
  }
  k.dy = k.dyv[0]; k.x = k.xv[0];
  const int bcos[5] = {0, 256, 256, 128, 128}, bcis[5] = {0, 256, 128, 256, 128};
  DSL_CHECK(k.cyp % bcos[cfg] == 0 && d->cs % bcis[cfg] == 0, "dsl_conv2d_wgrad: tile config %d does not divide cy=%d / cs=%d", cfg, d->cy, d->cs);
  k.gx = k.cyp / bcos[cfg];
  k.gy = d->kh * d->kw * d->cs / bcis[cfg];
  k.splits = splits;
  k.chunk = (k.gx * k.gy * count * splits + 7) / 8;
  const int ks = v3 ? kWgV3KS : 64;
  k.ktiles = (px + ks - 1) / ks;
  k.tiles_per_split = (k.ktiles + splits - 1) / splits;
  if (v3)
    if (int rc = wgrad_v3_fill(d, k, px)) return rc;
  *px_out = px;
  *xo_out = xo;
  return 0;
}

int wgrad_slots() {
  static const int slots = [] { const char* e = getenv("DSL_WGRAD_SLOTS"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 128; }();
  return slots;
}

// ---- launch planner of the multi launches (round 4) ----------------------------------------------------------------------
// A multi launch is a list of work items (one output tile x one K split) of very different lengths - the FPN's run from 3 to
// 525 64-pixel K tiles - on a persistent grid of <= `cap` workgroups.  Round 3 chose the split factors from one target length
// (total / slots, rounded per sub-launch) and let workgroup b walk the items b, b + G, b + 2G ...: the predictors came out as
// 144 equal items on 128 workgroups (two rounds for 16 of them: 2 x the ideal time), the FPN as 173 items whose second round
// paired the longest with the middle ones (91 units against an ideal 64), layer3 as 2 x 108 half-length items + a reduce pass
// where 108 whole ones fit one round without any partial tile.  The planner simulates what the grid will do: for every
// candidate vector of split factors it assigns the items to workgroups (longest first, each to the least loaded workgroup of
// the item's XCD class, so a block keeps the L2 its neighbours use), takes the longest workgroup's sum (+ a fixed cost per
// item: ring fill and the tile's stores) and adds the reduce pass the split partials would need; the cheapest vector wins and
// its assignment becomes the launch's schedule table.  Results do not depend on the schedule: an item computes the same tile
// from the same stages whoever runs it, and the reduce pass folds the splits in split order.
struct PlanSub { int stages, tiles, max_sp; long long tile_elems; };      // K stages of the kernel's unit, output tiles (all members), elements per tile set
int wgrad_plan_mode() {
  static const int v = [] { const char* e = getenv("DSL_WGRAD_PLAN"); return e ? atoi(e) : 1; }();
  return v;
}
inline int plan_norm_splits(int stages, int sp) {      // no empty split: sp -> ceil(stages / ceil(stages / sp))
  if (sp < 1) sp = 1;
  const int tps = (stages + sp - 1) / sp;
  return (stages + tps - 1) / tps;
}
// LPT assignment of the valid blocks of the launch (sub-launches in table order) to G workgroups.  Returns the makespan in
// stages (incl. `ovh` per item); sched (may be null) gets G * rounds entries.
long long plan_simulate(const PlanSub* subs, const int* splits, int nsub, int G, int ovh, short* sched, int sched_cap, int* rounds_out,
                        int* items_out) {
  struct It { int cost, vb; };
  std::vector<It> cls[8];
  int base = 0, items = 0;
  for (int i = 0; i < nsub; ++i) {
    const int witems = subs[i].tiles * splits[i];
    const int chunk = (witems + 7) / 8;
    const int tps = (subs[i].stages + splits[i] - 1) / splits[i];
    for (int bid = 0; bid < chunk * 8; ++bid) {
      const int xcd = bid & 7, jj = bid >> 3, w = xcd * chunk + jj;
      if (jj >= chunk || w >= witems) continue;
      const int sp = w / subs[i].tiles;
      const int k0 = sp * tps, k1 = std::min(k0 + tps, subs[i].stages);
      if (k1 <= k0) continue;
      cls[xcd].push_back({ovh + (k1 - k0), base + bid});
      ++items;
    }
    base += chunk * 8;
  }
  const int per = G / 8;
  long long makespan = 0;
  int rounds = 0;
  std::vector<long long> load(G, 0);
  std::vector<std::vector<int>> mine(G);
  for (int x = 0; x < 8; ++x) {
    std::stable_sort(cls[x].begin(), cls[x].end(), [](const It& a, const It& b) { return a.cost > b.cost; });
    for (const It& it : cls[x]) {
      int best = x;
      for (int j = 1; j < per; ++j)
        if (load[x + 8 * j] < load[best]) best = x + 8 * j;
      load[best] += it.cost;
      mine[best].push_back(it.vb);
    }
  }
  for (int b = 0; b < G; ++b) {
    makespan = std::max(makespan, load[b]);
    rounds = std::max(rounds, (int)mine[b].size());
  }
  if (sched) {
    if ((long long)rounds * G > sched_cap) { rounds = 0; }      // does not fit the table: the caller falls back to the stride form
    else {
      for (int i = 0; i < rounds * G; ++i) sched[i] = -1;
      for (int b = 0; b < G; ++b)
        for (size_t r = 0; r < mine[b].size(); ++r) sched[r * G + b] = (short)mine[b][r];
    }
  }
  if (rounds_out) *rounds_out = rounds;
  if (items_out) *items_out = items;
  return makespan;
}
// microseconds per stage / fixed stages per item of a tile configuration (fits of round 3's traces: the towers' direct tiles run
// 1 400 32-pixel stages in 868 us alone; the predictors' 128 x 256 items 198 stages in ~85 us)
inline double plan_stage_us(int cfg) { return cfg == 1 ? 0.62 : 0.43; }
inline int plan_ovh(int cfg) { return cfg == 1 ? 8 : 8; }
struct PlanOut { int splits[kMaxMulti]; int grid, makespan, items; double us; };
void wgrad_plan(const PlanSub* subs, int nsub, int cfg, int cap, PlanOut* out) {
  const int ovh = plan_ovh(cfg);
  long long total = 0;
  int smax = 1;
  for (int i = 0; i < nsub; ++i) { total += (long long)subs[i].stages * subs[i].tiles; smax = std::max(smax, subs[i].stages); }
  // candidate target lengths: every value ceil(stages_i / j) that changes some sub-launch's split factor, within a window around
  // the balanced length, plus "no split at all"
  const long long bal = std::max<long long>(8, (total + cap - 1) / cap);
  std::vector<int> cand;
  cand.push_back(smax);
  for (int i = 0; i < nsub; ++i)
    for (int j = 1; j <= subs[i].max_sp; ++j) {
      const int l = (subs[i].stages + j - 1) / j;
      if (l >= bal / 3 && l <= bal * 4) cand.push_back(l);
    }
  std::sort(cand.begin(), cand.end());
  cand.erase(std::unique(cand.begin(), cand.end()), cand.end());
  double best = 1e30;
  std::vector<std::vector<int>> seen;
  for (int l : cand) {
    std::vector<int> sp(nsub);
    long long items = 0;
    for (int i = 0; i < nsub; ++i) {
      int v = (subs[i].stages + l - 1) / l;
      if (v > subs[i].max_sp) v = subs[i].max_sp;
      sp[i] = plan_norm_splits(subs[i].stages, v);
      items += (long long)sp[i] * subs[i].tiles;
    }
    if (std::find(seen.begin(), seen.end(), sp) != seen.end()) continue;
    seen.push_back(sp);
    // the smallest grid that reaches the best makespan (an XCD class may hold more items than items / 8)
    int G = (int)std::min<long long>(cap, (items + 7) / 8 * 8);
    if (G < 8) G = 8;
    int n_items = 0;
    long long ms = plan_simulate(subs, sp.data(), nsub, cap, ovh, nullptr, 0, nullptr, &n_items);
    {
      int g = G;
      for (; g < cap; g += 8)
        if (plan_simulate(subs, sp.data(), nsub, g, ovh, nullptr, 0, nullptr, nullptr) <= ms) break;
      G = g;
    }
    double red_bytes = 0;
    for (int i = 0; i < nsub; ++i)
      if (sp[i] > 1) red_bytes += (double)(sp[i] + 1) * subs[i].tile_elems * 4.0;      // partials written, read back, dW written
    const double us = ms * plan_stage_us(cfg) + (red_bytes > 0 ? 6.0 + red_bytes / 3.0e6 : 0.0);
    if (us < best) {
      best = us;
      for (int i = 0; i < nsub; ++i) out->splits[i] = sp[i];
      out->grid = G; out->makespan = (int)ms; out->items = n_items; out->us = us;
    }
  }
}
long long wgrad_px(const dsl_wgrad_desc* d) {
  long long px = 0;
  for (int s = 0; s < d->nseg; ++s) px += (long long)d->n * d->gh[s] * d->gw[s];
  return px;
}
bool wgrad_multi_v3(const dsl_wgrad_desc* descs, const int* counts, int nsub, int cfg) {
  bool v3 = true;
  int off = 0;
  for (int s = 0; s < nsub; ++s) { v3 = v3 && wgrad_v3_ok(&descs[off], cfg); off += counts[s]; }
  return v3;
}
// workgroup budget of a multi launch: the library's (DSL_WGRAD_SLOTS, default 128) unless a descriptor asks for its own
// (dsl_wgrad_desc.slots > 0: the launches at the very end of a backward pass, with nothing left to run beside them, take more)
int wgrad_multi_cap(const dsl_wgrad_desc* descs, const int* counts, int nsub) {
  int n = 0, cap = 0;
  for (int s = 0; s < nsub; ++s) n += counts[s];
  for (int i = 0; i < n; ++i) cap = std::max(cap, descs[i].slots);
  if (cap <= 0) cap = wgrad_slots();
  return (std::min(cap, 256) + 7) / 8 * 8;
}

// the planner's view of a launch's sub-launches (in the caller's order)
void wgrad_plan_subs(const dsl_wgrad_desc* descs, const int* counts, int nsub, int ks, PlanSub* subs) {
  int off = 0;
  for (int s = 0; s < nsub; ++s) {
    int ktiles, tiles, bco;
    wgrad_geometry(&descs[off], &ktiles, &tiles, &bco);
    const long long px = wgrad_px(&descs[off]);
    subs[s].stages = (int)((px + ks - 1) / ks);
    subs[s].tiles = tiles * counts[s];
    subs[s].max_sp = std::max(1, subs[s].stages / (256 / ks));          // at least 256 pixels of K per split (round 3's rule)
    subs[s].tile_elems = (long long)counts[s] * (long long)wgrad_cy_pad(&descs[off]) * ((long long)descs[off].kh * descs[off].kw * descs[off].cs);
    off += counts[s];
  }
}

// split factors of a multi launch: every workgroup gets at most ~1/slots of the launch's K-tile iterations
int wgrad_multi_splits(const dsl_wgrad_desc* descs, const int* counts, int nsub, int* splits) {
  {
    const int cfg0 = wgrad_pick(descs);
    if (wgrad_plan_mode() && cfg0 >= 1 && cfg0 <= 3 && wgrad_persist() && wgrad_multi_v3(descs, counts, nsub, cfg0)) {
      PlanSub subs[kMaxMulti];
      wgrad_plan_subs(descs, counts, nsub, kWgV3KS, subs);
      PlanOut po;
      wgrad_plan(subs, nsub, cfg0, wgrad_multi_cap(descs, counts, nsub), &po);
      for (int s = 0; s < nsub; ++s) splits[s] = po.splits[s];
      return 0;
    }
  }
  long long total = 0;
  int off = 0;
  for (int s = 0; s < nsub; ++s) {
    int ktiles, tiles, bco;
    wgrad_geometry(&descs[off], &ktiles, &tiles, &bco);
    total += (long long)ktiles * tiles * counts[s];
    off += counts[s];
  }
  const int cfg = wgrad_pick(descs);
  const int slots = wgrad_multi_cap(descs, counts, nsub) * (cfg == 4 ? 2 : 1);
  long long lmax = (total + slots - 1) / slots;
  if (lmax < 4) lmax = 4;
  off = 0;
  for (int s = 0; s < nsub; ++s) {
    int ktiles, tiles, bco;
    wgrad_geometry(&descs[off], &ktiles, &tiles, &bco);
    int sp = (int)((ktiles + lmax - 1) / lmax);
    const int max_by_k = ktiles / 4 > 0 ? ktiles / 4 : 1;
    if (sp > max_by_k) sp = max_by_k;
    if (sp < 1) sp = 1;
    splits[s] = sp;
    off += counts[s];
  }
  return 0;
}

int wgrad_multi_check(const dsl_wgrad_desc* descs, const int* counts, int nsub) {
  DSL_CHECK(descs && counts && nsub >= 1 && nsub <= kMaxMulti, "dsl_wgrad_multi: bad sub-launch list (n=%d)", nsub);
  const int cfg = wgrad_pick(descs);
  DSL_CHECK(cfg >= 1 && cfg <= 4, "dsl_wgrad_multi: tile configuration %d has no multi form", cfg);
  int off = 0;
  for (int s = 0; s < nsub; ++s) {
    DSL_CHECK(counts[s] >= 1 && counts[s] <= DSL_MAX_GROUP, "dsl_wgrad_multi: sub-launch %d has %d members", s, counts[s]);
    const dsl_wgrad_desc* d = &descs[off];
    DSL_CHECK(d->nseg >= 1 && d->nseg <= DSL_MAX_SEG && d->cs % 128 == 0 && d->cy % 64 == 0 && d->cd <= d->cy,
              "dsl_wgrad_multi: bad geometry in sub-launch %d", s);
    DSL_CHECK(wgrad_pick(d) == cfg, "dsl_wgrad_multi: sub-launch %d needs tile configuration %d, the launch uses %d", s, wgrad_pick(d), cfg);
    for (int g = 0; g < counts[s]; ++g) {
      DSL_CHECK(d[g].dy && d[g].x && d[g].dw, "dsl_wgrad_multi: null pointer (sub-launch %d member %d)", s, g);
      DSL_CHECK(!d[g].shared, "dsl_wgrad_multi: shared-weight groups use dsl_conv2d_wgrad_group");
      DSL_CHECK(wgrad_same_geometry(d, &d[g]), "dsl_wgrad_multi: sub-launch %d member %d has a different geometry", s, g);
    }
    off += counts[s];
  }
  return 0;
}
}  // namespace

extern "C" int dsl_wgrad_multi_config(const dsl_wgrad_desc* d) {
  DSL_CHECK(d != nullptr, "dsl_wgrad_multi_config: null descriptor");
  return d->shared ? 0 : wgrad_pick(d);
}

extern "C" size_t dsl_wgrad_multi_table_bytes(void) { return sizeof(WgMultiTable); }

extern "C" size_t dsl_wgrad_multi_workspace_bytes(const dsl_wgrad_desc* descs, const int* counts, int nsub) {
  if (wgrad_multi_check(descs, counts, nsub)) return 0;
  int splits[kMaxMulti];
  wgrad_multi_splits(descs, counts, nsub, splits);
  size_t need = 0;
  int off = 0;
  for (int s = 0; s < nsub; ++s) {
    const dsl_wgrad_desc* d = &descs[off];
    if (splits[s] > 1) need += (size_t)splits[s] * counts[s] * wgrad_cy_pad(d) * ((size_t)d->kh * d->kw * d->cs + 1) * sizeof(float);
    off += counts[s];
  }
  return need ? need : 16;
}

// Fills `table_host` (dsl_wgrad_multi_table_bytes()) for the sub-launches descs[0 .. sum(counts)) (sub-launch s = counts[s]
// consecutive same-geometry descriptors, all of one tile configuration, dsl_wgrad_multi_config).  The caller copies the
// bytes to device memory once and passes both copies to dsl_conv2d_wgrad_multi; the table stays valid while the
// descriptors' pointers and `workspace` do.
extern "C" int dsl_wgrad_multi_build(const dsl_wgrad_desc* descs, const int* counts, int nsub, void* workspace, size_t ws_bytes,
                                     void* table_host, size_t table_bytes) {
  if (int rc = wgrad_multi_check(descs, counts, nsub)) return rc;
  DSL_CHECK(table_host && table_bytes >= sizeof(WgMultiTable), "dsl_wgrad_multi_build: table buffer too small");
  DSL_CHECK(workspace && ws_bytes >= dsl_wgrad_multi_workspace_bytes(descs, counts, nsub), "dsl_wgrad_multi_build: workspace too small");
  WgMultiTable* t = (WgMultiTable*)table_host;
  memset(t, 0, sizeof(*t));
  t->magic = kMultiMagic;
  t->cfg = wgrad_pick(descs);
  t->nsub = nsub;
  int splits[kMaxMulti], first[kMaxMulti], order[kMaxMulti];
  wgrad_multi_splits(descs, counts, nsub, splits);
  {                                    // the pipelined kernel serves the launch only if it can serve every sub-launch
    bool v3 = true;
    int off = 0;
    for (int s = 0; s < nsub; ++s) { v3 = v3 && wgrad_v3_ok(&descs[off], t->cfg); off += counts[s]; }
    t->v3 = v3 ? 1 : 0;
  }
  long long per_wg[kMaxMulti];
  {
    int off = 0;
    for (int s = 0; s < nsub; ++s) {
      int ktiles, tiles, bco;
      wgrad_geometry(&descs[off], &ktiles, &tiles, &bco);
      per_wg[s] = (ktiles + splits[s] - 1) / splits[s];
      first[s] = off;
      order[s] = s;
      off += counts[s];
    }
  }
  for (int i = 1; i < nsub; ++i)            // longest workgroups first (stable insertion sort)
    for (int j = i; j > 0 && per_wg[order[j]] > per_wg[order[j - 1]]; --j) { const int tmp = order[j]; order[j] = order[j - 1]; order[j - 1] = tmp; }
  unsigned char* ws = (unsigned char*)workspace;
  int blocks = 0, red_blocks = 0;
  for (int i = 0; i < nsub; ++i) {
    const int s = order[i];
    const dsl_wgrad_desc* d = &descs[first[s]];
    WgK& k = t->k[i];
    long long px, xo;
    if (int rc = wgrad_fill_k(d, counts[s], splits[s], t->cfg, t->v3 != 0, k, &px, &xo)) return rc;
    k.direct = splits[s] == 1 ? 1 : 0;
    k.ws = (float*)ws;
    blocks += k.chunk * 8;
    t->hdr.wg_end[i] = blocks;
    t->flops += 2.0 * counts[s] * px * (double)d->cd * d->kh * d->kw * d->cs;
    t->bytes += counts[s] * ((double)px * d->cd * 2.0 + (double)xo * d->cs * 2.0 + (double)d->cd * d->kh * d->kw * d->cs * 4.0);
    const long long sub_elems = (long long)counts[s] * k.cyp * k.krow;
    k.dbws = (float*)ws + (size_t)splits[s] * sub_elems;       // behind this sub-launch's dW partials
    for (int g = 0; g < counts[s]; ++g) {
      if (!k.direct) {
        DSL_CHECK(t->n_red < kMaxRed, "dsl_wgrad_multi_build: more than %d reduce entries", kMaxRed);
        RedEnt& r = t->red[t->n_red++];
        r.ws = (const float*)ws + (long long)g * k.cyp * k.krow;
        r.dw = d[g].dw; r.scale = d[g].scale; r.db = d[g].db;
        r.dbws = k.dbws + (long long)g * k.cyp; r.dbstride = (long long)counts[s] * k.cyp;
        r.krow = k.krow; r.sstride = sub_elems; r.splits = splits[s]; r.cd = d->cd;
        const long long total4 = (long long)d->cd * k.krow / 4;
        int nb = (int)((total4 + 1023) / 1024);        // ~4 f32x4 per thread
        if (nb > 512) nb = 512;
        if (nb < 1) nb = 1;
        r.blk_start = red_blocks; r.nblk = nb;
        red_blocks += nb;
      }
    }
    if (!k.direct) ws += (size_t)splits[s] * (sub_elems + (long long)counts[s] * k.cyp) * sizeof(float);
  }
  t->hdr.nsub = nsub;
  t->total_blocks = blocks;
  t->red_blocks = red_blocks;
  if (wgrad_plan_mode() && t->v3 && t->cfg <= 3 && wgrad_persist()) {
    // the schedule of the persistent grid, for the sub-launches in TABLE order (that is the block numbering the kernel sees)
    PlanSub subs[kMaxMulti], tsubs[kMaxMulti];
    int tsplits[kMaxMulti];
    wgrad_plan_subs(descs, counts, nsub, kWgV3KS, subs);
    long long items = 0;
    for (int i = 0; i < nsub; ++i) { tsubs[i] = subs[order[i]]; tsplits[i] = splits[order[i]]; items += (long long)tsplits[i] * tsubs[i].tiles; }
    const int cap_ = wgrad_multi_cap(descs, counts, nsub);
    int G = (int)std::min<long long>(cap_, (items + 7) / 8 * 8);
    if (G < 8) G = 8;
    {
      const long long ms_cap = plan_simulate(tsubs, tsplits, nsub, cap_, plan_ovh(t->cfg), nullptr, 0, nullptr, nullptr);
      for (; G < cap_; G += 8)
        if (plan_simulate(tsubs, tsplits, nsub, G, plan_ovh(t->cfg), nullptr, 0, nullptr, nullptr) <= ms_cap) break;
    }
    int rounds = 0, n_items = 0;
    const long long ms = plan_simulate(tsubs, tsplits, nsub, G, plan_ovh(t->cfg), t->sched, kSchedMax, &rounds, &n_items);
    if (blocks < 32767 && rounds > 0) { t->sched_grid = G; t->sched_rounds = rounds; }
    t->plan_makespan = (int)ms; t->plan_items = n_items;
  }
  return 0;
}

// Planner probe (tests, tools; no device needed): sub-launch s has stages[s] K stages, tiles[s] output tiles (all members) and
// tile_elems[s] elements per tile set; returns the chosen split factors and info = {grid, makespan, items, makespan of round
// 3's rule under the stride walk, its items}
extern "C" int dsl_wgrad_plan_probe(const int* stages, const int* tiles, const long long* tile_elems, int nsub, int cfg, int cap,
                                    int* splits_out, int* info) {
  DSL_CHECK(stages && tiles && nsub >= 1 && nsub <= kMaxMulti && cfg >= 1 && cfg <= 3 && cap >= 8 && cap % 8 == 0, "dsl_wgrad_plan_probe: bad arguments");
  PlanSub subs[kMaxMulti];
  for (int s = 0; s < nsub; ++s) {
    subs[s].stages = stages[s]; subs[s].tiles = tiles[s]; subs[s].max_sp = std::max(1, stages[s] / 8);
    subs[s].tile_elems = tile_elems ? tile_elems[s] : 0;
  }
  PlanOut po;
  wgrad_plan(subs, nsub, cfg, cap, &po);
  for (int s = 0; s < nsub; ++s) splits_out[s] = po.splits[s];
  info[0] = po.grid; info[1] = po.makespan; info[2] = po.items;
  {   // self-check of the schedule table the launch would use: every valid block exactly once, no holes in a workgroup's list
    std::vector<short> sched(kSchedMax);
    int rounds = 0, n_items = 0, total_blocks = 0;
    plan_simulate(subs, po.splits, nsub, po.grid, plan_ovh(cfg), sched.data(), kSchedMax, &rounds, &n_items);
    for (int s = 0; s < nsub; ++s) total_blocks += (subs[s].tiles * po.splits[s] + 7) / 8 * 8;
    if (rounds > 0) {
      std::vector<int> hit(total_blocks, 0);
      int seen_items = 0;
      for (int b = 0; b < po.grid; ++b) {
        bool ended = false;
        for (int r = 0; r < rounds; ++r) {
          const int vb = sched[r * po.grid + b];
          if (vb < 0) { ended = true; continue; }
          DSL_CHECK(!ended && vb < total_blocks && (vb & 7) == (b & 7) && hit[vb]++ == 0, "dsl_wgrad_plan_probe: bad schedule entry (workgroup %d round %d block %d)", b, r, vb);
          ++seen_items;
        }
      }
      DSL_CHECK(seen_items == n_items && n_items == po.items, "dsl_wgrad_plan_probe: schedule holds %d of %d items", seen_items, n_items);
    }
  }
  // round 3: one target length, stride walk of the blocks in "longest per workgroup first" order
  long long total = 0;
  for (int s = 0; s < nsub; ++s) total += (long long)((stages[s] + 1) / 2) * tiles[s];
  long long lmax = std::max<long long>(4, (total + cap - 1) / cap);
  int osp[kMaxMulti], ord[kMaxMulti];
  for (int s = 0; s < nsub; ++s) {
    const int kt = (stages[s] + 1) / 2;
    int sp = (int)((kt + lmax - 1) / lmax);
    sp = std::max(1, std::min(sp, std::max(1, kt / 4)));
    osp[s] = sp; ord[s] = s;
  }
  for (int i = 1; i < nsub; ++i)
    for (int j = i; j > 0 && (stages[ord[j]] + osp[ord[j]] - 1) / osp[ord[j]] > (stages[ord[j - 1]] + osp[ord[j - 1]] - 1) / osp[ord[j - 1]]; --j) std::swap(ord[j], ord[j - 1]);
  std::vector<long long> load(cap, 0);
  int base = 0, oitems = 0;
  for (int i = 0; i < nsub; ++i) {
    const int s = ord[i], witems = tiles[s] * osp[s], chunk = (witems + 7) / 8, tps = (stages[s] + osp[s] - 1) / osp[s];
    for (int bid = 0; bid < chunk * 8; ++bid) {
      const int xcd = bid & 7, jj = bid >> 3, w = xcd * chunk + jj;
      if (jj >= chunk || w >= witems) continue;
      const int sp = w / tiles[s], k0 = sp * tps, k1 = std::min(k0 + tps, stages[s]);
      if (k1 > k0) { load[(base + bid) % cap] += plan_ovh(cfg) + k1 - k0; ++oitems; }
    }
    base += chunk * 8;
  }
  info[3] = (int)*std::max_element(load.begin(), load.end());
  info[4] = oitems;
  return 0;
}

// what a table holds (profiling tools): algorithmic flops / bytes of the launch, its workgroups, reduce workgroups, sub-launches
extern "C" int dsl_wgrad_multi_info(const void* table_host, double* flops, double* bytes, int* blocks, int* red_blocks, int* nsub) {
  const WgMultiTable* t = (const WgMultiTable*)table_host;
  DSL_CHECK(t && t->magic == kMultiMagic, "dsl_wgrad_multi_info: not a table of dsl_wgrad_multi_build");
  if (flops) *flops = t->flops;
  if (bytes) *bytes = t->bytes;
  if (blocks) *blocks = t->total_blocks;
  if (red_blocks) *red_blocks = t->red_blocks;
  if (nsub) *nsub = t->nsub;
  return 0;
}

extern "C" int dsl_conv2d_wgrad_multi(const void* table_host, const void* table_dev, void* stream) {
  const WgMultiTable* t = (const WgMultiTable*)table_host;
  DSL_CHECK(t && table_dev && t->magic == kMultiMagic, "dsl_conv2d_wgrad_multi: not a table of dsl_wgrad_multi_build");
  const WgK* ktab = (const WgK*)((const unsigned char*)table_dev + offsetof(WgMultiTable, k));
  const RedEnt* rtab = (const RedEnt*)((const unsigned char*)table_dev + offsetof(WgMultiTable, red));
  hipStream_t st = (hipStream_t)stream;
  const int prof = dsl_prof_active() ? dsl_prof_begin(3, t->flops, st, t->bytes) : -1;
  const int bcos[5] = {0, 256, 256, 128, 128}, bcis[5] = {0, 256, 128, 256, 128}, nsts[5] = {2, 2, 3, 3, 2};
  const size_t lds2 = t->v3 ? wgrad_v3_lds(t->cfg) : (size_t)nsts[t->cfg] * 64 * 2 * (bcos[t->cfg] + bcis[t->cfg]);
  const dim3 grid(t->total_blocks);
#define LAUNCHM(KERNEL, A, B, C_, D, KS_, S_)                                                                        \
  do {                                                                                                               \
    static bool a_ = false;                                                                                          \
    if (!a_) {                                                                                                       \
      hipFuncSetAttribute((const void*)KERNEL<A, B, C_, D, KS_, S_>,                                                 \
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);                                    \
      a_ = true;                                                                                                     \
    }                                                                                                                \
    hipLaunchKernelGGL((KERNEL<A, B, C_, D, KS_, S_>), grid, dim3(64 * C_ * D), lds2, st, t->hdr, ktab);              \
  } while (0)
  const int cap = (wgrad_slots() + 7) / 8 * 8;
  if (t->v3 && t->sched_rounds > 0) {
    const short* sched = (const short*)((const unsigned char*)table_dev + offsetof(WgMultiTable, sched));
    const dim3 sgrid(t->sched_grid);
#define LAUNCHS(A, B, C_, D, KS_, S_)                                                                                \
  do {                                                                                                               \
    static bool a_ = false;                                                                                          \
    if (!a_) {                                                                                                       \
      hipFuncSetAttribute((const void*)wgrad_pipe_multi_sched_kernel<A, B, C_, D, KS_, S_>,                          \
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);                                    \
      a_ = true;                                                                                                     \
    }                                                                                                                \
    hipLaunchKernelGGL((wgrad_pipe_multi_sched_kernel<A, B, C_, D, KS_, S_>), sgrid, dim3(64 * C_ * D), lds2, st, t->hdr, ktab,   \
                       sched, t->sched_rounds);                                                                      \
  } while (0)
    switch (t->cfg) {
      case 1: LAUNCHS(256, 256, 2, 4, 32, 4); break;
      case 2: LAUNCHS(256, 128, 4, 2, 32, 5); break;
      default: LAUNCHS(128, 256, 2, 4, 32, 5); break;
    }
#undef LAUNCHS
  } else if (t->v3 && wgrad_persist() && t->total_blocks > cap) {
    const dim3 pgrid(cap);
#define LAUNCHP(A, B, C_, D, KS_, S_)                                                                                \
  do {                                                                                                               \
    static bool a_ = false;                                                                                          \
    if (!a_) {                                                                                                       \
      hipFuncSetAttribute((const void*)wgrad_pipe_multi_persist_kernel<A, B, C_, D, KS_, S_>,                        \
                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);                                    \
      a_ = true;                                                                                                     \
    }                                                                                                                \
    hipLaunchKernelGGL((wgrad_pipe_multi_persist_kernel<A, B, C_, D, KS_, S_>), pgrid, dim3(64 * C_ * D), lds2, st, t->hdr, ktab, \
                       t->total_blocks);                                                                             \
  } while (0)
    switch (t->cfg) {
      case 1: LAUNCHP(256, 256, 2, 4, 32, 4); break;
      case 2: LAUNCHP(256, 128, 4, 2, 32, 5); break;
      default: LAUNCHP(128, 256, 2, 4, 32, 5); break;
    }
#undef LAUNCHP
  } else if (t->v3) {
    switch (t->cfg) {
      case 1: LAUNCHM(wgrad_pipe_multi_kernel, 256, 256, 2, 4, 32, 4); break;
      case 2: LAUNCHM(wgrad_pipe_multi_kernel, 256, 128, 4, 2, 32, 5); break;
      default: LAUNCHM(wgrad_pipe_multi_kernel, 128, 256, 2, 4, 32, 5); break;
    }
  } else {
    switch (t->cfg) {
      case 1: LAUNCHM(wgrad_glds_multi_kernel, 256, 256, 2, 4, 64, 2); break;
      case 2: LAUNCHM(wgrad_glds_multi_kernel, 256, 128, 4, 2, 64, 3); break;
      case 3: LAUNCHM(wgrad_glds_multi_kernel, 128, 256, 2, 4, 64, 3); break;
      default: LAUNCHM(wgrad_glds_multi_kernel, 128, 128, 2, 2, 64, 2); break;
    }
  }
#undef LAUNCHM
  dsl_prof_end(prof, st);
  DSL_LAUNCH_CHECK("wgrad_glds_multi_kernel");
  if (t->n_red > 0) {
    hipLaunchKernelGGL(wgrad_reduce_multi_kernel, dim3(t->red_blocks), dim3(256), 0, st, rtab, t->n_red);
    DSL_LAUNCH_CHECK("wgrad_reduce_multi_kernel");
  }
  for (int i = 0; i < t->n_colsum; ++i) {
    const ColsumItem& c = t->colsum[i];
    const int rc = c.clear ? dsl_colsum(c.x, c.out, (long)c.rows, c.c, c.ld, stream) : dsl_colsum_acc(c.x, c.out, (long)c.rows, c.c, c.ld, stream);
    if (rc) return rc;
  }
  return 0;
}

extern "C" int dsl_conv2d_wgrad(const dsl_wgrad_desc* d, void* stream) {
  DSL_CHECK(d != nullptr, "dsl_conv2d_wgrad: null descriptor");
  return wgrad_launch(d, 1, stream);
}

extern "C" int dsl_conv2d_wgrad_group(const dsl_wgrad_desc* descs, int count, void* stream) {
  DSL_CHECK(descs != nullptr, "dsl_conv2d_wgrad_group: null descriptors");
  return wgrad_launch(descs, count, stream);
}

#ifdef DSL_TRACE_BUILD
// tools/trace_conv.py: the s_memtime stamps of the traced workgroup (DSL_TRACE_WG) of the last conv_pipe launch
extern "C" int dsl_debug_conv_trace(unsigned long long* host_out) {
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_conv_trace), sizeof(unsigned long long) * (8 * 40 * 8 + 8 * 16 + 8)) == hipSuccess ? 0 : -1;
}
#endif
