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
  int ktiles, kc;
  long long wrow;
  const uint16_t* src;
  const uint16_t* wgt;
  void* dst;
  const float* scale;
  const float* bias;
  const uint16_t* addend;
  const uint16_t* mask;
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
        v = *reinterpret_cast<const u32x4*>(p.src + pix * p.cs + coff);
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
  const bool out_f32 = (p.flags & DSL_CONV_OUT_F32) != 0;
  const bool relu_out = (p.flags & DSL_CONV_RELU_OUT) != 0;
  const bool mask_first = (p.flags & DSL_CONV_MASK_FIRST) != 0 && p.mask != nullptr;
  const bool mask_last = (p.flags & DSL_CONV_MASK_LAST) != 0 && p.mask != nullptr;
  const bool has_mask = mask_first || mask_last;
#pragma unroll
  for (int pt = 0; pt < 2; ++pt) {
    const int gp = px0 + wave_px * 64 + pt * 32 + frow;
    if (gp >= totpx) continue;
    int seg, img, y, x;
    decode_pixel(p, gp, seg, img, y, x);
    const int oy = y * p.os, ox = x * p.os;
    const long long dpix = p.doff[seg] + ((long long)img * p.dh[seg] + oy) * p.dw[seg] + ox;
    long long apix = dpix;
    if (p.flags & DSL_CONV_ADD_UPSAMPLE) {
      const int ay = (oy * p.ah[seg]) / p.dh[seg], ax = (ox * p.aw[seg]) / p.dw[seg];
      apix = p.aoff[seg] + ((long long)img * p.ah[seg] + ay) * p.aw[seg] + ax;
    }
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int co = co0 + wave_co * WM + ct * 32 + 8 * g + 4 * fhalf;
        if (co >= p.cd) continue;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[ct][pt][4 * g + e];
        const bool full = co + 3 < p.cd;
        if (full) {
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
    }
  }
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
  long long krow;
  const uint16_t* dy;
  const uint16_t* x;
  float* ws;
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
          v = *reinterpret_cast<const u32x4*>(p.x + pix * p.cs + ci0 + xchunk * 8);
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

__global__ void wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw,
                                    const float* __restrict__ scale, int splits, int cy, int cd,
                                    long long krow) {
  const long long total4 = (long long)cd * krow / 4;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total4;
       i += (long long)gridDim.x * blockDim.x) {
    const long long e = i * 4;
    const int co = (int)(e / krow);
    const long long k = e - (long long)co * krow;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int sp = 0; sp < splits; ++sp)
      s += *reinterpret_cast<const f32x4*>(ws + ((long long)sp * cy + co) * krow + k);
    if (scale) s *= scale[co];
    *reinterpret_cast<f32x4*>(dw + e) = s;
  }
}

}  // namespace

// ================================================================================================
// host side
// ================================================================================================
extern "C" int dsl_conv2d(const dsl_conv_desc* d, void* stream) {
  DSL_CHECK(d != nullptr, "dsl_conv2d: null descriptor");
  DSL_CHECK(d->nseg >= 1 && d->nseg <= DSL_MAX_SEG, "dsl_conv2d: nseg=%d out of range", d->nseg);
  DSL_CHECK(d->src && d->wgt && d->dst, "dsl_conv2d: null tensor pointer");
  const bool smallc = (d->flags & DSL_CONV_SMALL_C) != 0;
  if (smallc)
    DSL_CHECK(d->cs == 8 && d->mode == 0, "dsl_conv2d: SMALL_C needs cs == 8, forward mode");
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
  k.cs = d->cs; k.cd = d->cd; k.ldd = d->ldd; k.lda = d->lda; k.ldm = d->ldm;
  k.kh = d->kh; k.kw = d->kw; k.stride = d->stride; k.pad = d->pad; k.mode = d->mode; k.os = d->os;
  k.flags = d->flags;
  if (smallc) {
    k.ktiles = (d->kh * d->kw + 7) / 8;
    k.kc = 1;
  } else {
    k.kc = d->cs / 64;
    k.ktiles = d->kh * d->kw * k.kc;
  }
  k.wrow = (long long)k.ktiles * BK;
  k.src = (const uint16_t*)d->src; k.wgt = (const uint16_t*)d->wgt; k.dst = d->dst;
  k.scale = d->scale; k.bias = d->bias;
  k.addend = (const uint16_t*)d->addend; k.mask = (const uint16_t*)d->mask;

  const int bco = (d->cd_pad % 128 == 0) ? 128 : 64;
  dim3 grid(d->cd_pad / bco, (px + BPX - 1) / BPX);
  const size_t lds = 2 * (size_t)(bco + BPX) * BK * 2;
  hipStream_t st = (hipStream_t)stream;
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
  if (dsl_prof_active()) {
    const double cin_real = smallc ? 3.0 : (double)d->cs;
    prof = dsl_prof_begin((bco == 128 && !smallc) ? 0 : 1, 2.0 * px * (double)d->cd * d->kh * d->kw * cin_real, st);
  }
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

static int wgrad_geometry(const dsl_wgrad_desc* d, int* ktiles, int* tiles, int* bco) {
  long long px = 0;
  for (int s = 0; s < d->nseg; ++s) px += (long long)d->n * d->gh[s] * d->gw[s];
  *ktiles = (int)((px + 63) / 64);
  *bco = (d->cy % 128 == 0) ? 128 : 64;
  *tiles = (d->cy / *bco) * (d->kh * d->kw * d->cs / 128);
  return 0;
}

extern "C" int dsl_wgrad_splits(const dsl_wgrad_desc* d) {
  int ktiles, tiles, bco;
  wgrad_geometry(d, &ktiles, &tiles, &bco);
  int splits = (768 + tiles - 1) / tiles;
  const int max_by_k = ktiles / 4 > 0 ? ktiles / 4 : 1;    // at least 4 K stages per split
  if (splits > max_by_k) splits = max_by_k;
  if (splits < 1) splits = 1;
  if (splits > 256) splits = 256;
  return splits;
}

extern "C" size_t dsl_wgrad_workspace_bytes(const dsl_wgrad_desc* d) {
  const int splits = d->splits > 0 ? d->splits : dsl_wgrad_splits(d);
  return (size_t)splits * d->cy * (size_t)d->kh * d->kw * d->cs * sizeof(float);
}

extern "C" int dsl_colsum(const void* x, float* out, long rows, int c, int ld, void* stream);

extern "C" int dsl_conv2d_wgrad(const dsl_wgrad_desc* d, void* stream) {
  DSL_CHECK(d != nullptr, "dsl_conv2d_wgrad: null descriptor");
  DSL_CHECK(d->nseg >= 1 && d->nseg <= DSL_MAX_SEG, "dsl_conv2d_wgrad: nseg=%d", d->nseg);
  DSL_CHECK(d->cs % 128 == 0, "dsl_conv2d_wgrad: Cin=%d must be a multiple of 128", d->cs);
  DSL_CHECK(d->cy % 64 == 0 && d->cd <= d->cy, "dsl_conv2d_wgrad: bad cy=%d cd=%d", d->cy, d->cd);
  DSL_CHECK(d->dy && d->x && d->dw && d->workspace, "dsl_conv2d_wgrad: null pointer");
  int ktiles, tiles, bco;
  wgrad_geometry(d, &ktiles, &tiles, &bco);
  const int splits = d->splits > 0 ? d->splits : dsl_wgrad_splits(d);
  DSL_CHECK(d->workspace_bytes >= dsl_wgrad_workspace_bytes(d), "dsl_conv2d_wgrad: workspace too small (%zu < %zu)",
            d->workspace_bytes, dsl_wgrad_workspace_bytes(d));
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
  k.pxstart[d->nseg] = px;
  k.cs = d->cs; k.cy = d->cy; k.kh = d->kh; k.kw = d->kw; k.stride = d->stride; k.pad = d->pad;
  k.ktiles = ktiles;
  k.tiles_per_split = (ktiles + splits - 1) / splits;
  k.ctiles_per_tap = d->cs / 128;
  k.krow = (long long)d->kh * d->kw * d->cs;
  k.dy = (const uint16_t*)d->dy; k.x = (const uint16_t*)d->x; k.ws = (float*)d->workspace;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(d->cy / bco, d->kh * d->kw * d->cs / 128, splits);
  const size_t lds = 2 * (size_t)(64 * bco * 2 + 64 * 256);
  const int prof = dsl_prof_active() ? dsl_prof_begin(2, 2.0 * px * (double)d->cd * d->kh * d->kw * d->cs, st) : -1;
  if (bco == 128) {
    static bool a = false;
    if (!a) { hipFuncSetAttribute((const void*)wgrad_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); a = true; }
    hipLaunchKernelGGL((wgrad_kernel<128>), grid, dim3(256), lds, st, k);
  } else {
    static bool a = false;
    if (!a) { hipFuncSetAttribute((const void*)wgrad_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); a = true; }
    hipLaunchKernelGGL((wgrad_kernel<64>), grid, dim3(256), lds, st, k);
  }
  dsl_prof_end(prof, st);
  DSL_LAUNCH_CHECK("wgrad_kernel");
  const long long total4 = (long long)d->cd * k.krow / 4;
  int rb = (int)((total4 + 255) / 256);
  if (rb > 4096) rb = 4096;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(rb), dim3(256), 0, st, (const float*)d->workspace, d->dw,
                     d->scale, splits, d->cy, d->cd, k.krow);
  DSL_LAUNCH_CHECK("wgrad_reduce_kernel");
  if (d->db) return dsl_colsum(d->dy, d->db, (long)px, d->cd, d->cy, stream);
  return 0;
}
