// Memory-bound layers of the RLA_ResNet backbone (mmdet/models/backbones/resnet_rla.py of the reference): the
// recurrent-layer-aggregation path h <- conv3x3(tanh(BN(h + conv1x1(out)))) next to every bottleneck (:289-327), the 2x2
// average pool of h at stage transitions (:98-100,129-130), and what TRAINABLE eval-mode BatchNorm affine parameters need (the reference freezes only
// the statistics, norm_eval, :380-388): the per-step fold gamma/sqrt(var+eps), and the (gamma, beta) gradients of a
// conv -> BN pair from the unscaled weight gradient.
// All tensors NHWC bf16 with explicit row strides ("ld", in elements): the block input of RLA is cat(x, h), kept as ONE
// buffer [pixel][C + 128] = [x | h (32) | zeros] so that conv1 reads it as an ordinary source.
#include "common.hpp"

namespace {

// ---- 2x2 stride-2 average pool (nn.AvgPool2d((2, 2), stride=(2, 2))) and its backward ------------------------------------
__global__ void avgpool2_kernel(const uint16_t* __restrict__ x, int ldx, uint16_t* __restrict__ y, int ldy, int n, int h, int w,
                                int c8) {
  const int oh = h / 2, ow = w / 2;
  const long long total = (long long)n * oh * ow * c8;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int k = (int)(i % c8) * 8;
    long long t = i / c8;
    const int ox = (int)(t % ow);
    t /= ow;
    const int oy = (int)(t % oh);
    const int b = (int)(t / oh);
    float a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
      for (int dx = 0; dx < 2; ++dx) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(x + (((long long)b * h + 2 * oy + dy) * w + 2 * ox + dx) * ldx + k);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          a[2 * e] += bflo(v[e]);
          a[2 * e + 1] += bfhi(v[e]);
        }
      }
    u32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = pack2bf(a[2 * e] * 0.25f, a[2 * e + 1] * 0.25f);
    *reinterpret_cast<u32x4*>(y + (((long long)b * oh + oy) * ow + ox) * ldy + k) = o;
  }
}

__global__ void avgpool2_bwd_kernel(const uint16_t* __restrict__ gy, int ldgy, uint16_t* __restrict__ gx, int ldgx, int n, int h,
                                    int w, int c8) {
  const int oh = h / 2, ow = w / 2;
  const long long total = (long long)n * h * w * c8;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int k = (int)(i % c8) * 8;
    long long t = i / c8;
    const int x_ = (int)(t % w);
    t /= w;
    const int y_ = (int)(t % h);
    const int b = (int)(t / h);
    u32x4 o = {0u, 0u, 0u, 0u};
    if (y_ / 2 < oh && x_ / 2 < ow) {      // odd trailing rows / columns are outside every pooling window
      const u32x4 v = *reinterpret_cast<const u32x4*>(gy + (((long long)b * oh + y_ / 2) * ow + x_ / 2) * ldgy + k);
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = pack2bf(bflo(v[e]) * 0.25f, bfhi(v[e]) * 0.25f);
    }
    *reinterpret_cast<u32x4*>(gx + (((long long)b * h + y_) * w + x_) * ldgx + k) = o;
  }
}

// ---- t = tanh(u * scale[c] + bias[c])   (eval-mode BN folded to scale / bias) ---------------------------------------------
__global__ void bn_tanh_kernel(const uint16_t* __restrict__ u, int ldu, const float* __restrict__ scale,
                               const float* __restrict__ bias, uint16_t* __restrict__ t, int ldt, long long rows, int c8) {
  const long long total = rows * c8;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long r = i / c8;
    const int k = (int)(i - r * c8) * 8;
    const u32x4 a = *reinterpret_cast<const u32x4*>(u + r * ldu + k);
    u32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float lo = tanhf(__builtin_fmaf(bflo(a[e]), scale[k + 2 * e], bias[k + 2 * e]));        // (ONE rounding: rla_tail_fwd_kernel
      const float hi = tanhf(__builtin_fmaf(bfhi(a[e]), scale[k + 2 * e + 1], bias[k + 2 * e + 1]));   //  below must round the same way)
      o[e] = pack2bf(lo, hi);
    }
    *reinterpret_cast<u32x4*>(t + r * ldt + k) = o;
  }
}

// ---- the recurrent path of one block as ONE launch (round 6): u = h' + conv_out(out) (1x1, C4 -> 32), t = tanh(bn(u)),
// h = recurrent_conv(t) (3x3, 32 -> 32, pad 1) - resnet_rla.py:125-136 - which were three dependent launches on the chain that the next
// block's conv1 waits for (dsl_conv2d with 32 of 64 padded couts, dsl_bn_tanh_fwd, dsl_conv2d).  A workgroup (8 waves) owns a
// 14 x 14 pixel tile of one image: phase A computes conv_out for the tile + a one-pixel halo (16 x 16 = 256 slots, 32 per wave: MFMA
// 32x32x16, A = the 32 weight rows, B = the pixels' rows, both fragments straight from global memory / L2 - 2 x 32 B per pixel row and
// K step), adds h', rounds u, applies BN + tanh and leaves t in an LDS patch [slot][32] (zero outside the image: the 3x3's padding);
// phase B reads the nine taps out of the patch (7 waves x 32 pixels).  u and t still go to memory (the backward pass reads them),
// halo slots are recomputed by the neighbouring tiles (identical values), only a tile's own pixels are stored.
// Same K order per MFMA chain and the same rounding points as the three launches (u = bf16(acc + h'), t = bf16(tanh(fma(u, s, b))),
// h = bf16(acc)); the zero-padded K columns of the stored 3x3 weights (32 real of 64 / 128) are skipped (they add 0).
typedef __bf16 rla_bf16x8 __attribute__((ext_vector_type(8)));
typedef float rla_f32x16 __attribute__((ext_vector_type(16)));
struct RlaTailK {
  const uint16_t* x; const uint16_t* hin; const uint16_t* wco; const float* sc; const float* bi; const uint16_t* wrc;
  uint16_t* u; uint16_t* t; uint16_t* hout;
  int ldx, ldh, c4, tw, ldo, n, h, w, tiles_x, tiles_y;
};
// <RT, KS>: tile edge RT (halo patch (RT + 2)^2 slots = NS groups of 32) and KS waves per slot group, each taking 1 / KS of conv_out's
// K range (NS * KS = 8 waves): <14, 1> for the many-pixel stages; <6, 4> where an image is a few tiles and conv_out's K is long
// (stages 2, 3: a wave's chain of K / 16 dependent load -> MFMA steps is what the launch takes) - the four partial sums of a slot
// meet in LDS and are added in K order.
template <int RT, int KS>
__global__ __launch_bounds__(512) void rla_tail_fwd_kernel(const RlaTailK p) {
  constexpr int HW = RT + 2, NSLOT = HW * HW, NS = NSLOT / 32;
  static_assert(NSLOT % 32 == 0 && NS * KS == 8, "eight waves: slot groups x K parts");
  __shared__ __attribute__((aligned(16))) uint16_t patch[NSLOT * 32];
  __shared__ __attribute__((aligned(16))) float red[KS > 1 ? (KS - 1) * NSLOT * 32 : 4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int col = lane & 31, kh = lane >> 5;
  int b = blockIdx.x;
  const int tx = b % p.tiles_x; b /= p.tiles_x;
  const int ty = b % p.tiles_y;
  const int img = b / p.tiles_y;
  const int y0 = ty * RT, x0 = tx * RT;
  const long long ibase = (long long)img * p.h * p.w;
  // ---- phase A: conv_out + h' -> u -> t for halo slot s = 32 * (slot group) + col
  {
    const int sg = wave % NS, kq = wave / NS;
    const int s = sg * 32 + col;
    const int sy = s / HW, sx = s - sy * HW;
    const int py = y0 - 1 + sy, px = x0 - 1 + sx;
    const bool valid = (unsigned)py < (unsigned)p.h && (unsigned)px < (unsigned)p.w;
    const long long pix = ibase + (valid ? (long long)py * p.w + px : 0);
    const int klen = p.c4 / KS;
    const uint16_t* xr = p.x + pix * p.ldx + kq * klen + kh * 8;
    const uint16_t* wr = p.wco + (long long)col * p.c4 + kq * klen + kh * 8;
    rla_f32x16 acc;
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = 0.f;
    // eight K steps of fragments in flight beside the eight being multiplied (the loop is the loads' latency, not their bandwidth)
    constexpr int UB = 8;
    rla_bf16x8 fa[UB], fb[UB];
#pragma unroll
    for (int i = 0; i < UB; ++i) {
      const int k0 = i * 16 < klen ? i * 16 : 0;
      fa[i] = *reinterpret_cast<const rla_bf16x8*>(wr + k0);
      fb[i] = *reinterpret_cast<const rla_bf16x8*>(xr + k0);
    }
    for (int kb = 0; kb < klen; kb += UB * 16) {
      rla_bf16x8 ca[UB], cb[UB];
#pragma unroll
      for (int i = 0; i < UB; ++i) { ca[i] = fa[i]; cb[i] = fb[i]; }
      const int kn = kb + UB * 16;
#pragma unroll
      for (int i = 0; i < UB; ++i) {
        const int k0 = kn + i * 16 < klen ? kn + i * 16 : 0;        // (past the end: a harmless re-load, never multiplied)
        fa[i] = *reinterpret_cast<const rla_bf16x8*>(wr + k0);
        fb[i] = *reinterpret_cast<const rla_bf16x8*>(xr + k0);
      }
#pragma unroll
      for (int i = 0; i < UB; ++i)
        if (kb + i * 16 < klen) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ca[i], cb[i], acc, 0, 0, 0);
    }
    if (KS > 1) {
      if (kq > 0) {
#pragma unroll
        for (int j = 0; j < 16; ++j) red[((kq - 1) * NSLOT + s) * 32 + (j >> 2) * 8 + 4 * kh + (j & 3)] = acc[j];
      }
      __syncthreads();
      if (kq == 0) {
#pragma unroll
        for (int q = 1; q < KS; ++q)
#pragma unroll
          for (int j = 0; j < 16; ++j) acc[j] += red[((q - 1) * NSLOT + s) * 32 + (j >> 2) * 8 + 4 * kh + (j & 3)];
      }
    }
    if (kq == 0) {
      const bool own = valid && sy >= 1 && sy <= RT && sx >= 1 && sx <= RT;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int co = 8 * g + 4 * kh;
        uint2 tq = {0u, 0u};
        if (valid) {
          const uint2 hv = *reinterpret_cast<const uint2*>(p.hin + pix * p.ldh + co);
          const uint32_t u0 = pack2bf(acc[4 * g] + bflo(hv.x), acc[4 * g + 1] + bfhi(hv.x));
          const uint32_t u1 = pack2bf(acc[4 * g + 2] + bflo(hv.y), acc[4 * g + 3] + bfhi(hv.y));
          const f32x4 s4 = *reinterpret_cast<const f32x4*>(p.sc + co), b4 = *reinterpret_cast<const f32x4*>(p.bi + co);
          tq.x = pack2bf(tanhf(__builtin_fmaf(bflo(u0), s4[0], b4[0])), tanhf(__builtin_fmaf(bfhi(u0), s4[1], b4[1])));
          tq.y = pack2bf(tanhf(__builtin_fmaf(bflo(u1), s4[2], b4[2])), tanhf(__builtin_fmaf(bfhi(u1), s4[3], b4[3])));
          if (own) {
            *reinterpret_cast<uint2*>(p.u + pix * 32 + co) = uint2{u0, u1};
            *reinterpret_cast<uint2*>(p.t + pix * p.tw + co) = tq;
          }
        }
        *reinterpret_cast<uint2*>(patch + s * 32 + co) = tq;
      }
    }
  }
  __syncthreads();
  // ---- phase B: the 3x3 over the patch for own pixel q = 32 * wave + col
  if (wave * 32 < RT * RT) {
    const int q = wave * 32 + col;
    const bool live = q < RT * RT;
    const int iy = live ? q / RT : 0, ix = live ? q - (q / RT) * RT : 0;
    const uint16_t* wr = p.wrc + (long long)col * 9 * p.tw + kh * 8;
    rla_f32x16 acc;
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = 0.f;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int ky = tap / 3, kx = tap - ky * 3;
      const uint16_t* pr = patch + ((iy + ky) * HW + ix + kx) * 32 + kh * 8;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const rla_bf16x8 a = *reinterpret_cast<const rla_bf16x8*>(wr + tap * p.tw + ks * 16);
        const rla_bf16x8 bb = *reinterpret_cast<const rla_bf16x8*>(pr + ks * 16);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bb, acc, 0, 0, 0);
      }
    }
    const int py = y0 + iy, px = x0 + ix;
    if (live && py < p.h && px < p.w) {
      uint16_t* o = p.hout + (ibase + (long long)py * p.w + px) * p.ldo + 4 * kh;
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<uint2*>(o + 8 * g) = uint2{pack2bf(acc[4 * g], acc[4 * g + 1]), pack2bf(acc[4 * g + 2], acc[4 * g + 3])};
    }
  }
}

// backward: g_v = g_t (1 - t^2); g_u = g_v * scale;  block records of dgamma = sum g_v (u - mean) invstd, dbeta = sum g_v
constexpr int BT_T = 256, BT_ROWS = 256;       // rows per block
__global__ __launch_bounds__(BT_T) void bn_tanh_bwd_kernel(const uint16_t* __restrict__ gt, int ldgt, const uint16_t* __restrict__ t,
                                                            int ldt, const uint16_t* __restrict__ u, int ldu,
                                                            const float* __restrict__ scale, const float* __restrict__ mean,
                                                            const float* __restrict__ var, float eps, uint16_t* __restrict__ gu,
                                                            int ldgu, float* __restrict__ rec, long long rows, int c8) {
  __shared__ float sh[BT_T * 16];
  const int c = c8 * 8;
  const int rpi = BT_T / c8;                      // rows per iteration
  const int chunk = threadIdx.x % c8, prow = threadIdx.x / c8;
  const int k = chunk * 8;
  float sc[8], mu[8], is[8], dg[8], db[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    sc[e] = scale[k + e];
    mu[e] = mean[k + e];
    is[e] = rsqrtf(var[k + e] + eps);
    dg[e] = db[e] = 0.f;
  }
  const long long r0 = (long long)blockIdx.x * BT_ROWS, r1 = min(r0 + (long long)BT_ROWS, rows);
  if (prow < rpi)
    for (long long r = r0 + prow; r < r1; r += rpi) {
      const u32x4 gv = *reinterpret_cast<const u32x4*>(gt + r * ldgt + k);
      const u32x4 tv = *reinterpret_cast<const u32x4*>(t + r * ldt + k);
      const u32x4 uv = *reinterpret_cast<const u32x4*>(u + r * ldu + k);
      float o[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float g = (e & 1) ? bfhi(gv[e >> 1]) : bflo(gv[e >> 1]);
        const float tt = (e & 1) ? bfhi(tv[e >> 1]) : bflo(tv[e >> 1]);
        const float uu = (e & 1) ? bfhi(uv[e >> 1]) : bflo(uv[e >> 1]);
        const float gvv = g * (1.f - tt * tt);
        dg[e] += gvv * (uu - mu[e]) * is[e];
        db[e] += gvv;
        o[e] = gvv * sc[e];
      }
      *reinterpret_cast<u32x4*>(gu + r * ldgu + k) = u32x4{pack2bf(o[0], o[1]), pack2bf(o[2], o[3]), pack2bf(o[4], o[5]), pack2bf(o[6], o[7])};
    }
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    sh[threadIdx.x * 16 + e] = (prow < rpi) ? dg[e] : 0.f;
    sh[threadIdx.x * 16 + 8 + e] = (prow < rpi) ? db[e] : 0.f;
  }
  __syncthreads();
  for (int ch = threadIdx.x; ch < c; ch += BT_T) {
    float a = 0.f, b = 0.f;
    for (int r = 0; r < rpi; ++r) {
      a += sh[(r * c8 + ch / 8) * 16 + (ch % 8)];
      b += sh[(r * c8 + ch / 8) * 16 + 8 + (ch % 8)];
    }
    rec[(long long)blockIdx.x * 2 * c + ch] = a;
    rec[(long long)blockIdx.x * 2 * c + c + ch] = b;
  }
}

// ---- backward of the recurrent path's tail as ONE launch (round 6): g_t = recurrent_conv^T(g_h) (3x3 data gradient, 32 <- 32), then
// bn_tanh_bwd_kernel's arithmetic on it - g_v = g_t (1 - t^2), g_u = g_v * scale, per-tile records of dgamma / dbeta - which were a
// dsl_conv2d (mode 1, 32 of 64 padded channels) and dsl_bn_tanh_bwd per block on the data-gradient chain.  A workgroup (8 waves) owns a
// 14 x 14 tile: the g_h rows of the tile + a one-pixel halo go to an LDS patch (zero outside the image), 7 waves x 32 pixels run the
// nine taps as MFMA chains (A = the dgrad pack's rows [ci][r][s][co], B = the patch row of pixel (y + 1 - r, x + 1 - s): the K order of
// the convolution kernel, its zero co columns skipped), g_t is rounded to bf16 as the convolution stored it and never leaves the chip.
struct RlaTailBK {
  const uint16_t* gh; const uint16_t* wT; const uint16_t* t; const uint16_t* u;
  const float* scale; const float* mean; const float* var;
  uint16_t* gu; float* rec;
  int ldgh, ldw, ldt, ldgu, n, h, w, tiles_x, tiles_y;
  float eps;
};
__global__ __launch_bounds__(512) void rla_tail_bwd_kernel(const RlaTailBK p) {
  constexpr int RT = 14, HW = 16;
  __shared__ __attribute__((aligned(16))) uint16_t patch[HW * HW * 32];
  __shared__ float red[7 * 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int col = lane & 31, kh = lane >> 5;
  int b = blockIdx.x;
  const int tx = b % p.tiles_x; b /= p.tiles_x;
  const int ty = b % p.tiles_y;
  const int img = b / p.tiles_y;
  const int y0 = ty * RT, x0 = tx * RT;
  const long long ibase = (long long)img * p.h * p.w;
  // g_h patch: 256 slots x 32 channels = 1024 16-byte chunks, two per thread
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int c16 = tid + it * 512;
    const int s = c16 >> 2, part = c16 & 3;
    const int py = y0 - 1 + (s >> 4), px = x0 - 1 + (s & 15);
    u32x4 v = {0u, 0u, 0u, 0u};
    if ((unsigned)py < (unsigned)p.h && (unsigned)px < (unsigned)p.w)
      v = *reinterpret_cast<const u32x4*>(p.gh + (ibase + (long long)py * p.w + px) * p.ldgh + part * 8);
    *reinterpret_cast<u32x4*>(patch + s * 32 + part * 8) = v;
  }
  __syncthreads();
  float dg[16], db[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) dg[j] = db[j] = 0.f;
  if (wave < 7) {
    const int q = wave * 32 + col;
    const bool live = q < RT * RT;
    const int iy = live ? q / RT : 0, ix = live ? q - (q / RT) * RT : 0;
    const uint16_t* wr = p.wT + (long long)col * 9 * p.ldw + kh * 8;
    rla_f32x16 acc;
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = 0.f;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int r = tap / 3, s_ = tap - r * 3;
      const uint16_t* pr = patch + ((iy + 2 - r) * HW + ix + 2 - s_) * 32 + kh * 8;     // patch row of pixel (y + 1 - r, x + 1 - s)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const rla_bf16x8 a = *reinterpret_cast<const rla_bf16x8*>(wr + tap * p.ldw + ks * 16);
        const rla_bf16x8 bb = *reinterpret_cast<const rla_bf16x8*>(pr + ks * 16);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bb, acc, 0, 0, 0);
      }
    }
    const int py = y0 + iy, px = x0 + ix;
    if (live && py < p.h && px < p.w) {
      const long long pix = ibase + (long long)py * p.w + px;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int c0 = 8 * g + 4 * kh;
        const uint2 tv = *reinterpret_cast<const uint2*>(p.t + pix * p.ldt + c0);
        const uint2 uv = *reinterpret_cast<const uint2*>(p.u + pix * 32 + c0);
        const uint2 gq = {pack2bf(acc[4 * g], acc[4 * g + 1]), pack2bf(acc[4 * g + 2], acc[4 * g + 3])};     // g_t as the convolution stored it
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float gg = (e & 1) ? bfhi(e < 2 ? gq.x : gq.y) : bflo(e < 2 ? gq.x : gq.y);
          const float tt = (e & 1) ? bfhi(e < 2 ? tv.x : tv.y) : bflo(e < 2 ? tv.x : tv.y);
          const float uu = (e & 1) ? bfhi(e < 2 ? uv.x : uv.y) : bflo(e < 2 ? uv.x : uv.y);
          const float gvv = gg * (1.f - tt * tt);
          dg[4 * g + e] = gvv * (uu - p.mean[c0 + e]) * rsqrtf(p.var[c0 + e] + p.eps);
          db[4 * g + e] = gvv;
          o[e] = gvv * p.scale[c0 + e];
        }
        *reinterpret_cast<uint2*>(p.gu + pix * p.ldgu + c0) = uint2{pack2bf(o[0], o[1]), pack2bf(o[2], o[3])};
      }
    }
    // the tile's (dgamma, dbeta) record: the 32 pixels of a wave by shuffles, the 7 waves through LDS, in a fixed order
#pragma unroll
    for (int j = 0; j < 16; ++j) {
#pragma unroll
      for (int o_ = 16; o_ > 0; o_ >>= 1) {
        dg[j] += __shfl_xor(dg[j], o_, 64);
        db[j] += __shfl_xor(db[j], o_, 64);
      }
    }
    if (col == 0) {
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const int c = 8 * (j >> 2) + 4 * kh + (j & 3);
        red[wave * 64 + c] = dg[j];
        red[wave * 64 + 32 + c] = db[j];
      }
    }
  }
  __syncthreads();
  if (tid < 64) {
    float a = 0.f;
#pragma unroll
    for (int w_ = 0; w_ < 7; ++w_) a += red[w_ * 64 + tid];
    p.rec[(long long)blockIdx.x * 64 + tid] = a;
  }
}

// out[v] (+)= sum over blocks of rec[b][v], fixed order; one workgroup
__global__ __launch_bounds__(256) void rec_sum_kernel(const float* __restrict__ rec, int nblocks, int V, float* __restrict__ out_a,
                                                      float* __restrict__ out_b, int half, int accumulate) {
  __shared__ float sh[256];
  const int Q = 256 / V;
  const int v = threadIdx.x % V, q = threadIdx.x / V;
  float a = 0.f;
  if (q < Q) {
#pragma unroll 8
    for (int b = q; b < nblocks; b += Q) a += rec[(long long)b * V + v];
  }
  sh[threadIdx.x] = a;
  __syncthreads();
  if (threadIdx.x < V) {
    float t = 0.f;
    for (int k = 0; k < Q; ++k) t += sh[k * V + threadIdx.x];
    float* dst = threadIdx.x < half ? out_a + threadIdx.x : out_b + (threadIdx.x - half);
    *dst = accumulate ? *dst + t : t;
  }
}

// the same for several record sets at once (one workgroup each): the recurrent path's BatchNorms of a stage whose image-split
// backward chains wrote their block records side by side
struct RecSumItem {
  const float* rec; float* out_a; float* out_b;
  int nrec, pad_;
};
__global__ __launch_bounds__(256) void rec_sum_multi_kernel(const RecSumItem* __restrict__ items, int V, int half) {
  __shared__ float sh[256];
  const RecSumItem it = items[blockIdx.x];
  const int Q = 256 / V;
  const int v = threadIdx.x % V, q = threadIdx.x / V;
  float a = 0.f;
  if (q < Q) {
#pragma unroll 8
    for (int b = q; b < it.nrec; b += Q) a += it.rec[(long long)b * V + v];
  }
  sh[threadIdx.x] = a;
  __syncthreads();
  if (threadIdx.x < V) {
    float t = 0.f;
    for (int k = 0; k < Q; ++k) t += sh[k * V + threadIdx.x];
    float* dst = threadIdx.x < half ? it.out_a + threadIdx.x : it.out_b + (threadIdx.x - half);
    *dst = t;
  }
}

// ---- eval-mode BN fold: scale = gamma / sqrt(var + eps), bias = beta - mean * scale --------------------------------------
__global__ void bn_fold_kernel(const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ mean,
                               const float* __restrict__ var, float eps, float* __restrict__ scale, float* __restrict__ bias, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float s = gamma[i] * rsqrtf(var[i] + eps);
  scale[i] = s;
  bias[i] = beta[i] - mean[i] * s;
}

// ---- (gamma, beta) gradients of conv -> BN(eval) from the UNSCALED weight gradient ----------------------------------------
// y = gamma * (conv(x) - mean) * invstd + beta.  With dWu[c][k] = sum_p g_y[p][c] x[p@k] (the weight gradient w.r.t. the raw
// conv output, no BN scale) and S[c] = sum_p g_y[p][c]:
//   sum_p g_y[p][c] conv(x)[p][c] = <W[c,:], dWu[c,:]>     (conv is linear in W)
//   dgamma[c] = (<W[c], dWu[c]> - mean[c] S[c]) * invstd[c],   dbeta[c] = S[c],   dW[c,:] = dWu[c,:] * gamma[c] * invstd[c]
// - no pre-BN activation has to be kept and nothing is divided by gamma (zero_init_last_bn starts bn3.weight at 0).
// One workgroup per weight row; dbeta already holds S (the weight-gradient launch's column sum of g_y).
struct BnPostItem {
  const float* w; float* dw; float* dgamma; const float* dbeta; const float* gamma; const float* mean; const float* var;
  int rows, k, row_start, pad_;
};
__global__ __launch_bounds__(256) void bn_wgrad_post_kernel(const BnPostItem* __restrict__ items, int n, float eps) {
  __shared__ float sh[16];
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if ((int)blockIdx.x >= items[mid].row_start) lo = mid; else hi = mid - 1;
  }
  const BnPostItem I = items[lo];
  const int c = blockIdx.x - I.row_start;
  const float* w = I.w + (long long)c * I.k;
  float* dw = I.dw + (long long)c * I.k;
  float dot = 0.f;
  for (int i = threadIdx.x * 4; i < I.k; i += 256 * 4) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(w + i), b = *reinterpret_cast<const f32x4*>(dw + i);
    dot += a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3];
  }
  dot = block_sum(dot, sh);
  const float invstd = rsqrtf(I.var[c] + eps);
  if (threadIdx.x == 0) I.dgamma[c] = (dot - I.mean[c] * I.dbeta[c]) * invstd;
  const float s = I.gamma[c] * invstd;
  for (int i = threadIdx.x * 4; i < I.k; i += 256 * 4) {
    f32x4 b = *reinterpret_cast<const f32x4*>(dw + i);
    b *= s;
    *reinterpret_cast<f32x4*>(dw + i) = b;
  }
}

inline int grid_for(long long total, int threads, int cap = 16384) {
  long long b = (total + threads - 1) / threads;
  return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

}  // namespace

extern "C" int dsl_avgpool2x2(const void* x, int ldx, void* y, int ldy, int n, int h, int w, int c, void* stream) {
  DSL_CHECK(x && y && c % 8 == 0 && ldx % 8 == 0 && ldy % 8 == 0 && h >= 2 && w >= 2, "dsl_avgpool2x2: bad arguments");
  hipLaunchKernelGGL(avgpool2_kernel, dim3(grid_for((long long)n * (h / 2) * (w / 2) * (c / 8), 256)), dim3(256), 0, (hipStream_t)stream,
                     (const uint16_t*)x, ldx, (uint16_t*)y, ldy, n, h, w, c / 8);
  DSL_LAUNCH_CHECK("avgpool2_kernel");
  return 0;
}

extern "C" int dsl_avgpool2x2_bwd(const void* gy, int ldgy, void* gx, int ldgx, int n, int h, int w, int c, void* stream) {
  DSL_CHECK(gy && gx && c % 8 == 0 && ldgy % 8 == 0 && ldgx % 8 == 0, "dsl_avgpool2x2_bwd: bad arguments");
  hipLaunchKernelGGL(avgpool2_bwd_kernel, dim3(grid_for((long long)n * h * w * (c / 8), 256)), dim3(256), 0, (hipStream_t)stream,
                     (const uint16_t*)gy, ldgy, (uint16_t*)gx, ldgx, n, h, w, c / 8);
  DSL_LAUNCH_CHECK("avgpool2_bwd_kernel");
  return 0;
}

extern "C" int dsl_bn_tanh_fwd(const void* u, int ldu, const float* scale, const float* bias, void* t, int ldt, long rows, int c,
                               void* stream) {
  DSL_CHECK(u && scale && bias && t && c % 8 == 0 && ldu % 8 == 0 && ldt % 8 == 0, "dsl_bn_tanh_fwd: bad arguments");
  hipLaunchKernelGGL(bn_tanh_kernel, dim3(grid_for((long long)rows * (c / 8), 256)), dim3(256), 0, (hipStream_t)stream,
                     (const uint16_t*)u, ldu, scale, bias, (uint16_t*)t, ldt, (long long)rows, c / 8);
  DSL_LAUNCH_CHECK("bn_tanh_kernel");
  return 0;
}

extern "C" size_t dsl_bn_tanh_bwd_workspace_bytes(long rows, int c) {
  return (size_t)((rows + BT_ROWS - 1) / BT_ROWS) * 2 * c * sizeof(float);
}

extern "C" int dsl_bn_tanh_bwd(const void* gt, int ldgt, const void* t, int ldt, const void* u, int ldu, const float* scale,
                               const float* mean, const float* var, float eps, void* gu, int ldgu, float* dgamma, float* dbeta,
                               void* workspace, long rows, int c, void* stream) {
  DSL_CHECK(gt && t && u && scale && mean && var && gu && workspace && (!dgamma == !dbeta), "dsl_bn_tanh_bwd: null pointer");
  DSL_CHECK(c % 8 == 0 && 2 * c <= 256 && 256 % (2 * c) == 0 && BT_T % (c / 8) == 0, "dsl_bn_tanh_bwd: unsupported channel count %d", c);
  const int nb = (int)((rows + BT_ROWS - 1) / BT_ROWS);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(bn_tanh_bwd_kernel, dim3(nb), dim3(BT_T), 0, st, (const uint16_t*)gt, ldgt, (const uint16_t*)t, ldt,
                     (const uint16_t*)u, ldu, scale, mean, var, eps, (uint16_t*)gu, ldgu, (float*)workspace, (long long)rows, c / 8);
  if (dgamma)      // NULL: the block records stay in `workspace` for dsl_rec_sum_multi
    hipLaunchKernelGGL(rec_sum_kernel, dim3(1), dim3(256), 0, st, (const float*)workspace, nb, 2 * c, dgamma, dbeta, c, 0);
  DSL_LAUNCH_CHECK("bn_tanh_bwd_kernel");
  return 0;
}

extern "C" size_t dsl_rla_tail_bwd_workspace_bytes(int n, int h, int w) {
  return n > 0 && h > 0 && w > 0 ? (size_t)n * ((h + 13) / 14) * ((w + 13) / 14) * 64 * sizeof(float) : 0;
}

extern "C" int dsl_rla_tail_bwd(const void* g_h, int ldgh, const void* wT_recurrent, int ldw, const void* t, int ldt, const void* u,
                                const float* scale, const float* mean, const float* var, float eps, void* g_u, int ldgu, float* dgamma,
                                float* dbeta, void* workspace, int n, int h, int w, void* stream) {
  DSL_CHECK(g_h && wT_recurrent && t && u && scale && mean && var && g_u && workspace && (!dgamma == !dbeta), "dsl_rla_tail_bwd: null pointer");
  DSL_CHECK(n > 0 && h > 0 && w > 0 && ldgh >= 32 && ldgh % 8 == 0 && ldw >= 32 && ldw % 8 == 0 && ldt >= 32 && ldt % 4 == 0 && ldgu >= 32 &&
            ldgu % 4 == 0, "dsl_rla_tail_bwd: bad geometry (ldgh=%d ldw=%d ldt=%d ldgu=%d)", ldgh, ldw, ldt, ldgu);
  DSL_CHECK((((uintptr_t)g_h | (uintptr_t)wT_recurrent) & 15) == 0 && (((uintptr_t)t | (uintptr_t)u | (uintptr_t)g_u) & 7) == 0,
            "dsl_rla_tail_bwd: g_h and the pack must be 16-byte aligned, t / u / g_u 8-byte aligned");
  RlaTailBK k{(const uint16_t*)g_h, (const uint16_t*)wT_recurrent, (const uint16_t*)t, (const uint16_t*)u, scale, mean, var, (uint16_t*)g_u,
              (float*)workspace, ldgh, ldw, ldt, ldgu, n, h, w, (w + 13) / 14, (h + 13) / 14, eps};
  const long long blocks = (long long)k.tiles_x * k.tiles_y * n;
  DSL_CHECK(blocks < 0x7fffffffLL, "dsl_rla_tail_bwd: grid too large");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(rla_tail_bwd_kernel, dim3((unsigned)blocks), dim3(512), 0, st, k);
  if (dgamma)      // NULL: the tile records stay in `workspace` ([tiles][64]: dgamma | dbeta) for dsl_rec_sum_multi
    hipLaunchKernelGGL(rec_sum_kernel, dim3(1), dim3(256), 0, st, (const float*)workspace, (int)blocks, 64, dgamma, dbeta, 32, 0);
  DSL_LAUNCH_CHECK("rla_tail_bwd_kernel");
  return 0;
}

extern "C" int dsl_rec_sum_multi(const dsl_rec_sum_item* items_dev, int n, int c, void* stream) {
  static_assert(sizeof(dsl_rec_sum_item) == sizeof(RecSumItem), "item layout");
  DSL_CHECK(items_dev && n > 0 && c > 0 && 2 * c <= 256 && 256 % (2 * c) == 0, "dsl_rec_sum_multi: bad arguments");
  hipLaunchKernelGGL(rec_sum_multi_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, (const RecSumItem*)items_dev, 2 * c, c);
  DSL_LAUNCH_CHECK("rec_sum_multi_kernel");
  return 0;
}

extern "C" int dsl_bn_fold(const float* gamma, const float* beta, const float* mean, const float* var, float eps, float* scale,
                           float* bias, int n, void* stream) {
  DSL_CHECK(gamma && beta && mean && var && scale && bias && n > 0, "dsl_bn_fold: bad arguments");
  hipLaunchKernelGGL(bn_fold_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, gamma, beta, mean, var, eps, scale, bias, n);
  DSL_LAUNCH_CHECK("bn_fold_kernel");
  return 0;
}

extern "C" int dsl_bn_wgrad_post(const dsl_bn_post_item* items_dev, int n, int total_rows, float eps, void* stream) {
  static_assert(sizeof(dsl_bn_post_item) == sizeof(BnPostItem), "item layout");
  DSL_CHECK(items_dev && n > 0 && total_rows > 0, "dsl_bn_wgrad_post: bad arguments");
  hipLaunchKernelGGL(bn_wgrad_post_kernel, dim3(total_rows), dim3(256), 0, (hipStream_t)stream, (const BnPostItem*)items_dev, n, eps);
  DSL_LAUNCH_CHECK("bn_wgrad_post_kernel");
  return 0;
}

extern "C" int dsl_rla_tail_fwd(const void* x, int ldx, const void* h_in, int ldh, const void* w_conv_out, int c4, const float* bn_scale,
                                const float* bn_bias, const void* w_recurrent, int tw, void* u, void* t, void* h_out, int ldo, int n,
                                int h, int w, void* stream) {
  DSL_CHECK(x && h_in && w_conv_out && bn_scale && bn_bias && w_recurrent && u && t && h_out, "dsl_rla_tail_fwd: null pointer");
  DSL_CHECK(n > 0 && h > 0 && w > 0 && c4 > 0 && c4 % 16 == 0 && ldx >= c4 && ldx % 8 == 0 && ldh >= 32 && ldh % 4 == 0 && tw >= 32 &&
            tw % 8 == 0 && ldo >= 32 && ldo % 4 == 0, "dsl_rla_tail_fwd: bad geometry (c4=%d ldx=%d ldh=%d tw=%d ldo=%d)", c4, ldx, ldh, tw, ldo);
  DSL_CHECK((((uintptr_t)x | (uintptr_t)w_conv_out | (uintptr_t)w_recurrent) & 15) == 0 && (((uintptr_t)h_in | (uintptr_t)u | (uintptr_t)t |
            (uintptr_t)h_out) & 7) == 0, "dsl_rla_tail_fwd: x and the weights must be 16-byte aligned, h_in / u / t / h_out 8-byte aligned");
  // tile form: 14 x 14 tiles unless the image is a few tiles of them and K is long (dsl_set_option "rla_tail_form": 0 = this rule,
  // 1 / 2 = force <14, 1> / <6, 4>: tools/bench_rla_tail.py)
  const int form = dsl_option("rla_tail_form");
  const bool small = form == 2 || (form == 0 && c4 >= 512 && (long long)((w + 13) / 14) * ((h + 13) / 14) * n < 128) ;
  const int rt = small ? 6 : 14;
  DSL_CHECK(!small || c4 % 64 == 0, "dsl_rla_tail_fwd: the K-split form needs c4 %% 64 == 0");
  RlaTailK k{(const uint16_t*)x, (const uint16_t*)h_in, (const uint16_t*)w_conv_out, bn_scale, bn_bias, (const uint16_t*)w_recurrent,
             (uint16_t*)u, (uint16_t*)t, (uint16_t*)h_out, ldx, ldh, c4, tw, ldo, n, h, w, (w + rt - 1) / rt, (h + rt - 1) / rt};
  const long long blocks = (long long)k.tiles_x * k.tiles_y * n;
  DSL_CHECK(blocks < 0x7fffffffLL, "dsl_rla_tail_fwd: grid too large");
  if (small)
    hipLaunchKernelGGL((rla_tail_fwd_kernel<6, 4>), dim3((unsigned)blocks), dim3(512), 0, (hipStream_t)stream, k);
  else
    hipLaunchKernelGGL((rla_tail_fwd_kernel<14, 1>), dim3((unsigned)blocks), dim3(512), 0, (hipStream_t)stream, k);
  DSL_LAUNCH_CHECK("rla_tail_fwd_kernel");
  return 0;
}

extern "C" int dsl_rla_op(const dsl_rla_desc* d, void* stream) {
  DSL_CHECK(d != nullptr, "dsl_rla_op: null descriptor");
  const int32_t* i = d->i;
  void* const* p = d->p;
  switch (d->kind) {
    case DSL_RLA_AVGPOOL: return dsl_avgpool2x2(p[0], i[0], p[1], i[1], i[2], i[3], i[4], i[5], stream);
    case DSL_RLA_AVGPOOL_BWD: return dsl_avgpool2x2_bwd(p[0], i[0], p[1], i[1], i[2], i[3], i[4], i[5], stream);
    case DSL_RLA_BN_TANH: return dsl_bn_tanh_fwd(p[0], i[0], (const float*)p[1], (const float*)p[2], p[3], i[1], (long)d->rows, i[2], stream);
    case DSL_RLA_BN_TANH_BWD:
      return dsl_bn_tanh_bwd(p[0], i[0], p[1], i[1], p[2], i[2], (const float*)p[3], (const float*)p[4], (const float*)p[5], d->f[0],
                             p[6], i[3], (float*)p[7], (float*)p[8], p[9], (long)d->rows, i[4], stream);
    case DSL_RLA_BN_FOLD:
      return dsl_bn_fold((const float*)p[0], (const float*)p[1], (const float*)p[2], (const float*)p[3], d->f[0], (float*)p[4],
                         (float*)p[5], i[0], stream);
    case DSL_RLA_BN_POST: return dsl_bn_wgrad_post((const dsl_bn_post_item*)p[0], i[0], i[1], d->f[0], stream);
    case DSL_RLA_REC_SUM: return dsl_rec_sum_multi((const dsl_rec_sum_item*)p[0], i[0], i[1], stream);
    case DSL_RLA_TAIL_BWD:
      return dsl_rla_tail_bwd(p[0], i[0], p[1], i[1], p[2], i[2], p[3], (const float*)p[4], (const float*)p[5], (const float*)p[6], d->f[0], p[7],
                              i[3], (float*)p[8], (float*)p[9], p[10], i[4], i[5], i[6], stream);
    case DSL_RLA_TAIL_FWD:
      return dsl_rla_tail_fwd(p[0], i[0], p[1], i[1], p[2], i[2], (const float*)p[3], (const float*)p[4], p[5], i[3], p[6], p[7], p[8], i[4],
                              i[5], i[6], i[7], stream);
    default: dsl_set_error("dsl_rla_op: unknown kind %d", d->kind); return -1;
  }
}
