// Memory-bound layers of the FCOS R50-FPN step on gfx950: image pack, max pool, GroupNorm+ReLU
// (forward / backward), FPN upsample backward, column sums.  All NHWC bf16, 16-byte accesses.
#include "common.hpp"

namespace {

// ---- NCHW fp32 -> NHWC8 bf16 -------------------------------------------------------------------
__global__ void pack_image_kernel(const float* __restrict__ img, uint16_t* __restrict__ out, int n,
                                  int h, int w) {
  const long long hw = (long long)h * w;
  const long long total = (long long)n * hw;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const long long b = i / hw, r = i - b * hw;
    const float* src = img + b * 3 * hw + r;
    u32x4 v = {pack2bf(src[0], src[hw]), pack2bf(src[2 * hw], 0.f), 0u, 0u};
    *reinterpret_cast<u32x4*>(out + i * 8) = v;
  }
}

// ---- 3x3 s2 p1 max pool (resnet.py:610) ---------------------------------------------------------
__device__ __forceinline__ uint32_t max2bf(uint32_t a, uint32_t b) {
  const float lo = fmaxf(bflo(a), bflo(b)), hi = fmaxf(bfhi(a), bfhi(b));
  return (__float_as_uint(lo) >> 16) | (__float_as_uint(hi) & 0xffff0000u);
}
__global__ void maxpool_kernel(const uint16_t* __restrict__ x, uint16_t* __restrict__ y, int n, int h,
                               int w, int c, int oh, int ow, int ldy) {
  const int c8 = c / 8;
  const long long total = (long long)n * oh * ow * c8;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int cc = (int)(i % c8);
    long long t = i / c8;
    const int ox = (int)(t % ow);
    t /= ow;
    const int oy = (int)(t % oh);
    const int b = (int)(t / oh);
    u32x4 m = {0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u};   // -inf pairs
    for (int dy = 0; dy < 3; ++dy) {
      const int iy = oy * 2 - 1 + dy;
      if ((unsigned)iy >= (unsigned)h) continue;
      for (int dx = 0; dx < 3; ++dx) {
        const int ix = ox * 2 - 1 + dx;
        if ((unsigned)ix >= (unsigned)w) continue;
        const u32x4 v = *reinterpret_cast<const u32x4*>(x + (((long long)b * h + iy) * w + ix) * c + cc * 8);
#pragma unroll
        for (int e = 0; e < 4; ++e) m[e] = max2bf(m[e], v[e]);
      }
    }
    *reinterpret_cast<u32x4*>(y + (i / c8) * ldy + cc * 8) = m;
  }
}

// ---- GroupNorm(32) + ReLU ------------------------------------------------------------------------
// Two passes each way over a level-major tensor, statistics per (level segment, image, group) in fp32.
// Every reduction is two-level and in a FIXED order (no atomics): pass 1 writes one partial record per
// 128-pixel block into the workspace, pass 2's prologue (and, for the per-channel sums of the backward, an
// extra row of workgroups) adds the records up in block order - results are bit-identical run to run, and
// nothing has to be zeroed beforehand.  (The round-1 kernels added block partials with float atomics: up to
// 350 workgroups x 512 same-address L2 atomics made the reduction passes 2x slower than the streaming passes.)
struct GnK {
  int nseg, n, c, groups, cpg8;   // cpg8 = 16-byte chunks per group (channels per group / 8)
  int h[DSL_MAX_SEG], w[DSL_MAX_SEG];
  long long off[DSL_MAX_SEG];    // pixel offset of segment start
  int nblk;                      // workgroups per (segment, image) row of the grid = ceil(max hw / GN_PPB)
  float eps;
  const uint16_t* x;
  uint16_t* y;
  const float* gamma;
  const float* beta;
  float* stats;
  const uint16_t* dy;
  uint16_t* dx;
  float* dgamma;
  float* dbeta;
  float* dbias;                  // optional: gradient of the bias of the convolution that produced x = sum_p dx
  float* ws;                     // partial records [nseg*n][nblk][rec]; rec = 2*groups (fwd) or 3*c + 2*groups (bwd)
  uint8_t* y8;                   // forward, optional: e4m3 copy of y with the delayed scale y8_scale[0]; y8_amax[block] = the block's max y
  const float* y8_scale;
  float* y8_amax;
  int conv_nbk;                  // > 0: the forward records were left by the producing convolution's epilogue (conv.hip
                                 // conv_tile_epilogue): [64 floats of header, word 0 = its pixel tile][nseg*n][conv_nbk][2*groups],
                                 // a row's records indexed by pixel tile relative to the tile that holds the row's first pixel
};

constexpr int GN_PPB = 128;   // pixels per block (round 5, profiles/r05_gn_ppa_ab.txt: apply passes with 64 / 128 / 256 pixels per block tie,
                              // 512: - 4 %, 1024: - 7 % on the step - the passes live in the gaps the convolutions' workgroups leave)

// Where the block records of one (segment, image) row are and how many there are: the pass's own (one per GN_PPB pixels) or the
// ones a convolution's epilogue left (conv_nbk > 0: one per pixel tile of that launch that meets the row).
struct GnRows {
  const float* base;
  int nb;
};
__device__ __forceinline__ GnRows gn_rows(const GnK& p, int si, int R) {
  const int seg = si / p.n, img = si - seg * p.n;
  const int hw = p.h[seg] * p.w[seg];
  GnRows r;
  if (p.conv_nbk > 0) {
    const int bpx = *reinterpret_cast<const int*>(p.ws);
    const long long rs = p.off[seg] + (long long)img * hw;
    r.nb = (int)((rs + hw - 1) / bpx - rs / bpx) + 1;
    r.base = p.ws + 64 + (long long)si * p.conv_nbk * R;
  } else {
    r.nb = (hw + GN_PPB - 1) / GN_PPB;
    r.base = p.ws + (long long)si * p.nblk * R;
  }
  return r;
}
// Block sizes: 256 threads.  Standalone the passes like big blocks (1024 / 512 threads: 17 + 32 us forward + backward against 25 + 39),
// but in the step every pass runs BESIDE a convolution of the other tower or image chain whose workgroups already hold 2 waves x
// ~190 VGPRs per SIMD and 112 KB of LDS on every CU: a 16-wave block (or gn_bwd_reduce's 53 KB of LDS at 512 threads) cannot
// become resident until such a workgroup retires, so the pass took as long as that convolution (57 us in the trace).  Four-wave
// blocks fit in the registers and LDS the convolution leaves: + 1.8 % on the training step (403 -> 411 img/s, same box).
constexpr int GN_TA = 256;    // the two pure streaming (apply) passes
constexpr int GN_T = 256;     // the two reduction passes

// Sum of V-float records over the nb blocks of one (segment, image), by all T threads of the workgroup, in a fixed
// order: thread (q, v) adds blocks q, q+Q, ... (Q = T / V), then thread v adds the Q partial sums in order.
// out[] is LDS (V floats); sh[] is LDS scratch (T floats).  Needs V <= T and T % V == 0.
template <int T>
__device__ __forceinline__ void gn_sum_records(const float* __restrict__ rec0, int nb, int stride, int V, float* sh,
                                               float* out) {
  const int Q = T / V;
  const int v = threadIdx.x % V, q = threadIdx.x / V;
  float a = 0.f;
  if (q < Q) {
#pragma unroll 8                    // eight records in flight (every block of the apply passes starts with this sum: its
    for (int b = q; b < nb; b += Q) a += rec0[(long long)b * stride + v];     // latency is a serial prelude of the pass)
  }
  sh[threadIdx.x] = a;
  __syncthreads();
  if (threadIdx.x < V) {
    float s = 0.f;
    for (int k = 0; k < Q; ++k) s += sh[k * V + threadIdx.x];
    out[threadIdx.x] = s;
  }
  __syncthreads();
}

// pass 1 of forward: sum / sumsq per (seg, img, group) of this block's pixels -> ws record
// (register budget: the pass must fit into what a convolution workgroup leaves free on its CU - 2 x ~190 of 512 VGPRs per SIMD are
// taken.  With both load batches unrolled the kernel took 142 VGPRs and ran only on CUs without a convolution workgroup: 35 - 41 us
// in the step for a 9 us pass.  Batches of four loads, not unrolled, keep it under that.)
__global__ __launch_bounds__(GN_T) void gn_stats_kernel(const GnK p) {
  __shared__ float sh[2][GN_T];
  const int si = blockIdx.y, seg = si / p.n, img = si - seg * p.n;
  const int hw = p.h[seg] * p.w[seg];
  const int px0 = blockIdx.x * GN_PPB;
  if (px0 >= hw) return;
  const int cpr = p.c / 8;                 // 16-byte chunks per pixel (32 for C=256)
  const int ppi = GN_T / cpr;               // pixels per iteration
  const int chunk = threadIdx.x % cpr, prow = threadIdx.x / cpr;
  const uint16_t* base = p.x + (p.off[seg] + (long long)img * hw) * p.c;
  float s = 0.f, ss = 0.f;
  const int px1 = min(px0 + GN_PPB, hw);
  if (px1 - px0 == GN_PPB && GN_PPB % (ppi * 4) == 0) {      // full block: four loads of the thread in flight at a time
#pragma nounroll
    for (int b0 = 0; b0 < GN_PPB; b0 += ppi * 4) {
      u32x4 v[4];
#pragma unroll
      for (int i = 0; i < 4; ++i)
        v[i] = *reinterpret_cast<const u32x4*>(base + (long long)(px0 + b0 + prow + i * ppi) * p.c + chunk * 8);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float a = bflo(v[i][e]), b = bfhi(v[i][e]);
          s += a + b;
          ss += a * a + b * b;
        }
    }
  } else {
#pragma unroll 4
    for (int px = px0 + prow; px < px1; px += ppi) {
      const u32x4 v = *reinterpret_cast<const u32x4*>(base + (long long)px * p.c + chunk * 8);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float a = bflo(v[e]), b = bfhi(v[e]);
        s += a + b;
        ss += a * a + b * b;
      }
    }
  }
  sh[0][threadIdx.x] = s;
  sh[1][threadIdx.x] = ss;
  __syncthreads();
  // threads [0, groups): reduce over the cpg8 chunks of the group and the ppi pixel rows
  if (threadIdx.x < p.groups) {
    float a = 0.f, b = 0.f;
#pragma nounroll
    for (int r = 0; r < ppi; ++r)
#pragma nounroll
      for (int k = 0; k < p.cpg8; ++k) {
        a += sh[0][r * cpr + threadIdx.x * p.cpg8 + k];
        b += sh[1][r * cpr + threadIdx.x * p.cpg8 + k];
      }
    float* dst = p.ws + ((long long)si * p.nblk + blockIdx.x) * (2 * p.groups) + 2 * threadIdx.x;
    dst[0] = a;
    dst[1] = b;
  }
}

// pass 2 of forward: y = relu((x-mean)*rstd*gamma+beta); also writes (mean, rstd) to stats
__global__ __launch_bounds__(GN_TA) void gn_apply_kernel(const GnK p) {
  __shared__ float sh[GN_TA];
  __shared__ float tot[512];
  const int si = blockIdx.y, seg = si / p.n, img = si - seg * p.n;
  const int hw = p.h[seg] * p.w[seg];
  const int px0 = blockIdx.x * GN_PPB;
  if (px0 >= hw) return;
  {
    const GnRows rr = gn_rows(p, si, 2 * p.groups);
    gn_sum_records<GN_TA>(rr.base, rr.nb, 2 * p.groups, 2 * p.groups, sh, tot);
  }
  const int cpr = p.c / 8, ppi = GN_TA / cpr;
  const int chunk = threadIdx.x % cpr, prow = threadIdx.x / cpr;
  const int grp = chunk / p.cpg8;
  const float cnt = (float)hw * (float)(p.c / p.groups);
  const float mean = tot[2 * grp] / cnt;
  const float var = fmaxf(tot[2 * grp + 1] / cnt - mean * mean, 0.f);
  const float rstd = rsqrtf(var + p.eps);
  if (blockIdx.x == 0 && prow == 0 && (chunk % p.cpg8) == 0) {
    float* st = p.stats + ((long long)si * p.groups + grp) * 2;
    st[0] = mean;
    st[1] = rstd;
  }
  float ga[8], be[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    ga[e] = p.gamma[chunk * 8 + e] * rstd;
    be[e] = p.beta[chunk * 8 + e] - mean * ga[e];
  }
  const long long ibase = (p.off[seg] + (long long)img * hw) * p.c;
  const int px1 = min(px0 + GN_PPB, hw);
  if (p.y8) {
    // + the fp8 copy the next (fp8) convolution reads, quantised HERE with the previous step's scale (delayed scaling), and this
    // block's maximum of the rounded outputs for the next step's: no quantisation pass, no absmax pass
    const float s8 = p.y8_scale[0];
    float m = 0.f;
#pragma unroll 4
    for (int px = px0 + prow; px < px1; px += ppi) {
      const long long o = ibase + (long long)px * p.c + chunk * 8;
      const u32x4 v = *reinterpret_cast<const u32x4*>(p.x + o);
      u32x4 r;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float a = fmaxf(bflo(v[e]) * ga[2 * e] + be[2 * e], 0.f);
        const float b = fmaxf(bfhi(v[e]) * ga[2 * e + 1] + be[2 * e + 1], 0.f);
        r[e] = pack2bf(a, b);
        m = fmaxf(m, fmaxf(bflo(r[e]), bfhi(r[e])));         // (of the ROUNDED values: what a pass over y would find)
      }
      *reinterpret_cast<u32x4*>(p.y + o) = r;
      uint2 q;
      q.x = cvt4_fp8(bflo(r[0]) * s8, bfhi(r[0]) * s8, bflo(r[1]) * s8, bfhi(r[1]) * s8);
      q.y = cvt4_fp8(bflo(r[2]) * s8, bfhi(r[2]) * s8, bflo(r[3]) * s8, bfhi(r[3]) * s8);
      *reinterpret_cast<uint2*>(p.y8 + o) = q;
    }
#pragma unroll
    for (int o_ = 32; o_ > 0; o_ >>= 1) m = fmaxf(m, __shfl_xor(m, o_, 64));
    __syncthreads();                                           // (sh: the record sums above are consumed)
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
      for (int w_ = 1; w_ < GN_TA / 64; ++w_) m = fmaxf(m, sh[w_]);
      p.y8_amax[(long long)si * p.nblk + blockIdx.x] = m;
    }
    return;
  }
#pragma unroll 8
  for (int px = px0 + prow; px < px1; px += ppi) {
    const long long o = ibase + (long long)px * p.c + chunk * 8;
    const u32x4 v = *reinterpret_cast<const u32x4*>(p.x + o);
    u32x4 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float a = fmaxf(bflo(v[e]) * ga[2 * e] + be[2 * e], 0.f);
      const float b = fmaxf(bfhi(v[e]) * ga[2 * e + 1] + be[2 * e + 1], 0.f);
      r[e] = pack2bf(a, b);
    }
    *reinterpret_cast<u32x4*>(p.y + o) = r;
  }
}

// backward pass 1, per block record [3c + 2*groups]: per channel dg = sum dz*xhat, db = sum dz, sx = sum xhat;
// per group s1 = sum dz*gamma, s2 = sum dz*gamma*xhat       (dz = dy * [gamma*xhat+beta > 0])
constexpr int GN_BW = 26;     // floats per thread in the block reduction (8 dg + 8 db + 8 sx + s1 + s2)
__global__ __launch_bounds__(GN_T) void gn_bwd_reduce_kernel(const GnK p) {
  __shared__ float sh[GN_T * GN_BW];
  const int si = blockIdx.y, seg = si / p.n, img = si - seg * p.n;
  const int hw = p.h[seg] * p.w[seg];
  const int px0 = blockIdx.x * GN_PPB;
  if (px0 >= hw) return;
  const int cpr = p.c / 8, ppi = GN_T / cpr;
  const int chunk = threadIdx.x % cpr, prow = threadIdx.x / cpr;
  const int grp = chunk / p.cpg8;
  const float* st = p.stats + ((long long)si * p.groups + grp) * 2;
  const float mean = st[0], rstd = st[1];
  float ga[8], be[8], dg[8], db[8], sx[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    ga[e] = p.gamma[chunk * 8 + e];
    be[e] = p.beta[chunk * 8 + e];
    dg[e] = 0.f;
    db[e] = 0.f;
    sx[e] = 0.f;
  }
  float s1 = 0.f, s2 = 0.f;
  const long long ibase = (p.off[seg] + (long long)img * hw) * p.c;
  const int px1 = min(px0 + GN_PPB, hw);
#pragma unroll 8
  for (int px = px0 + prow; px < px1; px += ppi) {
    const long long o = ibase + (long long)px * p.c + chunk * 8;
    const u32x4 xv = *reinterpret_cast<const u32x4*>(p.x + o);
    const u32x4 gv = *reinterpret_cast<const u32x4*>(p.dy + o);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float xx = (e & 1) ? bfhi(xv[e >> 1]) : bflo(xv[e >> 1]);
      const float gg = (e & 1) ? bfhi(gv[e >> 1]) : bflo(gv[e >> 1]);
      const float xh = (xx - mean) * rstd;
      const float dz = (xh * ga[e] + be[e] > 0.f) ? gg : 0.f;
      dg[e] += dz * xh;
      db[e] += dz;
      sx[e] += xh;
      s1 += dz * ga[e];
      s2 += dz * ga[e] * xh;
    }
  }
  float* my = sh + threadIdx.x * GN_BW;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    my[e] = dg[e];
    my[8 + e] = db[e];
    my[16 + e] = sx[e];
  }
  my[24] = s1;
  my[25] = s2;
  __syncthreads();
  float* rec = p.ws + ((long long)si * p.nblk + blockIdx.x) * (3 * p.c + 2 * p.groups);
  // channel sums: thread t < c handles channel t
  for (int ch = threadIdx.x; ch < p.c; ch += GN_T) {
    const int ck = ch / 8, e = ch % 8;
    float a = 0.f, b = 0.f, s = 0.f;
    for (int r = 0; r < ppi; ++r) {
      const float* t = sh + (r * cpr + ck) * GN_BW;
      a += t[e];
      b += t[8 + e];
      s += t[16 + e];
    }
    rec[ch] = a;
    rec[p.c + ch] = b;
    rec[2 * p.c + ch] = s;
  }
  if (threadIdx.x < p.groups) {
    float a = 0.f, b = 0.f;
    for (int r = 0; r < ppi; ++r)
      for (int k = 0; k < p.cpg8; ++k) {
        const float* t = sh + (r * cpr + threadIdx.x * p.cpg8 + k) * GN_BW;
        a += t[24];
        b += t[25];
      }
    rec[3 * p.c + 2 * threadIdx.x] = a;
    rec[3 * p.c + 2 * threadIdx.x + 1] = b;
  }
}

// backward pass 2: dx = rstd * (dz*gamma - (s1 + xhat*s2)/cnt).
// Grid row blockIdx.y == 0 is the parameter-gradient row (dispatched first, it overlaps the streaming rows): its
// workgroups add the per-channel records of ALL (segment, image, block) up in a fixed order, GN_CW channels per pass:
//   dgamma[c] = sum dg,  dbeta[c] = sum db,
//   dbias[c]  = sum_p dx[p][c] = sum_si rstd_si * (gamma_c * DB_si[c] - hw_si * m1_si - m2_si * SX_si[c])
// (the gradient of the bias of the convolution in front of the norm, from the records instead of one more pass
// over dx).  Rows 1 .. nseg*n are the streaming rows.
constexpr int GN_CW = 16;     // channels per pass of a parameter-gradient workgroup
__global__ __launch_bounds__(GN_TA) void gn_bwd_apply_kernel(const GnK p) {
  __shared__ float sh[GN_TA * 3];
  __shared__ float tot[1024];
  const int S = p.nseg * p.n;
  const int R = 3 * p.c + 2 * p.groups;
  if (blockIdx.y == 0) {
    constexpr int Q = GN_TA / GN_CW;       // 64 record lanes per channel
    const int cl = threadIdx.x % GN_CW, q = threadIdx.x / GN_CW;
    const int cpg = p.c / p.groups;
    for (int c0 = blockIdx.x * GN_CW; c0 < p.c; c0 += gridDim.x * GN_CW) {
      __syncthreads();
      // phase 1: the (s1, s2) group sums of every (segment, image) for the (at most two) groups of this channel slice
      const int g_lo = c0 / cpg;
      const int NG = (min(c0 + GN_CW, p.c) - 1) / cpg - g_lo + 1;
      const int NV = S * NG * 2, Q1 = GN_TA / NV;
      {
        const int v = threadIdx.x % NV, j = threadIdx.x / NV;
        float a = 0.f;
        if (j < Q1) {
          const int si = v / (2 * NG), k = v - si * (2 * NG);
          const GnRows rr = gn_rows(p, si, R);
          const float* col = rr.base + 3 * p.c + 2 * g_lo + k;
#pragma unroll 8
          for (int b = j; b < rr.nb; b += Q1) a += col[(long long)b * R];
        }
        sh[threadIdx.x] = a;
        __syncthreads();
        if (threadIdx.x < NV) {
          float t = 0.f;
          for (int k = 0; k < Q1; ++k) t += sh[k * NV + threadIdx.x];
          tot[threadIdx.x] = t;              // [si][group - g_lo][s1, s2]
        }
        __syncthreads();
      }
      // phase 2: the per-channel records
      const int ch = c0 + cl;
      const bool ok = ch < p.c;
      const int gi = ok ? ch / cpg - g_lo : 0;
      const float gam = ok ? p.gamma[ch] : 0.f;
      float dg = 0.f, db = 0.f, dbi = 0.f;
      for (int si = 0; si < S; ++si) {
        const int seg = si / p.n;
        const int hw = p.h[seg] * p.w[seg];
        const GnRows rr = gn_rows(p, si, R);
        const float rstd = p.stats[((long long)si * p.groups + g_lo + gi) * 2 + 1];
        const float m2 = tot[(si * NG + gi) * 2 + 1] / ((float)hw * (float)cpg);
        if (ok) {
#pragma unroll 4
          for (int b = q; b < rr.nb; b += Q) {
            const float* rec = rr.base + (long long)b * R;
            const float rdb = rec[p.c + ch];
            dg += rec[ch];
            db += rdb;
            dbi += rstd * (gam * rdb - m2 * rec[2 * p.c + ch]);
          }
        }
      }
      sh[(0 * Q + q) * GN_CW + cl] = dg;
      sh[(1 * Q + q) * GN_CW + cl] = db;
      sh[(2 * Q + q) * GN_CW + cl] = dbi;
      __syncthreads();
      if (threadIdx.x < GN_CW && ok) {
        float t[3] = {0.f, 0.f, 0.f};
        for (int k = 0; k < 3; ++k)
          for (int j = 0; j < Q; ++j) t[k] += sh[(k * Q + j) * GN_CW + cl];
        for (int si = 0; si < S; ++si) {       // - rstd * hw * m1, m1 = s1 / (hw * cpg)
          const float rstd = p.stats[((long long)si * p.groups + g_lo + gi) * 2 + 1];
          t[2] -= rstd * tot[(si * NG + gi) * 2] / (float)cpg;
        }
        p.dgamma[ch] = t[0];
        p.dbeta[ch] = t[1];
        if (p.dbias) p.dbias[ch] = t[2];
      }
    }
    return;
  }
  const int si = blockIdx.y - 1, seg = si / p.n, img = si - seg * p.n;
  const int hw = p.h[seg] * p.w[seg];
  const int px0 = blockIdx.x * GN_PPB;
  if (px0 >= hw) return;
  const GnRows rr = gn_rows(p, si, R);
  gn_sum_records<GN_TA>(rr.base + 3 * p.c, rr.nb, R, 2 * p.groups, sh, tot);
  const int cpr = p.c / 8, ppi = GN_TA / cpr;
  const int chunk = threadIdx.x % cpr, prow = threadIdx.x / cpr;
  const int grp = chunk / p.cpg8;
  const float* st = p.stats + ((long long)si * p.groups + grp) * 2;
  const float mean = st[0], rstd = st[1];
  const float cnt = (float)hw * (float)(p.c / p.groups);
  const float m1 = tot[2 * grp] / cnt, m2 = tot[2 * grp + 1] / cnt;
  float ga[8], be[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    ga[e] = p.gamma[chunk * 8 + e];
    be[e] = p.beta[chunk * 8 + e];
  }
  const long long ibase = (p.off[seg] + (long long)img * hw) * p.c;
  const int px1 = min(px0 + GN_PPB, hw);
#pragma unroll 8
  for (int px = px0 + prow; px < px1; px += ppi) {
    const long long o = ibase + (long long)px * p.c + chunk * 8;
    const u32x4 xv = *reinterpret_cast<const u32x4*>(p.x + o);
    const u32x4 gv = *reinterpret_cast<const u32x4*>(p.dy + o);
    float r[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float xx = (e & 1) ? bfhi(xv[e >> 1]) : bflo(xv[e >> 1]);
      const float gg = (e & 1) ? bfhi(gv[e >> 1]) : bflo(gv[e >> 1]);
      const float xh = (xx - mean) * rstd;
      const float dz = (xh * ga[e] + be[e] > 0.f) ? gg : 0.f;
      r[e] = rstd * (dz * ga[e] - m1 - xh * m2);
    }
    u32x4 ov = {pack2bf(r[0], r[1]), pack2bf(r[2], r[3]), pack2bf(r[4], r[5]), pack2bf(r[6], r[7])};
    *reinterpret_cast<u32x4*>(p.dx + o) = ov;
  }
}

// ---- backward of nearest upsample: out(h,w) = sum of children in g(ch,cw) ----------------------
__global__ void sum_children_kernel(const uint16_t* __restrict__ g, uint16_t* __restrict__ out, int n,
                                    int h, int w, int ch, int cw, int c) {
  const int c8 = c / 8;
  const long long total = (long long)n * h * w * c8;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int cc = (int)(i % c8);
    long long t = i / c8;
    const int x = (int)(t % w);
    t /= w;
    const int y = (int)(t % h);
    const int b = (int)(t / h);
    // child (cy,cx) reads parent (cy*h/ch, cx*w/cw): children of y are cy in [ceil(y*ch/h), ceil((y+1)*ch/h))
    const int cy0 = (y * ch + h - 1) / h, cy1 = ((y + 1) * ch + h - 1) / h;
    const int cx0 = (x * cw + w - 1) / w, cx1 = ((x + 1) * cw + w - 1) / w;
    float a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int cy = cy0; cy < cy1 && cy < ch; ++cy)
      for (int cx = cx0; cx < cx1 && cx < cw; ++cx) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(g + (((long long)b * ch + cy) * cw + cx) * c + cc * 8);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          a[2 * e] += bflo(v[e]);
          a[2 * e + 1] += bfhi(v[e]);
        }
      }
    u32x4 o = {pack2bf(a[0], a[1]), pack2bf(a[2], a[3]), pack2bf(a[4], a[5]), pack2bf(a[6], a[7])};
    *reinterpret_cast<u32x4*>(out + i * 8) = o;
  }
}

// ---- column sums of a bf16 [rows][ld] matrix ---------------------------------------------------
constexpr int CS_RPB = 128;
constexpr int CS_T = 512;      // 8 waves per block (see GN_T)
__global__ __launch_bounds__(CS_T) void colsum_kernel(const uint16_t* __restrict__ x, float* __restrict__ out,
                                                      long long rows, int c, int ld) {
  __shared__ float sh[CS_T * 8];
  const int cpr = (c + 7) / 8;                 // chunks per row actually needed
  const int ppi = CS_T / cpr;
  const int chunk = threadIdx.x % cpr, prow = threadIdx.x / cpr;
  float a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const long long r0 = (long long)blockIdx.x * CS_RPB, r1 = min(r0 + CS_RPB, rows);
  if (prow < ppi)
    for (long long r = r0 + prow; r < r1; r += ppi) {
      const u32x4 v = *reinterpret_cast<const u32x4*>(x + r * ld + chunk * 8);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        a[2 * e] += bflo(v[e]);
        a[2 * e + 1] += bfhi(v[e]);
      }
    }
#pragma unroll
  for (int e = 0; e < 8; ++e) sh[threadIdx.x * 8 + e] = (prow < ppi) ? a[e] : 0.f;
  __syncthreads();
  for (int ch = threadIdx.x; ch < c; ch += CS_T) {
    float s = 0.f;
    for (int r = 0; r < ppi; ++r) s += sh[(r * cpr + ch / 8) * 8 + (ch % 8)];
    atomicAdd(out + ch, s);
  }
}

int fill_gn(const dsl_gn_desc* d, GnK& k, long long* total_px) {
  memset(&k, 0, sizeof(k));
  k.nseg = d->nseg; k.n = d->n; k.c = d->c; k.groups = d->groups;
  k.cpg8 = d->c / d->groups / 8;
  long long off = 0;
  for (int s = 0; s < d->nseg; ++s) {
    k.h[s] = d->h[s]; k.w[s] = d->w[s]; k.off[s] = off;
    off += (long long)d->n * d->h[s] * d->w[s];
  }
  *total_px = off;
  k.eps = d->eps;
  k.x = (const uint16_t*)d->x; k.y = (uint16_t*)d->y; k.gamma = d->gamma; k.beta = d->beta;
  k.stats = d->stats; k.dy = (const uint16_t*)d->dy; k.dx = (uint16_t*)d->dx;
  k.dgamma = d->dgamma; k.dbeta = d->dbeta; k.dbias = d->dbias; k.ws = (float*)d->workspace;
  k.y8 = (uint8_t*)d->y8; k.y8_scale = d->y8_scale; k.y8_amax = d->y8_amax;
  int maxhw = 0;
  for (int s = 0; s < d->nseg; ++s) maxhw = max(maxhw, d->h[s] * d->w[s]);
  k.nblk = (maxhw + GN_PPB - 1) / GN_PPB;
  k.conv_nbk = 0;
  return 0;
}

}  // namespace
extern "C" size_t dsl_groupnorm_workspace_bytes(const dsl_gn_desc* d) {
  if (!d || d->nseg < 1 || d->nseg > DSL_MAX_SEG) return 0;
  int maxhw = 0;
  for (int s = 0; s < d->nseg; ++s) maxhw = max(maxhw, d->h[s] * d->w[s]);
  const size_t nblk = (maxhw + GN_PPB - 1) / GN_PPB;
  const size_t own = (size_t)d->nseg * d->n * nblk * (3 * (size_t)d->c + 2 * (size_t)d->groups) * sizeof(float);
  const size_t conv = 64 * sizeof(float) +
                      (size_t)d->nseg * d->n * (maxhw / 64 + 2) * (3 * (size_t)d->c + 2 * (size_t)d->groups) * sizeof(float);
  return own > conv ? own : conv;       // (the records of a producing convolution, dsl_conv_desc.gn_ws, fit as well)
}
namespace {
int gn_check(const dsl_gn_desc* d, const char* who) {
  DSL_CHECK(d && d->nseg >= 1 && d->nseg <= DSL_MAX_SEG, "%s: bad descriptor", who);
  DSL_CHECK(d->c % 8 == 0 && d->c <= GN_T * 8 && GN_T % (d->c / 8) == 0, "%s: unsupported C=%d", who, d->c);
  DSL_CHECK(d->groups > 0 && d->c % d->groups == 0 && (d->c / d->groups) % 8 == 0 && d->groups <= 256,
            "%s: channels per group must be a multiple of 8 (C=%d groups=%d)", who, d->c, d->groups);
  DSL_CHECK(GN_TA % (2 * d->groups) == 0, "%s: 2*groups=%d must divide %d", who, 2 * d->groups, GN_TA);
  DSL_CHECK(d->nseg * d->n * 4 <= 1024, "%s: more than 256 (segment, image) pairs", who);
  DSL_CHECK(d->workspace && d->workspace_bytes >= dsl_groupnorm_workspace_bytes(d), "%s: workspace too small (%zu < %zu)",
            who, d->workspace_bytes, dsl_groupnorm_workspace_bytes(d));
  return 0;
}

}  // namespace

extern "C" int dsl_pack_image(const float* img, void* out, int n, int h, int w, void* stream) {
  DSL_CHECK(img && out && n > 0 && h > 0 && w > 0, "dsl_pack_image: bad arguments");
  const long long total = (long long)n * h * w;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(pack_image_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, img, (uint16_t*)out, n, h, w);
  DSL_LAUNCH_CHECK("pack_image_kernel");
  return 0;
}

extern "C" int dsl_maxpool3x3s2_ld(const void* x, void* y, int n, int h, int w, int c, int ldy, void* stream);
extern "C" int dsl_maxpool3x3s2(const void* x, void* y, int n, int h, int w, int c, void* stream) {
  return dsl_maxpool3x3s2_ld(x, y, n, h, w, c, c, stream);
}
extern "C" int dsl_maxpool3x3s2_ld(const void* x, void* y, int n, int h, int w, int c, int ldy, void* stream) {
  DSL_CHECK(x && y && c % 8 == 0 && ldy >= c && ldy % 8 == 0, "dsl_maxpool3x3s2: bad arguments (C=%d ldy=%d)", c, ldy);
  const int oh = (h + 2 - 3) / 2 + 1, ow = (w + 2 - 3) / 2 + 1;
  const long long total = (long long)n * oh * ow * (c / 8);
  int blocks = (int)((total + 255) / 256);
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(maxpool_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)x,
                     (uint16_t*)y, n, h, w, c, oh, ow, ldy);
  DSL_LAUNCH_CHECK("maxpool_kernel");
  return 0;
}

extern "C" int dsl_groupnorm_relu_fwd(const dsl_gn_desc* d, void* stream) {
  if (gn_check(d, "dsl_groupnorm_relu_fwd")) return -1;
  DSL_CHECK(d->x && d->y && d->gamma && d->beta && d->stats, "dsl_groupnorm_relu_fwd: null pointer");
  GnK k;
  long long tot;
  fill_gn(d, k, &tot);
  int maxhw = 0;
  for (int s = 0; s < d->nseg; ++s) maxhw = max(maxhw, d->h[s] * d->w[s]);
  dim3 grid((maxhw + GN_PPB - 1) / GN_PPB, d->nseg * d->n);
  hipStream_t st = (hipStream_t)stream;
  DSL_CHECK(!d->y8 || (d->y8_scale && d->y8_amax), "dsl_groupnorm_relu_fwd: the fp8 copy needs y8_scale and y8_amax");
  if (d->conv_stats) {
    DSL_CHECK(d->c / d->groups == 8, "dsl_groupnorm_relu_fwd: conv_stats needs 8 channels per group (C=%d groups=%d)", d->c, d->groups);
    k.conv_nbk = maxhw / 64 + 2;
  } else {
    hipLaunchKernelGGL(gn_stats_kernel, grid, dim3(GN_T), 0, st, k);
  }
  hipLaunchKernelGGL(gn_apply_kernel, grid, dim3(GN_TA), 0, st, k);
  DSL_LAUNCH_CHECK("gn forward");
  return 0;
}

extern "C" int dsl_groupnorm_relu_bwd(const dsl_gn_desc* d, void* stream) {
  if (gn_check(d, "dsl_groupnorm_relu_bwd")) return -1;
  DSL_CHECK(d->x && d->dy && d->dx && d->gamma && d->beta && d->stats && d->dgamma && d->dbeta,
            "dsl_groupnorm_relu_bwd: null pointer");
  GnK k;
  long long tot;
  fill_gn(d, k, &tot);
  int maxhw = 0;
  for (int s = 0; s < d->nseg; ++s) maxhw = max(maxhw, d->h[s] * d->w[s]);
  dim3 grid((maxhw + GN_PPB - 1) / GN_PPB, d->nseg * d->n);
  hipStream_t st = (hipStream_t)stream;
  if (d->conv_stats) {
    DSL_CHECK(d->c / d->groups == 8, "dsl_groupnorm_relu_bwd: conv_stats needs 8 channels per group (C=%d groups=%d)", d->c, d->groups);
    k.conv_nbk = maxhw / 64 + 2;
  } else {
    hipLaunchKernelGGL(gn_bwd_reduce_kernel, grid, dim3(GN_T), 0, st, k);
  }
  dim3 grid2(grid.x, grid.y + 1);        // + the parameter-gradient row
  hipLaunchKernelGGL(gn_bwd_apply_kernel, grid2, dim3(GN_TA), 0, st, k);
  DSL_LAUNCH_CHECK("gn backward");
  return 0;
}

extern "C" int dsl_sum2x2(const void* g, void* out, int n, int h, int w, int ch, int cw, int c, void* stream) {
  DSL_CHECK(g && out && c % 8 == 0, "dsl_sum2x2: bad arguments");
  const long long total = (long long)n * h * w * (c / 8);
  int blocks = (int)((total + 255) / 256);
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(sum_children_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)g,
                     (uint16_t*)out, n, h, w, ch, cw, c);
  DSL_LAUNCH_CHECK("sum_children_kernel");
  return 0;
}

// out += column sums (the caller cleared out): used behind the weight-gradient reduce kernel, which clears db
int dsl_colsum_acc(const void* x, float* out, long rows, int c, int ld, void* stream) {
  DSL_CHECK(x && out && c > 0 && c <= 2048 && ld % 8 == 0 && ld >= c, "dsl_colsum: bad arguments c=%d ld=%d", c, ld);
  DSL_CHECK((c + 7) / 8 <= CS_T, "dsl_colsum: too many channels");
  const int blocks = (int)((rows + CS_RPB - 1) / CS_RPB);
  if (blocks > 0)
    hipLaunchKernelGGL(colsum_kernel, dim3(blocks), dim3(CS_T), 0, (hipStream_t)stream, (const uint16_t*)x, out, (long long)rows, c, ld);
  DSL_LAUNCH_CHECK("colsum_kernel");
  return 0;
}

extern "C" int dsl_colsum(const void* x, float* out, long rows, int c, int ld, void* stream) {
  DSL_CHECK(x && out && c > 0 && c <= 2048 && ld % 8 == 0 && ld >= c, "dsl_colsum: bad arguments c=%d ld=%d", c, ld);
  DSL_CHECK((c + 7) / 8 <= CS_T, "dsl_colsum: too many channels");
  hipStream_t st = (hipStream_t)stream;
  hipMemsetAsync(out, 0, sizeof(float) * c, st);
  const int blocks = (int)((rows + CS_RPB - 1) / CS_RPB);
  if (blocks > 0)
    hipLaunchKernelGGL(colsum_kernel, dim3(blocks), dim3(CS_T), 0, st, (const uint16_t*)x, out, (long long)rows, c, ld);
  DSL_LAUNCH_CHECK("colsum_kernel");
  return 0;
}
