// FCOS per-location target assignment and the fused focal + GIoU + centerness (+ DSL scale-invariant
// soft) loss, forward and backward in one pass, fp32.
//
// Restates mmdet/models/dense_heads/fcos_head.py:170-338 (loss), :550-560 (points), :562-705
// (targets), :707-726 (centerness target); mmdet/models/losses/focal_loss.py:11-56;
// mmdet/models/losses/iou_loss.py:85-102,329-366 with core/bbox/iou_calculators/iou2d_calculator.py:212-260;
// mmdet/models/losses/cross_entropy_loss.py:73-112 of the reference.
//
// The assignment must be BIT-identical to the reference's fp32 torch arithmetic, so this file is
// compiled with floating-point contraction off and the comparisons are written exactly as there.
#include "common.hpp"
#pragma clang fp contract(off)

namespace {

constexpr float kINF = 1e8f;   // fcos_head.py:11

struct FcK {
  int nlvl, n, num_classes, n_labeled;
  int h[DSL_MAX_SEG], w[DSL_MAX_SEG], stride[DSL_MAX_SEG];
  int mstart[DSL_MAX_SEG + 1];
  float lo[DSL_MAX_SEG], hi[DSL_MAX_SEG];
  float radius, loss_weight, soft_weight, grad_scale, inv_world;
  const float* gt_boxes; const long long* gt_labels; const int* gt_off;
  const float* ig_boxes; const int* ig_off;
  long long* labels; float* bbox_targets; int* assign_idx; float* cls_weight; float* pos_weight;
  float* stats;
  const float* cls_logits; const float* regctr; int ld_cls, ld_rc;
  const float* scales; const float* norm;
  uint16_t* g_cls; int ld_gcls; uint16_t* g_rc; int ld_grc;
  float* g_scales; float* losses;
  float* part;            // block records of the loss / assignment sums (fixed-order second pass, no float atomics)
};

__device__ __forceinline__ void decode_loc(const FcK& p, int m, int& lvl, int& img, int& y, int& x) {
  lvl = 0;
#pragma unroll
  for (int s = 1; s < DSL_MAX_SEG; ++s)
    if (s < p.nlvl && m >= p.mstart[s]) lvl = s;
  const int q = m - p.mstart[lvl];
  const int hw = p.h[lvl] * p.w[lvl];
  img = q / hw;
  const int r = q - img * hw;
  y = r / p.w[lvl];
  x = r - y * p.w[lvl];
}

// one image, one location against a list of boxes: area-argmin with centre sampling + range test
// (fcos_head.py:632-700).  Returns min_area; idx = first argmin; ltrb of that box in t[4].
__device__ __forceinline__ float assign_one(float px, float py, float rs, float lo, float hi,
                                            const float* __restrict__ boxes, int g0, int g1, int& idx,
                                            float t[4]) {
  float best = kINF;
  idx = 0;
  t[0] = t[1] = t[2] = t[3] = 0.f;
  for (int g = g0; g < g1; ++g) {
    const float x1 = boxes[4 * g], y1 = boxes[4 * g + 1], x2 = boxes[4 * g + 2], y2 = boxes[4 * g + 3];
    float area = (x2 - x1) * (y2 - y1);
    const float l = px - x1, tp = py - y1, r = x2 - px, b = y2 - py;
    const float cx = (x1 + x2) / 2, cy = (y1 + y2) / 2;
    const float xmin = cx - rs, ymin = cy - rs, xmax = cx + rs, ymax = cy + rs;
    const float c0 = xmin > x1 ? xmin : x1;
    const float c1 = ymin > y1 ? ymin : y1;
    const float c2 = xmax > x2 ? x2 : xmax;
    const float c3 = ymax > y2 ? y2 : ymax;
    const float cmin = fminf(fminf(px - c0, py - c1), fminf(c2 - px, c3 - py));
    const bool inside = cmin > 0.f;
    const float mx = fmaxf(fmaxf(l, tp), fmaxf(r, b));
    const bool in_range = (mx >= lo) && (mx <= hi);
    if (!inside || !in_range) area = kINF;
    if (g == g0 || area < best) {   // first index wins ties (torch.min over dim)
      best = area;
      idx = g - g0;
      t[0] = l; t[1] = tp; t[2] = r; t[3] = b;
    }
  }
  return best;
}

__global__ __launch_bounds__(256) void assign_kernel(const FcK p) {
  __shared__ float sh[16];
  const int M = p.mstart[p.nlvl];
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  float is_pos = 0.f, ctr = 0.f;
  if (m < M) {
    int lvl, img, y, x;
    decode_loc(p, m, lvl, img, y, x);
    const int s = p.stride[lvl];
    const float px = (float)x * (float)s + (float)(s / 2);
    const float py = (float)y * (float)s + (float)(s / 2);
    const float rs = (float)s * p.radius;
    int idx;
    float t[4];
    const int g0 = p.gt_off[img], g1 = p.gt_off[img + 1];
    long long label = p.num_classes;
    int aidx = -1;
    if (g1 > g0) {
      const float best = assign_one(px, py, rs, p.lo[lvl], p.hi[lvl], p.gt_boxes, g0, g1, idx, t);
      if (best != kINF) {
        label = p.gt_labels[g0 + idx];
        aidx = idx;
      }
    } else {
      t[0] = t[1] = t[2] = t[3] = 0.f;
    }
    const float fs = (float)s;
    const float n0 = t[0] / fs, n1 = t[1] / fs, n2 = t[2] / fs, n3 = t[3] / fs;   // norm_on_bbox
    p.labels[m] = label;
    p.assign_idx[m] = aidx;
    *reinterpret_cast<f32x4*>(p.bbox_targets + 4ll * m) = f32x4{n0, n1, n2, n3};
    // ignore weight: 0 iff the location falls in an ignore box (same procedure) and is background
    float wgt = 1.f;
    if (p.ig_boxes && p.ig_off) {
      const int i0 = p.ig_off[img], i1 = p.ig_off[img + 1];
      if (i1 > i0) {
        int ii;
        float tt[4];
        const float bi = assign_one(px, py, rs, p.lo[lvl], p.hi[lvl], p.ig_boxes, i0, i1, ii, tt);
        if (bi != kINF && label == p.num_classes) wgt = 0.f;
      }
    }
    const float sw = (p.loss_weight != 1.0f && img >= p.n_labeled) ? p.loss_weight : 1.f;
    p.cls_weight[m] = wgt * sw;
    p.pos_weight[m] = sw;
    if (label < p.num_classes) {
      is_pos = 1.f;
      const float lr_min = fminf(n0, n2), lr_max = fmaxf(n0, n2);
      const float tb_min = fminf(n1, n3), tb_max = fmaxf(n1, n3);
      ctr = sqrtf((lr_min / lr_max) * (tb_min / tb_max));
    }
  }
  const float np = block_sum(is_pos, sh);
  const float cs = block_sum(ctr, sh);
  if (threadIdx.x == 0) {
    p.part[2 * blockIdx.x] = np;
    p.part[2 * blockIdx.x + 1] = cs;
  }
}

// fixed-order sum of per-block records: out[v] = sum_b part[b * V + v] * scale(v); one workgroup
constexpr int FIN_T = 256;
__global__ __launch_bounds__(FIN_T) void fcos_finalize_kernel(const float* __restrict__ part, int nblocks, int V, float* out0,
                                                              int n0, float* out1, const float* norm, float inv_world,
                                                              float soft_weight, int mode, float* logvec = nullptr) {
  __shared__ float sh[FIN_T];
  // thread (q, v): blocks q, q + Q, ... ; then thread v adds the Q partial sums in order
  const int Q = FIN_T / V;
  const int v = threadIdx.x % V, q = threadIdx.x / V;
  float a = 0.f;
  if (q < Q) {
#pragma unroll 8                    // (the loads of eight records in flight; the additions stay in record order)
    for (int b = q; b < nblocks; b += Q) a += part[(long long)b * V + v];
  }
  sh[threadIdx.x] = a;
  __syncthreads();
  if (threadIdx.x < V) {
    float t = 0.f;
    for (int k = 0; k < Q; ++k) t += sh[k * V + threadIdx.x];
    if (mode == 1) {           // loss records: [cls, sisoft, bbox, centerness, g_scale x 5] -> losses[4] + g_scales[5]
      const float num_pos = fmaxf(norm[0] * inv_world, 1.0f), denorm = fmaxf(norm[1] * inv_world, 1e-6f);
      float fv = t;
      if (threadIdx.x == 0) out0[0] = fv = t / num_pos;
      else if (threadIdx.x == 1) out0[3] = fv = t * soft_weight;
      else if (threadIdx.x == 2) out0[1] = fv = t / denorm;
      else if (threadIdx.x == 3) out0[2] = fv = t / num_pos;
      else if (threadIdx.x - 4 < n0) out1[threadIdx.x - 4] = t;
      if (threadIdx.x < 4) sh[threadIdx.x] = fv;          // (slot x of sh is read by thread x only, above)
    } else if (threadIdx.x < n0) {
      out0[threadIdx.x] = t;
    }
  }
  if (mode == 1 && logvec) {       // the log vector of _parse_losses (detectors/base.py:175-208): the loss terms and their sum
    __syncthreads();
    if (threadIdx.x == 0) {
      const float cls = sh[0], soft = sh[1], bbox = sh[2], ctr = sh[3];
      float tot = (cls + bbox) + ctr;
      int n = 3;
      logvec[0] = cls; logvec[1] = bbox; logvec[2] = ctr;
      if (soft_weight != 0.f) { logvec[n++] = soft; tot += soft; }
      logvec[n] = tot;
    }
  }
  if (mode == 0 && (int)threadIdx.x >= V && threadIdx.x < 8) out0[threadIdx.x] = 0.f;      // stats[2..7] are reserved, kept zero
}

__global__ void points_kernel(const FcK p, float* __restrict__ pts) {
  int P = 0;
  for (int l = 0; l < p.nlvl; ++l) P += p.h[l] * p.w[l];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P) return;
  int l = 0, q = i;
  while (q >= p.h[l] * p.w[l]) {
    q -= p.h[l] * p.w[l];
    ++l;
  }
  const int y = q / p.w[l], x = q - y * p.w[l], s = p.stride[l];
  pts[2 * i] = (float)x * (float)s + (float)(s / 2);
  pts[2 * i + 1] = (float)y * (float)s + (float)(s / 2);
}

// ---- focal helpers: loss and dloss/dx for target 0/1 (alpha .25, gamma 2) ----------------------
__device__ __forceinline__ void focal_fb(float x, bool t, float& loss, float& grad) {
  const float alpha = 0.25f;
  const float e = __expf(-fabsf(x));
  const float l1p = log1pf(e);
  const float inv = 1.f / (1.f + e);
  const float pp = x >= 0.f ? inv : e * inv;   // sigmoid(x)
  const float q = 1.f - pp;
  if (t) {
    const float sp = fmaxf(-x, 0.f) + l1p;      // -log p
    loss = alpha * q * q * sp;
    grad = -alpha * q * q * (2.f * pp * sp + q);
  } else {
    const float sp = fmaxf(x, 0.f) + l1p;       // -log(1-p)
    loss = (1.f - alpha) * pp * pp * sp;
    grad = (1.f - alpha) * pp * pp * (2.f * q * sp + pp);
  }
}

// grad of max(a,b) w.r.t. a (ties split evenly, as torch.maximum) and of min(a,b) w.r.t. a
__device__ __forceinline__ float dmax_a(float a, float b) { return a > b ? 1.f : (a == b ? 0.5f : 0.f); }
__device__ __forceinline__ float dmin_a(float a, float b) { return a < b ? 1.f : (a == b ? 0.5f : 0.f); }

__global__ __launch_bounds__(256) void loss_kernel(const FcK p) {
  __shared__ float sh[16];
  const int M = p.mstart[p.nlvl];
  const int C = p.num_classes;
  const int c4 = C / 4;
  const float num_pos = fmaxf(p.norm[0] * p.inv_world, 1.0f);
  const float denorm = fmaxf(p.norm[1] * p.inv_world, 1e-6f);
  const bool sisoft = (p.soft_weight != 0.f) && (p.n % 2 != 0) && p.n >= 3;

  // ---------------- classification: one thread per 4 consecutive classes of one location ----------
  float lsum = 0.f, ssum = 0.f;
  const long long total = (long long)M * c4;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int m = (int)(i / c4);
    const int c = (int)(i - (long long)m * c4) * 4;
    const f32x4 xv = *reinterpret_cast<const f32x4*>(p.cls_logits + (long long)m * p.ld_cls + c);
    const int label = (int)p.labels[m];
    const float wgt = p.cls_weight[m];
    const float gs = wgt / num_pos * p.grad_scale;
    float g[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float l, d;
      focal_fb(xv[e], label == c + e, l, d);
      lsum += l * wgt;
      g[e] = d * gs;
    }
    if (sisoft) {
      int lvl, img, y, x;
      decode_loc(p, m, lvl, img, y, x);
      if (img == p.n - 2 && lvl >= 1) {
        // this is cls_scores[lvl][B-2]; partner cls_scores[lvl-1][B-1][:, :h, :w]  (fcos_head.py:315-319)
        const int pl = lvl - 1;
        const int m2 = p.mstart[pl] + ((p.n - 1) * p.h[pl] + y) * p.w[pl] + x;
        const f32x4 bv = *reinterpret_cast<const f32x4*>(p.cls_logits + (long long)m2 * p.ld_cls + c);
        const float inv_cnt = 1.f / ((float)C * (float)p.h[lvl] * (float)p.w[lvl]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float d = xv[e] - bv[e];
          ssum += d * d * inv_cnt;
          g[e] += 2.f * d * inv_cnt * p.soft_weight * p.grad_scale;
        }
      }
      if (img == p.n - 1 && lvl + 1 < p.nlvl && y < p.h[lvl + 1] && x < p.w[lvl + 1]) {
        const int nl = lvl + 1;
        const int m2 = p.mstart[nl] + ((p.n - 2) * p.h[nl] + y) * p.w[nl] + x;
        const f32x4 av = *reinterpret_cast<const f32x4*>(p.cls_logits + (long long)m2 * p.ld_cls + c);
        const float inv_cnt = 1.f / ((float)C * (float)p.h[nl] * (float)p.w[nl]);
#pragma unroll
        for (int e = 0; e < 4; ++e) g[e] -= 2.f * (av[e] - xv[e]) * inv_cnt * p.soft_weight * p.grad_scale;
      }
    }
    u32x2 o = {pack2bf(g[0], g[1]), pack2bf(g[2], g[3])};
    *reinterpret_cast<u32x2*>(p.g_cls + (long long)m * p.ld_gcls + c) = o;
  }
  lsum = block_sum(lsum, sh);
  ssum = block_sum(ssum, sh);
  float* rec = p.part + 16ll * blockIdx.x;          // [cls, sisoft, bbox, centerness, g_scale x 5, pad]
  if (threadIdx.x == 0) {
    rec[0] = lsum;
    rec[1] = ssum;
  }

  // ---------------- boxes + centerness: one thread per location ---------------------------------
  float bsum = 0.f, csum = 0.f;
  float gsc[DSL_MAX_SEG] = {0.f, 0.f, 0.f, 0.f, 0.f};
  for (int m = blockIdx.x * blockDim.x + threadIdx.x; m < M; m += gridDim.x * blockDim.x) {
    float gr[4] = {0.f, 0.f, 0.f, 0.f}, gc = 0.f;
    const int label = (int)p.labels[m];
    if (label < C) {
      int lvl, img, y, x;
      decode_loc(p, m, lvl, img, y, x);
      const int s = p.stride[lvl];
      const float px = (float)x * (float)s + (float)(s / 2), py = (float)y * (float)s + (float)(s / 2);
      const float sc = p.scales[lvl];
      const float* rc = p.regctr + (long long)m * p.ld_rc;
      float raw[4], d[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        raw[k] = rc[k];
        d[k] = fmaxf(raw[k] * sc, 0.f);
      }
      const f32x4 t = *reinterpret_cast<const f32x4*>(p.bbox_targets + 4ll * m);
      const float pw = p.pos_weight[m];
      const float ct = sqrtf((fminf(t[0], t[2]) / fmaxf(t[0], t[2])) * (fminf(t[1], t[3]) / fmaxf(t[1], t[3])));
      // decode (distance2bbox, core/bbox/transforms.py:119-162; no clip in training)
      const float x1 = px - d[0], y1 = py - d[1], x2 = px + d[2], y2 = py + d[3];
      const float X1 = px - t[0], Y1 = py - t[1], X2 = px + t[2], Y2 = py + t[3];
      const float eps = 1e-6f;
      const float a1 = (x2 - x1) * (y2 - y1), a2 = (X2 - X1) * (Y2 - Y1);
      const float ltx = fmaxf(x1, X1), lty = fmaxf(y1, Y1), rbx = fminf(x2, X2), rby = fminf(y2, Y2);
      const float w0 = rbx - ltx, h0 = rby - lty;
      const float iw = fmaxf(w0, 0.f), ih = fmaxf(h0, 0.f);
      const float ov = iw * ih;
      const float u0 = a1 + a2 - ov;
      const float U = fmaxf(u0, eps);
      const float ex1 = fminf(x1, X1), ey1 = fminf(y1, Y1), ex2 = fmaxf(x2, X2), ey2 = fmaxf(y2, Y2);
      const float ew0 = ex2 - ex1, eh0 = ey2 - ey1;
      const float ew = fmaxf(ew0, 0.f), eh = fmaxf(eh0, 0.f);
      const float e0 = ew * eh;
      const float E = fmaxf(e0, eps);
      const float giou = ov / U - (E - U) / E;
      const float wb = ct * pw;
      bsum += wb * (1.f - giou);
      // ---- backward of (1 - giou) w.r.t. (x1, y1, x2, y2) ----
      const float cw = w0 >= 0.f ? 1.f : 0.f, chh = h0 >= 0.f ? 1.f : 0.f;   // clamp(min=0) passes grad at 0
      const float dw_x1 = -dmax_a(x1, X1) * cw, dw_x2 = dmin_a(x2, X2) * cw;
      const float dh_y1 = -dmax_a(y1, Y1) * chh, dh_y2 = dmin_a(y2, Y2) * chh;
      const float dov[4] = {ih * dw_x1, iw * dh_y1, ih * dw_x2, iw * dh_y2};
      const float da1[4] = {-(y2 - y1), -(x2 - x1), (y2 - y1), (x2 - x1)};
      const float ug = u0 > eps ? 1.f : (u0 == eps ? 0.5f : 0.f);
      const float cew = ew0 >= 0.f ? 1.f : 0.f, ceh = eh0 >= 0.f ? 1.f : 0.f;
      const float dew_x1 = -dmin_a(x1, X1) * cew, dew_x2 = dmax_a(x2, X2) * cew;
      const float deh_y1 = -dmin_a(y1, Y1) * ceh, deh_y2 = dmax_a(y2, Y2) * ceh;
      const float eg = e0 > eps ? 1.f : (e0 == eps ? 0.5f : 0.f);
      const float de[4] = {eh * dew_x1 * eg, ew * deh_y1 * eg, eh * dew_x2 * eg, ew * deh_y2 * eg};
      const float coef = -wb / denorm * p.grad_scale;   // dL/dgiou
      float dbox[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float dU = (da1[k] - dov[k]) * ug;
        const float dg = dov[k] / U - ov * dU / (U * U) + dU / E - U * de[k] / (E * E);
        dbox[k] = coef * dg;
      }
      const float dd[4] = {-dbox[0], -dbox[1], dbox[2], dbox[3]};   // x1 = px - d0, ... x2 = px + d2
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float on = (raw[k] * sc > 0.f) ? 1.f : 0.f;
        gr[k] = dd[k] * on * sc;
        gsc[lvl] += dd[k] * on * raw[k];
      }
      // centerness BCE-with-logits (cross_entropy_loss.py:73-112)
      const float cl = rc[4];
      const float e = __expf(-fabsf(cl));
      const float sig = cl >= 0.f ? 1.f / (1.f + e) : e / (1.f + e);
      csum += (fmaxf(cl, 0.f) - cl * ct + log1pf(e)) * pw;
      gc = (sig - ct) * pw / num_pos * p.grad_scale;
    }
    u32x4 o = {pack2bf(gr[0], gr[1]), pack2bf(gr[2], gr[3]), pack2bf(gc, 0.f), 0u};
    *reinterpret_cast<u32x4*>(p.g_rc + (long long)m * p.ld_grc) = o;
  }
  bsum = block_sum(bsum, sh);
  csum = block_sum(csum, sh);
  if (threadIdx.x == 0) {
    rec[2] = bsum;
    rec[3] = csum;
  }
#pragma unroll
  for (int l = 0; l < DSL_MAX_SEG; ++l) {
    const float v = block_sum(gsc[l], sh);
    if (threadIdx.x == 0) rec[4 + l] = v;
  }
#pragma unroll
  for (int l = 4 + DSL_MAX_SEG; l < 16; ++l)
    if (threadIdx.x == 0) rec[l] = 0.f;
}

int fill(const dsl_fcos_desc* d, FcK& k) {
  DSL_CHECK(d && d->nlvl >= 1 && d->nlvl <= DSL_MAX_SEG && d->n >= 1, "fcos: bad descriptor");
  memset(&k, 0, sizeof(k));
  k.nlvl = d->nlvl; k.n = d->n; k.num_classes = d->num_classes;
  k.n_labeled = d->n / 2;      // fcos_head.py:225-231: first floor(B/2) images are the labeled stream
  int m = 0;
  for (int l = 0; l < d->nlvl; ++l) {
    k.h[l] = d->h[l]; k.w[l] = d->w[l]; k.stride[l] = d->stride[l];
    k.lo[l] = d->range_lo[l]; k.hi[l] = d->range_hi[l];
    k.mstart[l] = m;
    m += d->n * d->h[l] * d->w[l];
  }
  k.mstart[d->nlvl] = m;
  k.part = (float*)d->workspace;
  k.radius = d->radius; k.loss_weight = d->loss_weight; k.soft_weight = d->soft_weight;
  k.grad_scale = d->grad_scale; k.inv_world = d->inv_world;
  k.gt_boxes = d->gt_boxes; k.gt_labels = (const long long*)d->gt_labels; k.gt_off = d->gt_off;
  k.ig_boxes = d->ig_boxes; k.ig_off = d->ig_off;
  k.labels = (long long*)d->labels; k.bbox_targets = d->bbox_targets; k.assign_idx = d->assign_idx;
  k.cls_weight = d->cls_weight; k.pos_weight = d->pos_weight; k.stats = d->stats;
  k.cls_logits = d->cls_logits; k.regctr = d->regctr; k.ld_cls = d->ld_cls; k.ld_rc = d->ld_rc;
  k.scales = d->scales; k.norm = d->norm;
  k.g_cls = (uint16_t*)d->g_cls; k.ld_gcls = d->ld_gcls; k.g_rc = (uint16_t*)d->g_rc; k.ld_grc = d->ld_grc;
  k.g_scales = d->g_scales; k.losses = d->losses;
  return 0;
}

}  // namespace

extern "C" size_t dsl_fcos_workspace_bytes(const dsl_fcos_desc* d) {
  if (!d) return 0;
  long long M = 0;
  for (int l = 0; l < d->nlvl; ++l) M += (long long)d->n * d->h[l] * d->w[l];
  const size_t assign_rec = (size_t)((M + 255) / 256) * 2, loss_rec = 2048 * 16;
  return (assign_rec > loss_rec ? assign_rec : loss_rec) * sizeof(float);
}

extern "C" int dsl_fcos_points(const dsl_fcos_desc* d, float* points, void* stream) {
  FcK k;
  if (fill(d, k)) return -1;
  DSL_CHECK(points, "dsl_fcos_points: null output");
  int P = 0;
  for (int l = 0; l < d->nlvl; ++l) P += d->h[l] * d->w[l];
  hipLaunchKernelGGL(points_kernel, dim3((P + 255) / 256), dim3(256), 0, (hipStream_t)stream, k, points);
  DSL_LAUNCH_CHECK("points_kernel");
  return 0;
}

extern "C" int dsl_fcos_assign(const dsl_fcos_desc* d, void* stream) {
  FcK k;
  if (fill(d, k)) return -1;
  DSL_CHECK(d->gt_boxes && d->gt_labels && d->gt_off && d->labels && d->bbox_targets && d->assign_idx &&
                d->cls_weight && d->pos_weight && d->stats,
            "dsl_fcos_assign: null pointer");
  hipStream_t st = (hipStream_t)stream;
  const int M = k.mstart[k.nlvl];
  const int nb = (M + 255) / 256;
  DSL_CHECK(d->workspace && d->workspace_bytes >= dsl_fcos_workspace_bytes(d), "dsl_fcos_assign: workspace too small");
  hipLaunchKernelGGL(assign_kernel, dim3(nb), dim3(256), 0, st, k);
  // stats[0] = num_pos, stats[1] = sum of centerness targets (this rank), stats[2..7] = 0: block records added in block order
  hipLaunchKernelGGL(fcos_finalize_kernel, dim3(1), dim3(FIN_T), 0, st, (const float*)k.part, nb, 2, d->stats, 2, (float*)nullptr,
                     (const float*)nullptr, 1.f, 0.f, 0);
  DSL_LAUNCH_CHECK("assign_kernel");
  return 0;
}

extern "C" int dsl_fcos_loss(const dsl_fcos_desc* d, void* stream) {
  FcK k;
  if (fill(d, k)) return -1;
  DSL_CHECK(d->labels && d->bbox_targets && d->cls_weight && d->pos_weight && d->cls_logits && d->regctr &&
                d->scales && d->norm && d->g_cls && d->g_rc && d->g_scales && d->losses,
            "dsl_fcos_loss: null pointer");
  DSL_CHECK(d->num_classes % 4 == 0 && d->ld_cls % 4 == 0 && d->ld_gcls % 4 == 0 && d->ld_grc % 8 == 0 && d->ld_rc >= 5,
            "dsl_fcos_loss: unsupported strides / class count");
  hipStream_t st = (hipStream_t)stream;
  DSL_CHECK(d->workspace && d->workspace_bytes >= dsl_fcos_workspace_bytes(d), "dsl_fcos_loss: workspace too small");
  const int M = k.mstart[k.nlvl];
  int blocks = (int)(((long long)M * (d->num_classes / 4) + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(loss_kernel, dim3(blocks), dim3(256), 0, st, k);
  hipLaunchKernelGGL(fcos_finalize_kernel, dim3(1), dim3(FIN_T), 0, st, (const float*)k.part, blocks, 16, d->losses,
                     DSL_MAX_SEG, d->g_scales, d->norm, d->inv_world, d->soft_weight, 1, d->logvec);
  DSL_LAUNCH_CHECK("loss_kernel");
  return 0;
}
