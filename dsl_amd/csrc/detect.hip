// Teacher-sweep post-processing on the GPU: per-level top-k by max_c(score*centerness), box decode +
// clip + rescale, score threshold, class-aware greedy NMS, top max_per_img.
//
// Restates FCOSHead._get_bboxes (mmdet/models/dense_heads/fcos_head.py:406-548), multiclass_nms
// (mmdet/core/post_processing/bbox_nms.py:7-94) and mmcv.ops.batched_nms / nms (mmcv-full 1.3.10,
// un-vendored: class-offset trick boxes + label*(max+1), suppress IoU > thr, offset 0, scores sorted
// descending) of the reference, replacing the GPU->CPU numpy->JSON round trip of
// mmdet/runner/hooks/unlabel_pred_hook.py:194-293.
#include "common.hpp"
#pragma clang fp contract(off)

namespace {

constexpr int CAND_CAP = 16384;
constexpr int NMS_THREADS = 1024;

struct DetK {
  int nlvl, n, num_classes, nms_pre, max_per_img;
  int h[DSL_MAX_SEG], w[DSL_MAX_SEG], stride[DSL_MAX_SEG];
  int mstart[DSL_MAX_SEG + 1];
  float score_thr, iou_thr;
  const float* cls; int ld_cls;
  const float* rc; int ld_rc;
  const float* scales; const float* img_shapes; const float* scale_factors;
  float* dets; long long* det_labels; int* det_count;
  // workspace
  float* keys; int* sel; int* selcnt; float* cbox; float* cscore; int* clabel; int* ccount; float* pairscore;
};

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

// key[m] = max_c sigmoid(cls) * sigmoid(ctr)     (fcos_head.py:472-473)
__global__ void det_key_kernel(const DetK p) {
  const int M = p.mstart[p.nlvl];
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= M) return;
  const float* c = p.cls + (long long)m * p.ld_cls;
  const float ctr = sigmoidf_(p.rc[(long long)m * p.ld_rc + 4]);
  float best = -1.f;
  for (int k = 0; k < p.num_classes; k += 4) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(c + k);
#pragma unroll
    for (int e = 0; e < 4; ++e) best = fmaxf(best, sigmoidf_(v[e]) * ctr);
  }
  p.keys[m] = best;
}

// Block-wide radix select over non-negative floats (bit pattern is monotonic): returns the bit pattern
// of the k-th largest value and how many elements equal to it belong to the top k.  11 + 11 + 10 bits.
__device__ void radix_select_block(const float* keys, int P, int k, unsigned* hist /*2048 + 1024, blockDim 1024*/, unsigned* s_tmp /*2*/,
                                   unsigned& prefix_out, unsigned& need_out) {
  unsigned prefix = 0, mask = 0, need = (unsigned)k;
  const int shifts[3] = {21, 10, 0};
  const int bits[3] = {11, 11, 10};
  for (int pass = 0; pass < 3; ++pass) {
    const int nb = 1 << bits[pass];
    for (int i = threadIdx.x; i < 2048; i += blockDim.x) hist[i] = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < P; i += blockDim.x) {
      const float f = keys[i];
      const unsigned u = f > 0.f ? __float_as_uint(f) : 0u;
      // (non-positive keys all fall into bin 0 of the first pass: hundreds of thousands of atomics on one LDS word for the
      // dense pair scores; the callers only select among positive keys, so they are left out of the histogram)
      if (u != 0u && (u & mask) == prefix) atomicAdd(&hist[(u >> shifts[pass]) & (nb - 1)], 1u);
    }
    __syncthreads();
    // the bin that holds the need-th largest key: suffix sums over the bins, in parallel (a single thread walking 2 048 LDS
    // words took ~80 us per pass).  Thread t owns bins 2t, 2t+1; Hillis-Steele suffix scan over the 1 024 pair sums in place.
    {
      unsigned* pair = hist + 2048;                  // caller provides 2048 + 1024 words
      const int t = threadIdx.x;
      const unsigned h0 = hist[2 * t], h1 = hist[2 * t + 1];
      pair[t] = h0 + h1;
      __syncthreads();
      for (int d = 1; d < 1024; d <<= 1) {
        const unsigned add = t + d < 1024 ? pair[t + d] : 0u;
        __syncthreads();
        pair[t] += add;
        __syncthreads();
      }
      // pair[t] = number of keys in bins >= 2t; above bin 2t+1: pair[t] - h0 - h1
      const unsigned above1 = pair[t] - h0 - h1, above0 = above1 + h1;       // keys in bins > 2t+1, > 2t
      if (above1 < need && above1 + h1 >= need) { s_tmp[0] = prefix | ((unsigned)(2 * t + 1) << shifts[pass]); s_tmp[1] = need - above1; }
      if (above0 < need && above0 + h0 >= need) { s_tmp[0] = prefix | ((unsigned)(2 * t) << shifts[pass]); s_tmp[1] = need - above0; }
      if (t == 0 && pair[0] < need) { s_tmp[0] = prefix; s_tmp[1] = need - (pair[0] - h0); }     // fewer keys than asked for: bin 0 (as before)
    }
    __syncthreads();
    prefix = s_tmp[0];
    need = s_tmp[1];
    mask |= ((unsigned)(nb - 1)) << shifts[pass];
    __syncthreads();
  }
  prefix_out = prefix;
  need_out = need;
}

// Ordered stream compaction by one workgroup: element i of [0, P) is taken when take(i) says so, taken elements get
// consecutive output slots IN INDEX ORDER (no atomics: results do not depend on wave arrival order).  Among the
// elements for which tie(i) holds only the first `need` (lowest indices) are taken - the deterministic version of
// "any `need` of the equal keys", and the choice torch.topk makes on the CPU path of the reference.
// Two passes over wave-contiguous segments: count (ballot + popcount), workgroup scan of the 16 wave totals, write.
template <typename Take, typename Tie, typename Emit>
__device__ void ordered_compact(int P, unsigned need, unsigned* s_wave /* >= 2 * 16 + 2 */, Take take, Tie tie, Emit emit,
                                unsigned* total_out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  constexpr int U = 4;                                           // 64-element groups per iteration: U predicate loads in flight
  const int seg = ((P + nw - 1) / nw + 64 * U - 1) / (64 * U) * (64 * U);      // elements per wave, multiple of 64 * U
  const int lo = wave * seg, hi = min(P, lo + seg);
  unsigned ntake = 0, ntie = 0;
  for (int i0 = lo; i0 < hi; i0 += 64 * U) {
    bool tk[U], ti[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = i0 + u * 64 + lane;
      const bool in = i < hi;
      tk[u] = in && take(i);
      ti[u] = in && tie(i);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      ntake += __popcll(__ballot(tk[u]));
      ntie += __popcll(__ballot(ti[u]));
    }
  }
  if (lane == 0) {
    s_wave[wave] = ntake;
    s_wave[16 + wave] = ntie;
  }
  __syncthreads();
  // ties taken from the waves in front of this one, and the output slots they (and their sure takes) use
  unsigned tie_before = 0, out_before = 0;
  for (int w = 0; w < wave; ++w) {
    const unsigned t = s_wave[16 + w];
    const unsigned used = tie_before < need ? min(t, need - tie_before) : 0u;
    out_before += s_wave[w] + used;
    tie_before += t;
  }
  unsigned out = out_before, tr = tie_before;
  const unsigned long long lt = (1ull << lane) - 1ull;
  for (int i0 = lo; i0 < hi; i0 += 64 * U) {
    bool tk[U], ti[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = i0 + u * 64 + lane;
      const bool in = i < hi;
      tk[u] = in && take(i);
      ti[u] = in && tie(i);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {                   // index order: group u before group u + 1
      const int i = i0 + u * 64 + lane;
      const unsigned long long bt = __ballot(ti[u]);
      const unsigned my_tr = tr + (unsigned)__popcll(bt & lt);
      const bool sel = tk[u] || (ti[u] && my_tr < need);
      const unsigned long long bs = __ballot(sel);
      if (sel) emit(i, out + (unsigned)__popcll(bs & lt));
      out += (unsigned)__popcll(bs);
      tr += (unsigned)__popcll(bt);
    }
  }
  if (total_out) {
    __syncthreads();
    if (threadIdx.x == blockDim.x - 1) *total_out = out;       // the last wave's running offset = the total
    __syncthreads();
  }
}

// one block per (image, level): indices of the top nms_pre keys (all of them when the level is smaller), in index order
__global__ __launch_bounds__(1024) void det_select_kernel(const DetK p) {
  __shared__ unsigned hist[2048 + 1024];
  __shared__ unsigned s_tmp[2], s_wave[34], s_total;
  const int lvl = blockIdx.x, img = blockIdx.y;
  const int P = p.h[lvl] * p.w[lvl];
  const int base = p.mstart[lvl] + img * P;
  int* sel = p.sel + ((long long)img * p.nlvl + lvl) * p.nms_pre;
  int* cnt = p.selcnt + img * p.nlvl + lvl;
  const int k = p.nms_pre;
  if (!(k > 0 && k < P)) {        // get_k_for_topk: take everything
    for (int i = threadIdx.x; i < P; i += blockDim.x) sel[i] = i;
    if (threadIdx.x == 0) *cnt = P;
    return;
  }
  unsigned prefix, need;
  radix_select_block(p.keys + base, P, k, hist, s_tmp, prefix, need);
  const float* keys = p.keys + base;
  auto bits = [&](int i) { const float f = keys[i]; return f > 0.f ? __float_as_uint(f) : 0u; };
  ordered_compact(
      P, need, s_wave, [&](int i) { return bits(i) > prefix; }, [&](int i) { return bits(i) == prefix; },
      [&](int i, unsigned slot) { sel[slot] = i; }, &s_total);
  if (threadIdx.x == 0) *cnt = (int)s_total;
}

// dense final scores of every (selected location, class) pair: score*centerness if score > score_thr
// (bbox_nms.py:54: validity is tested BEFORE the centerness factor), else -1
__global__ void det_pairscore_kernel(const DetK p) {
  const int img = blockIdx.z, lvl = blockIdx.y;
  const int nsel = p.selcnt[img * p.nlvl + lvl];
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= p.nms_pre * p.num_classes) return;
  const int slot = t / p.num_classes, c = t - slot * p.num_classes;
  float out = -1.f;
  if (slot < nsel) {
    const int P = p.h[lvl] * p.w[lvl];
    const int loc = p.sel[((long long)img * p.nlvl + lvl) * p.nms_pre + slot];
    const int m = p.mstart[lvl] + img * P + loc;
    const float score = sigmoidf_(p.cls[(long long)m * p.ld_cls + c]);
    if (score > p.score_thr) out = score * sigmoidf_(p.rc[(long long)m * p.ld_rc + 4]);
  }
  p.pairscore[((long long)img * p.nlvl + lvl) * p.nms_pre * p.num_classes + t] = out;
}

// one candidate: decode the box of pair i (level, slot, class) and store it in candidate slot `out`
__device__ __forceinline__ void det_emit(const DetK& p, int img, const float* ps, int i, unsigned out) {
  if (out >= (unsigned)CAND_CAP) return;
  const float f = ps[i];
  const int c = i % p.num_classes;
  const int slot = (i / p.num_classes) % p.nms_pre;
  const int lvl = i / (p.num_classes * p.nms_pre);
  const int P = p.h[lvl] * p.w[lvl];
  const int loc = p.sel[((long long)img * p.nlvl + lvl) * p.nms_pre + slot];
  const int m = p.mstart[lvl] + img * P + loc;
  const float* rc = p.rc + (long long)m * p.ld_rc;
  const int s = p.stride[lvl];
  const int y = loc / p.w[lvl], x = loc - y * p.w[lvl];
  const float px = (float)x * (float)s + (float)(s / 2), py = (float)y * (float)s + (float)(s / 2);
  const float sc = p.scales[lvl];
  float d[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) d[e] = fmaxf(rc[e] * sc, 0.f) * (float)s;      // fcos_head.py:159-165 (eval)
  const float H = p.img_shapes[2 * img], W = p.img_shapes[2 * img + 1];
  float b[4] = {px - d[0], py - d[1], px + d[2], py + d[3]};
  b[0] = fminf(fmaxf(b[0], 0.f), W);                    // distance2bbox clip (transforms.py:150-160)
  b[1] = fminf(fmaxf(b[1], 0.f), H);
  b[2] = fminf(fmaxf(b[2], 0.f), W);
  b[3] = fminf(fmaxf(b[3], 0.f), H);
  if (p.scale_factors) {
#pragma unroll
    for (int e = 0; e < 4; ++e) b[e] = b[e] / p.scale_factors[4 * img + e];
  }
  float* cb = p.cbox + ((long long)img * CAND_CAP + out) * 4;
  cb[0] = b[0]; cb[1] = b[1]; cb[2] = b[2]; cb[3] = b[3];
  p.cscore[(long long)img * CAND_CAP + out] = f;
  p.clabel[(long long)img * CAND_CAP + out] = c;
}

// Candidate compaction, common case (at most CAND_CAP valid pairs - always, once the detector is trained): DET_CB
// workgroups per image, each owns a contiguous chunk of the (level, slot, class) index range.  Pass 1 counts the chunk's
// valid pairs; pass 2 places them behind the chunks in front of it, in index order inside the chunk (ordered_compact):
// the same candidate order as one workgroup walking all 400 000 pairs, which took 370 us.
constexpr int DET_CB = 64;
__device__ __forceinline__ void det_chunk(const DetK& p, int& lo, int& hi) {
  const int per_img = p.nlvl * p.nms_pre * p.num_classes;
  const int chunk = ((per_img + DET_CB - 1) / DET_CB + 255) / 256 * 256;
  lo = min(per_img, (int)blockIdx.x * chunk);
  hi = min(per_img, lo + chunk);
}
__global__ __launch_bounds__(1024) void det_count_kernel(const DetK p) {
  __shared__ unsigned s_cnt;
  const int img = blockIdx.y;
  const float* ps = p.pairscore + (long long)img * p.nlvl * p.nms_pre * p.num_classes;
  int lo, hi;
  det_chunk(p, lo, hi);
  if (threadIdx.x == 0) s_cnt = 0;
  __syncthreads();
  unsigned local = 0;
  for (int i = lo + threadIdx.x; i < hi; i += blockDim.x) local += ps[i] > 0.f ? 1u : 0u;
  atomicAdd(&s_cnt, local);            // integer count: order-independent
  __syncthreads();
  if (threadIdx.x == 0) p.ccount[p.n + img * DET_CB + blockIdx.x] = (int)s_cnt;
}
__global__ __launch_bounds__(1024) void det_scatter_kernel(const DetK p) {
  __shared__ unsigned s_wave[34], s_total;
  const int img = blockIdx.y;
  const int* cnt = p.ccount + p.n + img * DET_CB;
  unsigned total = 0, before = 0;
  for (int b = 0; b < DET_CB; ++b) {
    const unsigned c = (unsigned)cnt[b];
    total += c;
    before += b < (int)blockIdx.x ? c : 0u;
  }
  if (total > (unsigned)CAND_CAP) return;            // rare: det_compact_kernel selects the best CAND_CAP
  if (blockIdx.x == 0 && threadIdx.x == 0) p.ccount[img] = (int)total;
  const float* ps = p.pairscore + (long long)img * p.nlvl * p.nms_pre * p.num_classes;
  int lo, hi;
  det_chunk(p, lo, hi);
  if (cnt[blockIdx.x] == 0) return;
  ordered_compact(
      hi - lo, 0u, s_wave, [&](int i) { return ps[lo + i] > 0.f; }, [&](int) { return false; },
      [&](int i, unsigned out) { det_emit(p, img, ps, lo + i, before + out); }, &s_total);
}

// one block per image: keep the CAND_CAP best pairs when more than that are valid; candidate slots are assigned in
// (level, slot, class) index order (ordered_compact), so the score-tie order of the NMS below - and with it the whole
// result - is reproducible
__global__ __launch_bounds__(1024) void det_compact_kernel(const DetK p) {
  __shared__ unsigned hist[2048 + 1024];
  __shared__ unsigned s_tmp[2], s_wave[34], s_total;
  const int img = blockIdx.x;
  const int per_img = p.nlvl * p.nms_pre * p.num_classes;
  const float* ps = p.pairscore + (long long)img * per_img;
  unsigned nvalid = 0;
  for (int b = 0; b < DET_CB; ++b) nvalid += (unsigned)p.ccount[p.n + img * DET_CB + b];
  if (nvalid <= (unsigned)CAND_CAP) return;          // det_scatter_kernel did it
  unsigned prefix = 0, need = 0;
  radix_select_block(ps, per_img, CAND_CAP, hist, s_tmp, prefix, need);
  ordered_compact(
      per_img, need, s_wave, [&](int i) { const float f = ps[i]; return f > 0.f && __float_as_uint(f) > prefix; },
      [&](int i) { const float f = ps[i]; return f > 0.f && __float_as_uint(f) == prefix; },
      [&](int i, unsigned out) { det_emit(p, img, ps, i, out); }, &s_total);
  if (threadIdx.x == 0) p.ccount[img] = min((int)s_total, CAND_CAP);
}

__device__ __forceinline__ bool iou_gt(const float* a, const float* b, float thr) {
  // mmcv nms_cuda_kernel.cuh IoU with offset 0
  const float left = fmaxf(a[0], b[0]), right = fminf(a[2], b[2]);
  const float top = fmaxf(a[1], b[1]), bottom = fminf(a[3], b[3]);
  const float width = fmaxf(right - left, 0.f), height = fmaxf(bottom - top, 0.f);
  const float inter = width * height;
  const float sa = (a[2] - a[0]) * (a[3] - a[1]), sb = (b[2] - b[0]) * (b[3] - b[1]);
  return inter / (sa + sb - inter) > thr;
}

// one block per image: sort candidates by score (bitonic, LDS), greedy class-aware NMS, keep max_per_img
__global__ __launch_bounds__(NMS_THREADS) void det_nms_kernel(const DetK p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned long long* keyidx = reinterpret_cast<unsigned long long*>(smem);          // CAND_CAP * 8
  float* kept = reinterpret_cast<float*>(smem + (size_t)CAND_CAP * 8);                // max_per_img * 4 (offset boxes)
  __shared__ float s_red[NMS_THREADS / 64];
  __shared__ int s_nkept, s_flag;
  const int img = blockIdx.x;
  int n = p.ccount[img];
  if (n > CAND_CAP) n = CAND_CAP;
  const float* cbox = p.cbox + (long long)img * CAND_CAP * 4;
  const float* cscore = p.cscore + (long long)img * CAND_CAP;
  const int* clabel = p.clabel + (long long)img * CAND_CAP;
  // max coordinate over the valid boxes (batched_nms: offsets = idxs * (boxes.max() + 1))
  float mx = -3.0e38f;
  for (int i = threadIdx.x; i < n * 4; i += blockDim.x) mx = fmaxf(mx, cbox[i]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = mx;
  __syncthreads();
  mx = s_red[0];
  for (int i = 1; i < NMS_THREADS / 64; ++i) mx = fmaxf(mx, s_red[i]);
  const float offs = mx + 1.0f;
  // sort descending by score; ties by ascending candidate slot (deterministic given the slot order)
  int np2 = 1;
  while (np2 < n) np2 <<= 1;
  for (int i = threadIdx.x; i < np2; i += blockDim.x) {
    unsigned long long k = 0ull;
    if (i < n) k = ((unsigned long long)__float_as_uint(cscore[i]) << 32) | (unsigned)(0xffffffffu - (unsigned)i);
    keyidx[i] = k;
  }
  __syncthreads();
  for (int k = 2; k <= np2; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = threadIdx.x; i < np2; i += blockDim.x) {
        const int ixj = i ^ j;
        if (ixj > i) {
          const unsigned long long a = keyidx[i], b = keyidx[ixj];
          const bool desc = (i & k) == 0;
          if (desc ? (a < b) : (a > b)) {
            keyidx[i] = b;
            keyidx[ixj] = a;
          }
        }
      }
      __syncthreads();
    }
  if (threadIdx.x == 0) s_nkept = 0;
  __syncthreads();
  // greedy scan: candidate i survives iff no kept box (same label via the offset trick) has IoU > thr
  const int maxk = p.max_per_img;
  for (int i = 0; i < n; ++i) {
    const int nk = s_nkept;
    if (nk >= maxk) break;
    const int ci = (int)(0xffffffffu - (unsigned)(keyidx[i] & 0xffffffffu));
    const float off = (float)clabel[ci] * offs;
    const float bi[4] = {cbox[4 * ci] + off, cbox[4 * ci + 1] + off, cbox[4 * ci + 2] + off, cbox[4 * ci + 3] + off};
    if (threadIdx.x == 0) s_flag = 0;
    __syncthreads();
    if ((int)threadIdx.x < nk && iou_gt(kept + 4 * threadIdx.x, bi, p.iou_thr)) s_flag = 1;
    __syncthreads();
    if (threadIdx.x == 0 && !s_flag) {
      kept[4 * nk] = bi[0]; kept[4 * nk + 1] = bi[1]; kept[4 * nk + 2] = bi[2]; kept[4 * nk + 3] = bi[3];
      float* o = p.dets + ((long long)img * maxk + nk) * 5;
      o[0] = cbox[4 * ci]; o[1] = cbox[4 * ci + 1]; o[2] = cbox[4 * ci + 2]; o[3] = cbox[4 * ci + 3];
      o[4] = cscore[ci];
      p.det_labels[(long long)img * maxk + nk] = clabel[ci];
      s_nkept = nk + 1;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) p.det_count[img] = s_nkept;
}

// ---- pseudo-label fuse step of the refresh (unlabel_pred_hook.py:84-171) -----------------------------------------
// One block per image over its <= max_per_img detections: keep score >= parse_thr, truncate the coordinates toward
// zero (int(), parse_det_results :27), round the score to 6 decimals (:35), then per class (ascending, classes
// 0 .. num_classes-1) mmcv.ops.nms(boxes, scores, iou_threshold, score_threshold): candidates with score > nms_thr in
// descending score order (ties: earlier detection first), greedy, suppress IoU > iou_thr (offset 0).
// fuse_history=True (:131-141): the image's previous labels (old_*: boxes, scores and class indices as the label file
// held them - not truncated, not thresholded by parse_thr) come first in the candidate list, the new detections after.
struct FuseK {
  int n, maxk, num_classes, max_old, max_out;
  float parse_thr, iou_thr, nms_thr;
  const float* dets; const long long* labels; const int* count;
  const float* old_boxes; const float* old_scores; const long long* old_labels; const int* old_count;
  float* out_boxes; float* out_scores; long long* out_labels; int* out_count;
};

constexpr int FUSE_T = 256, FUSE_MAX = 1024;
__global__ __launch_bounds__(FUSE_T) void pseudo_fuse_kernel(const FuseK p) {
  __shared__ float bx[FUSE_MAX][4];
  __shared__ float sc[FUSE_MAX];
  __shared__ int lb[FUSE_MAX];          // -1: dropped
  __shared__ int order[FUSE_MAX];       // candidate index at each rank of (label asc, score desc, index asc)
  __shared__ int kept[FUSE_MAX];
  __shared__ int s_nk, s_flag, s_ncand, s_cls0;
  const int img = blockIdx.x;
  const int ko = p.old_boxes ? min(p.old_count[img], p.max_old) : 0;
  const int k = ko + min(p.count[img], p.maxk);          // ko + k <= FUSE_MAX (checked by the caller)
  for (int i = threadIdx.x; i < ko; i += FUSE_T) {
    const long long o = (long long)img * p.max_old + i;
    const float s = p.old_scores[o];
    const int l = (int)p.old_labels[o];
#pragma unroll
    for (int e = 0; e < 4; ++e) bx[i][e] = p.old_boxes[o * 4 + e];
    sc[i] = s;
    lb[i] = (l >= 0 && l < p.num_classes && s > p.nms_thr) ? l : -1;
  }
  for (int i = ko + threadIdx.x; i < k; i += FUSE_T) {
    const float* d = p.dets + ((long long)img * p.maxk + (i - ko)) * 5;
    const float s = d[4];
    const int l = (int)p.labels[(long long)img * p.maxk + (i - ko)];
    const float rs = (float)(nearbyint((double)s * 1e6) / 1e6);
#pragma unroll
    for (int e = 0; e < 4; ++e) bx[i][e] = (float)(int)d[e];
    sc[i] = rs;
    lb[i] = (s >= p.parse_thr && l >= 0 && l < p.num_classes && rs > p.nms_thr) ? l : -1;
  }
  if (threadIdx.x == 0) {
    s_nk = 0;
    s_ncand = 0;
  }
  __syncthreads();
  // rank by counting (k <= 1024): dropped entries go last
  for (int i = threadIdx.x; i < k; i += FUSE_T) {
    int r = 0;
    const int li = lb[i] < 0 ? 0x7fffffff : lb[i];
    for (int j = 0; j < k; ++j) {
      const int lj = lb[j] < 0 ? 0x7fffffff : lb[j];
      const bool before = lj < li || (lj == li && (sc[j] > sc[i] || (sc[j] == sc[i] && j < i)));
      r += before ? 1 : 0;
    }
    order[r] = i;
    if (lb[i] >= 0) atomicAdd(&s_ncand, 1);
  }
  __syncthreads();
  const int nc = s_ncand;
  for (int r = 0; r < nc; ++r) {
    const int ci = order[r];
    const int nk = s_nk;
    if (threadIdx.x == 0) {
      s_flag = 0;
      if (r == 0 || lb[order[r - 1]] != lb[ci]) s_cls0 = nk;      // first kept slot of this class
    }
    __syncthreads();
    for (int t = s_cls0 + threadIdx.x; t < nk; t += FUSE_T)
      if (iou_gt(bx[kept[t]], bx[ci], p.iou_thr)) s_flag = 1;
    __syncthreads();
    if (threadIdx.x == 0 && !s_flag) {
      kept[nk] = ci;
      s_nk = nk + 1;
    }
    __syncthreads();
  }
  const int nk = s_nk;      // <= max_out: max_out >= max_old + maxk
  for (int t = threadIdx.x; t < nk; t += FUSE_T) {
    const int ci = kept[t];
    float* ob = p.out_boxes + ((long long)img * p.max_out + t) * 4;
    ob[0] = bx[ci][0]; ob[1] = bx[ci][1]; ob[2] = bx[ci][2]; ob[3] = bx[ci][3];
    p.out_scores[(long long)img * p.max_out + t] = sc[ci];
    p.out_labels[(long long)img * p.max_out + t] = lb[ci];
  }
  if (threadIdx.x == 0) p.out_count[img] = nk;
}

size_t ws_layout(const dsl_det_desc* d, size_t off[8]) {
  long long M = 0;
  for (int l = 0; l < d->nlvl; ++l) M += (long long)d->n * d->h[l] * d->w[l];
  size_t o = 0;
  auto take = [&](size_t bytes) { size_t r = o; o += (bytes + 255) / 256 * 256; return r; };
  off[0] = take(M * 4);                                           // keys
  off[1] = take((size_t)d->n * d->nlvl * d->nms_pre * 4);         // sel
  off[2] = take((size_t)d->n * d->nlvl * 4);                      // selcnt
  off[3] = take((size_t)d->n * CAND_CAP * 16);                    // cbox
  off[4] = take((size_t)d->n * CAND_CAP * 4);                     // cscore
  off[5] = take((size_t)d->n * CAND_CAP * 4);                     // clabel
  off[6] = take((size_t)d->n * 4 * (1 + DET_CB));                 // ccount, then DET_CB chunk counts per image
  off[7] = take((size_t)d->n * d->nlvl * d->nms_pre * d->num_classes * 4);   // pairscore
  return o;
}

}  // namespace

extern "C" size_t dsl_detect_workspace_bytes(const dsl_det_desc* d) {
  size_t off[8];
  return ws_layout(d, off);
}

extern "C" int dsl_fcos_detect(const dsl_det_desc* d, void* stream) {
  DSL_CHECK(d && d->nlvl >= 1 && d->nlvl <= DSL_MAX_SEG && d->n >= 1, "dsl_fcos_detect: bad descriptor");
  DSL_CHECK(d->cls_logits && d->regctr && d->scales && d->img_shapes && d->dets && d->det_labels && d->det_count &&
                d->workspace,
            "dsl_fcos_detect: null pointer");
  DSL_CHECK(d->num_classes % 4 == 0 && d->ld_cls % 4 == 0 && d->ld_rc >= 5, "dsl_fcos_detect: unsupported layout");
  DSL_CHECK(d->max_per_img > 0 && d->max_per_img <= NMS_THREADS && d->nms_pre > 0, "dsl_fcos_detect: max_per_img must be in 1..%d", NMS_THREADS);
  size_t off[8];
  const size_t need = ws_layout(d, off);
  DSL_CHECK(d->workspace_bytes >= need, "dsl_fcos_detect: workspace too small (%zu < %zu)", d->workspace_bytes, need);
  DetK k;
  memset(&k, 0, sizeof(k));
  k.nlvl = d->nlvl; k.n = d->n; k.num_classes = d->num_classes; k.nms_pre = d->nms_pre; k.max_per_img = d->max_per_img;
  int m = 0, maxP = 0;
  for (int l = 0; l < d->nlvl; ++l) {
    k.h[l] = d->h[l]; k.w[l] = d->w[l]; k.stride[l] = d->stride[l];
    k.mstart[l] = m;
    m += d->n * d->h[l] * d->w[l];
    maxP = max(maxP, d->h[l] * d->w[l]);
  }
  k.mstart[d->nlvl] = m;
  k.score_thr = d->score_thr; k.iou_thr = d->iou_thr;
  k.cls = d->cls_logits; k.ld_cls = d->ld_cls; k.rc = d->regctr; k.ld_rc = d->ld_rc;
  k.scales = d->scales; k.img_shapes = d->img_shapes; k.scale_factors = d->scale_factors;
  k.dets = d->dets; k.det_labels = (long long*)d->det_labels; k.det_count = d->det_count;
  unsigned char* ws = (unsigned char*)d->workspace;
  k.keys = (float*)(ws + off[0]); k.sel = (int*)(ws + off[1]); k.selcnt = (int*)(ws + off[2]);
  k.cbox = (float*)(ws + off[3]); k.cscore = (float*)(ws + off[4]); k.clabel = (int*)(ws + off[5]);
  k.ccount = (int*)(ws + off[6]);
  k.pairscore = (float*)(ws + off[7]);
  hipStream_t st = (hipStream_t)stream;
  hipMemsetAsync(k.ccount, 0, sizeof(int) * d->n, st);
  hipMemsetAsync(d->dets, 0, sizeof(float) * 5 * d->n * d->max_per_img, st);
  hipLaunchKernelGGL(det_key_kernel, dim3((m + 255) / 256), dim3(256), 0, st, k);
  hipLaunchKernelGGL(det_select_kernel, dim3(d->nlvl, d->n), dim3(1024), 0, st, k);
  const int per_lvl = d->nms_pre * d->num_classes;
  hipLaunchKernelGGL(det_pairscore_kernel, dim3((per_lvl + 255) / 256, d->nlvl, d->n), dim3(256), 0, st, k);
  hipLaunchKernelGGL(det_count_kernel, dim3(DET_CB, d->n), dim3(1024), 0, st, k);
  hipLaunchKernelGGL(det_scatter_kernel, dim3(DET_CB, d->n), dim3(1024), 0, st, k);
  hipLaunchKernelGGL(det_compact_kernel, dim3(d->n), dim3(1024), 0, st, k);
  const size_t lds = (size_t)CAND_CAP * 8 + (size_t)d->max_per_img * 16;
  static bool attr = false;
  if (!attr) {
    hipFuncSetAttribute((const void*)det_nms_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)((size_t)CAND_CAP * 8 + NMS_THREADS * 16));
    attr = true;
  }
  hipLaunchKernelGGL(det_nms_kernel, dim3(d->n), dim3(NMS_THREADS), lds, st, k);
  DSL_LAUNCH_CHECK("dsl_fcos_detect");
  return 0;
}

extern "C" int dsl_pseudo_label_fuse_history(const float* dets, const int64_t* labels, const int32_t* count, int n, int max_per_img,
                                             const float* old_boxes, const float* old_scores, const int64_t* old_labels,
                                             const int32_t* old_count, int max_old, int num_classes, float parse_thr, float iou_thr,
                                             float nms_thr, float* out_boxes, float* out_scores, int64_t* out_labels,
                                             int32_t* out_count, int max_out, void* stream) {
  DSL_CHECK(dets && labels && count && out_boxes && out_scores && out_labels && out_count, "dsl_pseudo_label_fuse: null pointer");
  DSL_CHECK(n >= 1 && max_per_img >= 1 && max_old >= 0 && max_per_img + max_old <= FUSE_MAX,
            "dsl_pseudo_label_fuse: max_per_img + max_old must be in 1..%d", FUSE_MAX);
  DSL_CHECK(max_old == 0 || (old_boxes && old_scores && old_labels && old_count), "dsl_pseudo_label_fuse_history: null pointer (old labels)");
  DSL_CHECK(max_out >= max_per_img + max_old, "dsl_pseudo_label_fuse: max_out (%d) < max_per_img + max_old (%d)", max_out,
            max_per_img + max_old);
  FuseK k;
  k.n = n; k.maxk = max_per_img; k.num_classes = num_classes; k.max_old = max_old; k.max_out = max_out;
  k.parse_thr = parse_thr; k.iou_thr = iou_thr; k.nms_thr = nms_thr;
  k.dets = dets; k.labels = (const long long*)labels; k.count = count;
  k.old_boxes = max_old ? old_boxes : nullptr; k.old_scores = old_scores; k.old_labels = (const long long*)old_labels;
  k.old_count = old_count;
  k.out_boxes = out_boxes; k.out_scores = out_scores; k.out_labels = (long long*)out_labels; k.out_count = out_count;
  hipLaunchKernelGGL(pseudo_fuse_kernel, dim3(n), dim3(FUSE_T), 0, (hipStream_t)stream, k);
  DSL_LAUNCH_CHECK("pseudo_fuse_kernel");
  return 0;
}

extern "C" int dsl_pseudo_label_fuse(const float* dets, const int64_t* labels, const int32_t* count, int n, int max_per_img,
                                     int num_classes, float parse_thr, float iou_thr, float nms_thr, float* out_boxes,
                                     float* out_scores, int64_t* out_labels, int32_t* out_count, void* stream) {
  DSL_CHECK(dets && labels && count && out_boxes && out_scores && out_labels && out_count, "dsl_pseudo_label_fuse: null pointer");
  DSL_CHECK(n >= 1 && max_per_img >= 1 && max_per_img <= FUSE_MAX, "dsl_pseudo_label_fuse: max_per_img must be in 1..%d", FUSE_MAX);
  return dsl_pseudo_label_fuse_history(dets, labels, count, n, max_per_img, nullptr, nullptr, nullptr, nullptr, 0, num_classes,
                                       parse_thr, iou_thr, nms_thr, out_boxes, out_scores, out_labels, out_count, max_per_img,
                                       stream);
}
