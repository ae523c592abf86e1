// GPU data path: decoded uint8 images -> the padded, normalised fp32 NCHW batch the detector
// consumes, one launch per batch.  Fuses, per image, the reference's CPU pipeline stages
//   Resize(keep_ratio)      mmdet/datasets/pipelines/transforms.py:218-247  (mmcv.imrescale -> cv2.resize INTER_LINEAR, uint8)
//   PatchShuffle            transforms.py:2143-2248                         (the image part: a cyclic shift of columns / rows)
//   RandomFlip(horizontal)  transforms.py:334-470
//   Normalize               transforms.py:652-690                           (mmcv.imnormalize: BGR->RGB, (x - mean) * (1 / std), fp32)
//   Pad(size_divisor) + the loader's merge/pad to the batch's largest image  transforms.py:581-650, datasets/builder.py:236-267
// The bilinear resize restates OpenCV's 8-bit fixed-point path (coefficients scaled by 2^11 and rounded, horizontal pass in
// int32, vertical pass ((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2), so the intermediate is the same uint8
// image the CPU pipeline holds; cv2 is not in this image: that step's parity is UNPINNED (oracle/datapath_oracle.py restates
// the same arithmetic, the rest is exact).  HBM-bound: reads <= 4 source bytes x 3 channels per output pixel (L2-resident
// neighbours), writes 12 bytes.
#include "common.hpp"

// every float operation below is meant as ONE IEEE operation (the CPU libraries this file restates round after each): hipcc's
// default would contract a * b + c into an fma
#pragma clang fp contract(off)

namespace {

// one correctly rounded operation each; defined HERE, below the pragma (HIP's __fadd_rn & co. are plain operators inside a header
// compiled under the default contraction mode: after inlining they fuse into v_fmac_f32 regardless of this file's pragma)
__device__ __forceinline__ float fadd1(float a, float b) { return a + b; }
__device__ __forceinline__ float fsub1(float a, float b) { return a - b; }
__device__ __forceinline__ float fmul1(float a, float b) { return a * b; }
__device__ __forceinline__ float fdiv1(float a, float b) { return a / b; }
__device__ __forceinline__ double dadd1(double a, double b) { return a + b; }
__device__ __forceinline__ double dsub1(double a, double b) { return a - b; }
__device__ __forceinline__ double dmul1(double a, double b) { return a * b; }
__device__ __forceinline__ double ddiv1(double a, double b) { return a / b; }

struct PrepK {
  const dsl_image_prep_item* items;
  int n, hc, wc;
  float* dst;
  unsigned char* dst_u8;      // != NULL: stop in front of Normalize - the uint8 BGR image [n][hc][wc][3] the augmentations work on
};

__device__ __forceinline__ void axis_coef(int d, double scale, int ssize, int& s0, int& s1, int& a0, int& a1) {
  // OpenCV resize.cpp (INTER_LINEAR): fx = (dx + 0.5) * scale - 0.5; sx = floor(fx); fx -= sx; clamped at both borders
  float f = (float)((d + 0.5) * scale - 0.5);
  int s = (int)floorf(f);
  f -= s;
  if (s < 0) { f = 0.f; s = 0; }
  if (s >= ssize - 1) { f = 0.f; s = ssize - 1; }
  s0 = s;
  s1 = s + 1 < ssize ? s + 1 : s;
  // saturate_cast<short>(c * 2048): round to nearest even (cvRound)
  a0 = (int)rintf((1.f - f) * 2048.f);
  a1 = (int)rintf(f * 2048.f);
  if (s + 1 >= ssize) { a0 = 2048; a1 = 0; }     // dx >= xmax: D[dx] = S[sx] * ONE
}

__global__ __launch_bounds__(256) void image_prep_kernel(const PrepK p) {
  const int img = blockIdx.z;
  const dsl_image_prep_item it = p.items[img];
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= p.wc) return;
  float* o = p.dst + (long long)img * 3 * p.hc * p.wc + (long long)y * p.wc + x;
  unsigned char* o8 = p.dst_u8 ? p.dst_u8 + (((long long)img * p.hc + y) * p.wc + x) * 3 : nullptr;
  const long long plane = (long long)p.hc * p.wc;
  if (y >= it.new_h || x >= it.new_w) {        // Pad / merge-pad region
    if (o8) { o8[0] = 0; o8[1] = 0; o8[2] = 0; }
    else { o[0] = 0.f; o[plane] = 0.f; o[2 * plane] = 0.f; }
    return;
  }
  // undo RandomFlip, then PatchShuffle: position in the resized image
  int xr = it.flip ? it.new_w - 1 - x : x, yr = y;
  if (it.ps_mode == 1) xr = xr < it.new_w - it.ps_crop ? xr + it.ps_crop : xr - (it.new_w - it.ps_crop);
  if (it.ps_mode == 2) yr = yr < it.new_h - it.ps_crop ? yr + it.ps_crop : yr - (it.new_h - it.ps_crop);
  int v[3];
  if (it.new_h == it.src_h && it.new_w == it.src_w) {
    const unsigned char* s = it.src + ((long long)yr * it.src_w + xr) * 3;
    v[0] = s[0]; v[1] = s[1]; v[2] = s[2];
  } else {
    int sx0, sx1, ax0, ax1, sy0, sy1, by0, by1;
    axis_coef(xr, 1.0 / ((double)it.new_w / it.src_w), it.src_w, sx0, sx1, ax0, ax1);      // scale = 1. / inv_scale, as OpenCV forms it
    axis_coef(yr, 1.0 / ((double)it.new_h / it.src_h), it.src_h, sy0, sy1, by0, by1);
    const unsigned char* r0 = it.src + (long long)sy0 * it.src_w * 3;
    const unsigned char* r1 = it.src + (long long)sy1 * it.src_w * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const int h0 = r0[sx0 * 3 + c] * ax0 + r0[sx1 * 3 + c] * ax1;
      const int h1 = r1[sx0 * 3 + c] * ax0 + r1[sx1 * 3 + c] * ax1;
      v[c] = (((by0 * (h0 >> 4)) >> 16) + ((by1 * (h1 >> 4)) >> 16) + 2) >> 2;
      v[c] = v[c] < 0 ? 0 : (v[c] > 255 ? 255 : v[c]);
    }
  }
  if (o8) {
    o8[0] = (unsigned char)v[0]; o8[1] = (unsigned char)v[1]; o8[2] = (unsigned char)v[2];
    return;
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const int sc = it.to_rgb ? 2 - c : c;       // output channel c reads source (BGR) channel
    o[c * plane] = fmul1(fsub1((float)v[sc], it.mean[c]), it.inv_std[c]);
  }
}

// ---- the unlabeled stream's augmentations on the uint8 canvases -----------------------------------------------------------
// RandomAugmentBBox_Fast(aug_type='affine') (mmdet/datasets/pipelines/semi_aug.py:344-531: imgaug Affine, global or inside one
// box, order 0 / 1, cval 125) and UBAug (transforms.py:2098-2140: torchvision ColorJitter / RandomGrayscale, PIL GaussianBlur,
// RandomErasing).  imgaug, torchvision and PIL's filters are not in this image and the reference's tests hold no vectors for
// them: every pass below restates the PUBLISHED arithmetic of its library (cited per pass); parity with the libraries is UNPINNED.
// The host (dsl_amd/datapath.py) draws the random parameters and transforms the boxes; one launch = one pass over the batch,
// item.kind selects the pass per image (0: copy).
struct AugK {
  const dsl_aug_item* items;
  int n, hc, wc;
  const unsigned char* src;
  unsigned char* dst;
  unsigned long long* sums;      // per image: sum of the 8-bit luma (contrast's mean), filled by aug_luma_sum_kernel
  unsigned* hist;                // per image and band: 256-bin histogram (AUTOCONTRAST / EQUALIZE), filled by aug_hist_kernel
  unsigned char* lut;            // per image and band: the 256-entry table aug_lut_kernel derives from it
};

__device__ __forceinline__ int luma_u8(int r, int g, int b) {       // PIL "L": (R * 19595 + G * 38470 + B * 7471 + 0x8000) >> 16
  return (r * 19595 + g * 38470 + b * 7471 + 0x8000) >> 16;
}
__device__ __forceinline__ unsigned hash_u32(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

constexpr int kLumaRows = 32;            // image rows per block of aug_luma_sum_kernel: one 64-bit atomic per 256 x 32 pixels
__global__ __launch_bounds__(256) void aug_luma_sum_kernel(const AugK p) {
  const int img = blockIdx.z;
  const dsl_aug_item it = p.items[img];
  if (it.kind != DSL_AUG_CONTRAST) return;
  __shared__ unsigned long long sh[4];
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y0 = blockIdx.y * kLumaRows;
  if (y0 >= it.h || (int)(blockIdx.x * blockDim.x) >= it.w) return;              // block-uniform
  unsigned long long v = 0;
  if (x < it.w) {
    const int y1 = min(y0 + kLumaRows, it.h);
    for (int y = y0; y < y1; ++y) {
      const unsigned char* s = p.src + (((long long)img * p.hc + y) * p.wc + x) * 3;
      v += (unsigned long long)luma_u8(s[0], s[1], s[2]);        // the stored channel order plays RGB (the reference hands
    }                                                            // mmcv's BGR array to ToPILImage as is)
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(p.sums + img, sh[0] + sh[1] + sh[2] + sh[3]);      // integer sum: order-independent
}

// Image.histogram() of the images whose pass is AUTOCONTRAST or EQUALIZE: 256 bins per band; a block folds 256 x 32 pixels in LDS and
// adds its non-empty bins to the image's table (integer sums: order-independent)
__global__ __launch_bounds__(256) void aug_hist_kernel(const AugK p) {
  const int img = blockIdx.z;
  const dsl_aug_item it = p.items[img];
  if (it.kind != DSL_AUG_AUTOCONTRAST && it.kind != DSL_AUG_EQUALIZE) return;
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y0 = blockIdx.y * kLumaRows;
  if (y0 >= it.h || (int)(blockIdx.x * blockDim.x) >= it.w) return;              // block-uniform
  __shared__ unsigned sh[768];
  for (int i = threadIdx.x; i < 768; i += 256) sh[i] = 0u;
  __syncthreads();
  if (x < it.w) {
    const int y1 = min(y0 + kLumaRows, it.h);
    for (int y = y0; y < y1; ++y) {
      const unsigned char* s = p.src + (((long long)img * p.hc + y) * p.wc + x) * 3;
      atomicAdd(&sh[s[0]], 1u); atomicAdd(&sh[256 + s[1]], 1u); atomicAdd(&sh[512 + s[2]], 1u);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 768; i += 256)
    if (sh[i]) atomicAdd(p.hist + (size_t)img * 768 + i, sh[i]);
}
// ImageOps.autocontrast(img) (cutoff 0) / ImageOps.equalize(img), PIL/ImageOps.py: one band's table per 256-thread slice of the block
//  autocontrast: lo / hi = the first / last occupied bin; identity when hi <= lo, else lut[i] = clamp(int(i * scale + offset)) with the
//                DOUBLE scale = 255.0 / (hi - lo), offset = -lo * scale (two rounded operations, as Python evaluates them)
//  equalize:     step = (pixels - count of the last occupied bin) // 255; identity when fewer than two bins are occupied or step == 0,
//                else lut[i] = (step // 2 + sum(h[:i])) // step, clipped to 8 bits by Image.point
__global__ __launch_bounds__(768) void aug_lut_kernel(const AugK p) {
  const int img = blockIdx.x;
  const dsl_aug_item it = p.items[img];
  if (it.kind != DSL_AUG_AUTOCONTRAST && it.kind != DSL_AUG_EQUALIZE) return;
  __shared__ unsigned h[768];
  __shared__ unsigned long long pre[768];
  __shared__ int lo_[3], hi_[3], occ_[3];
  const int t = threadIdx.x, band = t >> 8, i = t & 255;
  h[t] = p.hist[(size_t)img * 768 + t];
  __syncthreads();
  if (i == 0) {                                   // 256 bins: one thread per band walks them once
    int lo = 256, hi = -1, occ = 0;
    unsigned long long run = 0;
    for (int q = 0; q < 256; ++q) {
      const unsigned v = h[band * 256 + q];
      pre[band * 256 + q] = run;
      run += v;
      if (v) { lo = min(lo, q); hi = q; ++occ; }
    }
    lo_[band] = lo; hi_[band] = hi; occ_[band] = occ;
  }
  __syncthreads();
  const int lo = lo_[band], hi = hi_[band];
  int v = i;
  if (it.kind == DSL_AUG_AUTOCONTRAST) {
    if (hi > lo) {
      const double scale = ddiv1(255.0, (double)(hi - lo));
      const double offset = dmul1((double)(-lo), scale);
      v = min(max((int)dadd1(dmul1((double)i, scale), offset), 0), 255);
    }
  } else if (occ_[band] > 1) {
    const unsigned long long total = pre[band * 256 + 255] + h[band * 256 + 255];
    const unsigned long long step = (total - h[band * 256 + hi]) / 255ull;
    if (step) v = (int)min((step / 2ull + pre[band * 256 + i]) / step, 255ull);
  }
  p.lut[(size_t)img * 768 + t] = (unsigned char)v;
}

// Pillow's Image.blend(degenerate, image, f) as ImageEnhance uses it (Blend.c): float32 deg + f * (img - deg), clipped, TRUNCATED
// (explicitly rounded operations: a fused multiply-add would round differently from Pillow's x86 build)
__device__ __forceinline__ int blend_u8(float img, float deg, float f) {
  const float t = fadd1(deg, fmul1(f, fsub1(img, deg)));
  return (int)fminf(fmaxf(t, 0.f), 255.f);
}
// Pillow Convert.c rgb2hsv / hsv2rgb on 8-bit channels, operation for operation (float / double mix as in the C source)
__device__ __forceinline__ void rgb2hsv_u8(int r, int g, int b, int& uh, int& us, int& uv) {
  const int mx = max(r, max(g, b)), mn = min(r, min(g, b));
  uv = mx;
  uh = us = 0;
  if (mx == mn) return;
  const float cr = (float)(mx - mn);
  const float s = fdiv1(cr, (float)mx);
  const float rc = fdiv1((float)(mx - r), cr), gc = fdiv1((float)(mx - g), cr), bc = fdiv1((float)(mx - b), cr);
  float h;
  if (r == mx) h = fsub1(bc, gc);
  else if (g == mx) h = (float)dsub1(dadd1(2.0, (double)rc), (double)bc);
  else h = (float)dsub1(dadd1(4.0, (double)gc), (double)rc);
  const double q = dadd1(ddiv1((double)h, 6.0), 1.0);
  const float hf = (float)(q - floor(q));                      // fmod(q, 1.0) for q > 0: exact
  uh = min(max((int)dmul1((double)hf, 255.0), 0), 255);
  us = min(max((int)dmul1((double)s, 255.0), 0), 255);
}
__device__ __forceinline__ void hsv2rgb_u8(int h, int s, int v, int& r, int& g, int& b) {
  if (s == 0) { r = g = b = v; return; }
  const double fh = ddiv1(dmul1((double)h, 6.0), 255.0), fs = ddiv1((double)s, 255.0);
  const double fi = floor(fh);
  const double f = dsub1(fh, fi);
  const double V = (double)v;
  auto rnd = [](double x) { return min(max((int)floor(dadd1(x, 0.5)), 0), 255); };
  const int p = rnd(dmul1(V, dsub1(1.0, fs)));
  const int q = rnd(dmul1(V, dsub1(1.0, dmul1(fs, f))));
  const int t = rnd(dmul1(V, dsub1(1.0, dmul1(fs, dsub1(1.0, f)))));
  switch (((int)fi) % 6) {
    case 0: r = v; g = t; b = p; break;
    case 1: r = q; g = v; b = p; break;
    case 2: r = p; g = v; b = t; break;
    case 3: r = p; g = q; b = v; break;
    case 4: r = t; g = p; b = v; break;
    default: r = v; g = p; b = q; break;
  }
}

__global__ __launch_bounds__(256) void image_aug_kernel(const AugK p) {
  const int img = blockIdx.z;
  const dsl_aug_item it = p.items[img];
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= p.wc) return;
  const long long base = ((long long)img * p.hc) * p.wc;
  const unsigned char* s = p.src + (base + (long long)y * p.wc + x) * 3;
  unsigned char* d = p.dst + (base + (long long)y * p.wc + x) * 3;
  int r = s[0], g = s[1], b = s[2];         // the stored order plays (R, G, B): see aug_luma_sum_kernel
  const bool inside = x < it.w && y < it.h;
  if (!inside || it.kind == 0) {
    d[0] = r; d[1] = g; d[2] = b;
    return;
  }
  switch (it.kind) {
    case DSL_AUG_AFFINE: {
      // imgaug Affine -> cv2.warpAffine / skimage warp with the INVERSE map (output pixel -> source position), order 0 (nearest,
      // round half to even) or 1 (bilinear), mode constant, cval; inside roi only (in-box affine: source = the same box)
      if (x < it.roi[0] || x >= it.roi[2] || y < it.roi[1] || y >= it.roi[3]) break;
      // (explicitly rounded operations: no fma contraction, so that the positions equal the float32 restatement's bit for bit)
      const float xr = (float)(x - it.roi[0]), yr = (float)(y - it.roi[1]);
      const float xs = fadd1(fadd1(fmul1(it.m[0], xr), fmul1(it.m[1], yr)), it.m[2]);
      const float ys = fadd1(fadd1(fmul1(it.m[3], xr), fmul1(it.m[4], yr)), it.m[5]);
      const int rw = it.roi[2] - it.roi[0], rh = it.roi[3] - it.roi[1];
      auto at = [&](int xi, int yi, int c) -> float {
        if (xi < 0 || yi < 0 || xi >= rw || yi >= rh) return (float)it.cval;
        return (float)p.src[(base + (long long)(yi + it.roi[1]) * p.wc + xi + it.roi[0]) * 3 + c];
      };
      int o[3];
      if (it.order == 0) {
        const int xi = (int)rintf(xs), yi = (int)rintf(ys);
#pragma unroll
        for (int c = 0; c < 3; ++c) o[c] = (int)at(xi, yi, c);
      } else {
        const float xf = floorf(xs), yf = floorf(ys);
        const int x0 = (int)xf, y0 = (int)yf;
        const float fx = xs - xf, fy = ys - yf, gx = fsub1(1.f, fx), gy = fsub1(1.f, fy);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const float top = fadd1(fmul1(at(x0, y0, c), gx), fmul1(at(x0 + 1, y0, c), fx));
          const float bot = fadd1(fmul1(at(x0, y0 + 1, c), gx), fmul1(at(x0 + 1, y0 + 1, c), fx));
          o[c] = (int)fminf(fmaxf(rintf(fadd1(fmul1(top, gy), fmul1(bot, fy))), 0.f), 255.f);
        }
      }
      r = o[0]; g = o[1]; b = o[2];
      break;
    }
    case DSL_AUG_BRIGHTNESS:      // torchvision adjust_brightness = ImageEnhance.Brightness: blend with black
      r = blend_u8((float)r, 0.f, it.f[0]); g = blend_u8((float)g, 0.f, it.f[0]); b = blend_u8((float)b, 0.f, it.f[0]);
      break;
    case DSL_AUG_CONTRAST: {      // ImageEnhance.Contrast: blend with int(mean of the L image + 0.5)
      const float mean = (float)(int)((double)p.sums[img] / ((double)it.w * (double)it.h) + 0.5);
      r = blend_u8((float)r, mean, it.f[0]); g = blend_u8((float)g, mean, it.f[0]); b = blend_u8((float)b, mean, it.f[0]);
      break;
    }
    case DSL_AUG_SATURATION: {    // ImageEnhance.Color: blend with the pixel's own L
      const float l = (float)luma_u8(r, g, b);
      r = blend_u8((float)r, l, it.f[0]); g = blend_u8((float)g, l, it.f[0]); b = blend_u8((float)b, l, it.f[0]);
      break;
    }
    case DSL_AUG_HUE: {           // adjust_hue: 8-bit HSV, H += uint8(factor * 255) (wraps), back to RGB; f[0] = int(factor * 255),
                                  // formed on the host in double precision as torchvision does
      int uh, us, uv;
      rgb2hsv_u8(r, g, b, uh, us, uv);
      uh = (uh + (int)it.f[0] + 512) & 255;
      hsv2rgb_u8(uh, us, uv, r, g, b);
      break;
    }
    case DSL_AUG_GRAY: {          // RandomGrayscale: convert('L') replicated to the three channels
      const int l = luma_u8(r, g, b);
      r = g = b = l;
      break;
    }
    case DSL_AUG_BLUR_H:
    case DSL_AUG_BLUR_V: {        // one pass of Pillow's ImagingLineBoxBlur8 in direct form (BoxBlur.c): f[0] = the fractional box radius
      const float fr = it.f[0];   // of ImageFilter.GaussianBlur (3 horizontal + 3 vertical passes); 24-bit fixed point, edges replicated
      const int rad = (int)fr;
      const unsigned ww = (unsigned)fdiv1(16777216.f, fadd1(fmul1(fr, 2.f), 1.f));
      const unsigned fw = ((1u << 24) - (unsigned)(rad * 2 + 1) * ww) / 2u;
      const bool hz = it.kind == DSL_AUG_BLUR_H;
      const int pos = hz ? x : y, last = (hz ? it.w : it.h) - 1;
      unsigned acc[3] = {0u, 0u, 0u}, far_[3] = {0u, 0u, 0u};
      auto px = [&](int q) -> const unsigned char* {
        q = min(max(q, 0), last);
        return p.src + (base + (hz ? (long long)y * p.wc + q : (long long)q * p.wc + x)) * 3;
      };
      for (int k = -rad; k <= rad; ++k) {
        const unsigned char* q = px(pos + k);
        acc[0] += q[0]; acc[1] += q[1]; acc[2] += q[2];
      }
      {
        const unsigned char *qa = px(pos - rad - 1), *qb = px(pos + rad + 1);
        far_[0] = qa[0] + qb[0]; far_[1] = qa[1] + qb[1]; far_[2] = qa[2] + qb[2];
      }
      r = (int)((acc[0] * ww + far_[0] * fw + (1u << 23)) >> 24);
      g = (int)((acc[1] * ww + far_[1] * fw + (1u << 23)) >> 24);
      b = (int)((acc[2] * ww + far_[2] * fw + (1u << 23)) >> 24);
      break;
    }
    case DSL_AUG_AUTOCONTRAST:    // RandAug AutoContrast / Equalize (autoaug_fast.py:219-224): per-band tables out of aug_lut_kernel
    case DSL_AUG_EQUALIZE: {
      const unsigned char* lut = p.lut + (size_t)img * 768;
      r = lut[r]; g = lut[256 + g]; b = lut[512 + b];
      break;
    }
    case DSL_AUG_SOLARIZE: {      // ImageOps.solarize(img, threshold): f[0] = 256 - int(level * 256 / 10) (autoaug_fast.py:371-372)
      const int th = (int)it.f[0];
      r = r < th ? r : 255 - r; g = g < th ? g : 255 - g; b = b < th ? b : 255 - b;
      break;
    }
    case DSL_AUG_POSTERIZE: {     // ImageOps.posterize(img, bits): f[0] = bits = 4 - int(level * 4 / 10) (autoaug_fast.py:244-247)
      const int mask = ~((1 << (8 - (int)it.f[0])) - 1) & 255;
      r &= mask; g &= mask; b &= mask;
      break;
    }
    case DSL_AUG_SHARPNESS: {     // ImageEnhance.Sharpness: blend(img.filter(SMOOTH), img, f).  Filter.c ImagingFilter3x3 on 8-bit bands:
                                  // float32 weights 1 / 13 and 5 / 13, 0.5 + row(y + 1) + row(y) + row(y - 1), each row's three products
                                  // added left to right first, truncated; the image's one-pixel frame keeps its input values
      int deg[3] = {r, g, b};
      if (x > 0 && y > 0 && x < it.w - 1 && y < it.h - 1) {
        const float k1 = fdiv1(1.f, 13.f), k5 = fdiv1(5.f, 13.f);
        float ss[3] = {0.5f, 0.5f, 0.5f};
#pragma unroll
        for (int dy = 1; dy >= -1; --dy) {
          const unsigned char* q = p.src + (base + (long long)(y + dy) * p.wc + x - 1) * 3;
          const float kc = dy == 0 ? k5 : k1;
#pragma unroll
          for (int c = 0; c < 3; ++c)
            ss[c] = fadd1(ss[c], fadd1(fadd1(fmul1((float)q[c], k1), fmul1((float)q[3 + c], kc)), fmul1((float)q[6 + c], k1)));
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) deg[c] = (int)fminf(fmaxf(ss[c], 0.f), 255.f);
      }
      r = blend_u8((float)r, (float)deg[0], it.f[0]); g = blend_u8((float)g, (float)deg[1], it.f[0]);
      b = blend_u8((float)b, (float)deg[2], it.f[0]);
      break;
    }
    case DSL_AUG_ERASE: {         // RandomErasing(value='random'): N(0, 1) noise in [0, 1] units, then ToPILImage's mul(255).byte()
      for (int k = 0; k < 3; ++k) {
        if (it.rect[k][2] <= it.rect[k][0]) continue;
        if (x < it.rect[k][0] || x >= it.rect[k][2] || y < it.rect[k][1] || y >= it.rect[k][3]) continue;
        int o[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const unsigned h1 = hash_u32(it.seed ^ (unsigned)(((y * 8192 + x) * 3 + c) * 2 + 1) ^ (unsigned)(k * 0x9e3779b9u));
          const unsigned h2 = hash_u32(h1 ^ 0x85ebca6bu);
          const float u1 = ((float)(h1 >> 8) + 1.f) * (1.f / 16777217.f), u2 = (float)(h2 >> 8) * (1.f / 16777216.f);
          const float z = sqrtf(-2.f * __logf(u1)) * __cosf(6.2831853f * u2);
          o[c] = (int)(z * 255.f) & 255;               // float -> uint8 of an out-of-range value wraps (two's complement)
        }
        r = o[0]; g = o[1]; b = o[2];
      }
      break;
    }
    default: break;
  }
  d[0] = (unsigned char)r; d[1] = (unsigned char)g; d[2] = (unsigned char)b;
}

// Normalize + Pad of a uint8 canvas (the tail of image_prep_kernel for batches that went through the augmentation passes)
__global__ __launch_bounds__(256) void image_normalize_kernel(const PrepK p, const unsigned char* __restrict__ src) {
  const int img = blockIdx.z;
  const dsl_image_prep_item it = p.items[img];
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= p.wc) return;
  float* o = p.dst + (long long)img * 3 * p.hc * p.wc + (long long)y * p.wc + x;
  const long long plane = (long long)p.hc * p.wc;
  if (y >= it.new_h || x >= it.new_w) {
    o[0] = 0.f; o[plane] = 0.f; o[2 * plane] = 0.f;
    return;
  }
  const unsigned char* s = src + (((long long)img * p.hc + y) * p.wc + x) * 3;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const int sc = it.to_rgb ? 2 - c : c;
    o[c * plane] = fmul1(fsub1((float)s[sc], it.mean[c]), it.inv_std[c]);
  }
}

}  // namespace

extern "C" int dsl_image_prep(const dsl_image_prep_item* items_dev, int n, float* dst, int hc, int wc, void* stream) {
  DSL_CHECK(items_dev && dst && n > 0 && hc > 0 && wc > 0, "dsl_image_prep: bad arguments");
  PrepK k{items_dev, n, hc, wc, dst, nullptr};
  hipLaunchKernelGGL(image_prep_kernel, dim3((wc + 255) / 256, hc, n), dim3(256), 0, (hipStream_t)stream, k);
  DSL_LAUNCH_CHECK("image_prep_kernel");
  return 0;
}

extern "C" int dsl_image_prep_u8(const dsl_image_prep_item* items_dev, int n, unsigned char* dst_u8, int hc, int wc, void* stream) {
  DSL_CHECK(items_dev && dst_u8 && n > 0 && hc > 0 && wc > 0, "dsl_image_prep_u8: bad arguments");
  PrepK k{items_dev, n, hc, wc, nullptr, dst_u8};
  hipLaunchKernelGGL(image_prep_kernel, dim3((wc + 255) / 256, hc, n), dim3(256), 0, (hipStream_t)stream, k);
  DSL_LAUNCH_CHECK("image_prep_kernel (u8)");
  return 0;
}

extern "C" size_t dsl_image_aug_scratch_bytes(int n) {
  return n > 0 ? (size_t)n * (8 + 768 * 4 + 768) : 0;      // luma sums [n] u64 | histograms [n][3][256] u32 | tables [n][3][256] u8
}

extern "C" int dsl_image_aug(const dsl_aug_item* items_dev, int n, const unsigned char* src, unsigned char* dst, int hc, int wc,
                             void* scratch, int need_stats, void* stream) {
  DSL_CHECK(items_dev && src && dst && src != dst && n > 0 && hc > 0 && wc > 0 && wc < 8192, "dsl_image_aug: bad arguments");
  DSL_CHECK(!need_stats || (scratch && ((uintptr_t)scratch & 7) == 0),
            "dsl_image_aug: CONTRAST / AUTOCONTRAST / EQUALIZE passes need the 8-byte aligned dsl_image_aug_scratch_bytes(n) scratch");
  unsigned char* sc = (unsigned char*)scratch;
  AugK k{items_dev, n, hc, wc, src, dst, (unsigned long long*)sc, (unsigned*)(sc ? sc + 8 * (size_t)n : nullptr),
         sc ? sc + (8 + 768 * 4) * (size_t)n : nullptr};
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((wc + 255) / 256, hc, n);
  const dim3 sgrid(grid.x, (hc + kLumaRows - 1) / kLumaRows, n);
  if (need_stats & 1) {
    if (hipMemsetAsync(k.sums, 0, 8 * (size_t)n, st) != hipSuccess) {
      dsl_set_error("dsl_image_aug: memset failed");
      return -2;
    }
    hipLaunchKernelGGL(aug_luma_sum_kernel, sgrid, dim3(256), 0, st, k);
  }
  if (need_stats & 2) {
    if (hipMemsetAsync(k.hist, 0, 768 * 4 * (size_t)n, st) != hipSuccess) {
      dsl_set_error("dsl_image_aug: memset failed");
      return -2;
    }
    hipLaunchKernelGGL(aug_hist_kernel, sgrid, dim3(256), 0, st, k);
    hipLaunchKernelGGL(aug_lut_kernel, dim3(n), dim3(768), 0, st, k);
  }
  hipLaunchKernelGGL(image_aug_kernel, grid, dim3(256), 0, st, k);
  DSL_LAUNCH_CHECK("image_aug_kernel");
  return 0;
}

extern "C" int dsl_image_normalize(const unsigned char* src_u8, const dsl_image_prep_item* items_dev, int n, float* dst, int hc, int wc,
                                   void* stream) {
  DSL_CHECK(src_u8 && items_dev && dst && n > 0 && hc > 0 && wc > 0, "dsl_image_normalize: bad arguments");
  PrepK k{items_dev, n, hc, wc, dst, nullptr};
  hipLaunchKernelGGL(image_normalize_kernel, dim3((wc + 255) / 256, hc, n), dim3(256), 0, (hipStream_t)stream, k, src_u8);
  DSL_LAUNCH_CHECK("image_normalize_kernel");
  return 0;
}
