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

namespace {

struct PrepK {
  const dsl_image_prep_item* items;
  int n, hc, wc;
  float* dst;
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
  const long long plane = (long long)p.hc * p.wc;
  if (y >= it.new_h || x >= it.new_w) {        // Pad / merge-pad region
    o[0] = 0.f; o[plane] = 0.f; o[2 * plane] = 0.f;
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
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const int sc = it.to_rgb ? 2 - c : c;       // output channel c reads source (BGR) channel
    o[c * plane] = __fmul_rn(__fsub_rn((float)v[sc], it.mean[c]), it.inv_std[c]);
  }
}

}  // namespace

extern "C" int dsl_image_prep(const dsl_image_prep_item* items_dev, int n, float* dst, int hc, int wc, void* stream) {
  DSL_CHECK(items_dev && dst && n > 0 && hc > 0 && wc > 0, "dsl_image_prep: bad arguments");
  PrepK k{items_dev, n, hc, wc, dst};
  hipLaunchKernelGGL(image_prep_kernel, dim3((wc + 255) / 256, hc, n), dim3(256), 0, (hipStream_t)stream, k);
  DSL_LAUNCH_CHECK("image_prep_kernel");
  return 0;
}
