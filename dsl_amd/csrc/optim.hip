// Flat-buffer optimizer / EMA kernels (HBM-bound, float4).  Replace torch.optim.SGD +
// clip_grad_norm_ (mmcv OptimizerHook wired at mmdet/apis/train.py:111,157-166) and the EMA teacher
// lerp of mmdet/runner/hooks/semi_epoch_based_runner.py:368-409 of the reference.
#include "common.hpp"

namespace {

__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ x, long long n, float* out) {
  __shared__ float sh[16];
  float s = 0.f;
  const long long n4 = n / 4;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(x + 4 * i);
    s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
  }
  if (blockIdx.x == 0 && threadIdx.x == 0)
    for (long long i = n4 * 4; i < n; ++i) s += x[i] * x[i];
  s = block_sum(s, sh);
  if (threadIdx.x == 0) atomicAdd(out, s);
}

// deterministic form: block partials in block order, folded by one workgroup (the clip coefficient of every run is the same)
__global__ __launch_bounds__(256) void sumsq_part_kernel(const float* __restrict__ x, long long n, float* __restrict__ part) {
  __shared__ float sh[16];
  float s = 0.f;
  const long long n4 = n / 4;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(x + 4 * i);
    s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
  }
  if (blockIdx.x == 0 && threadIdx.x == 0)
    for (long long i = n4 * 4; i < n; ++i) s += x[i] * x[i];
  s = block_sum(s, sh);
  if (threadIdx.x == 0) part[blockIdx.x] = s;
}
__global__ __launch_bounds__(256) void sumsq_fold_kernel(const float* __restrict__ part, int nb, float* out) {
  __shared__ float sh[16];
  float s = 0.f;
  for (int i = threadIdx.x; i < nb; i += blockDim.x) s += part[i];
  s = block_sum(s, sh);
  if (threadIdx.x == 0) *out = s;
}

__global__ __launch_bounds__(256) void sgd_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                  float* __restrict__ m, uint16_t* __restrict__ p16,
                                                  const uint8_t* __restrict__ group, long long n, float lr,
                                                  float mom, float wd, float blr, float bwd,
                                                  const float* gnorm_sq, float max_norm, int first) {
  float clip = 1.f;
  if (gnorm_sq) clip = fminf(max_norm / (sqrtf(*gnorm_sq) + 1e-6f), 1.f);
  const long long n4 = n / 4;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    f32x4 pv = *reinterpret_cast<const f32x4*>(p + 4 * i);
    const f32x4 gv = *reinterpret_cast<const f32x4*>(g + 4 * i);
    f32x4 mv = *reinterpret_cast<const f32x4*>(m + 4 * i);
    const uint32_t gr = *reinterpret_cast<const uint32_t*>(group + 4 * i);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const bool b = ((gr >> (8 * e)) & 0xff) != 0;
      const float l = b ? lr * blr : lr, w = b ? wd * bwd : wd;
      const float d = gv[e] * clip + w * pv[e];
      mv[e] = first ? d : mom * mv[e] + d;
      pv[e] -= l * mv[e];
    }
    *reinterpret_cast<f32x4*>(p + 4 * i) = pv;
    *reinterpret_cast<f32x4*>(m + 4 * i) = mv;
    if (p16) *reinterpret_cast<u32x2*>(p16 + 4 * i) = u32x2{pack2bf(pv[0], pv[1]), pack2bf(pv[2], pv[3])};
  }
}

__global__ void ema_kernel(float* __restrict__ t, const float* __restrict__ s, long long n, float keep, uint16_t* __restrict__ t16) {
  const long long n4 = n / 4;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    f32x4 tv = *reinterpret_cast<const f32x4*>(t + 4 * i);
    const f32x4 sv = *reinterpret_cast<const f32x4*>(s + 4 * i);
#pragma unroll
    for (int e = 0; e < 4; ++e) tv[e] = tv[e] * keep + sv[e] * (1.f - keep);
    *reinterpret_cast<f32x4*>(t + 4 * i) = tv;
    if (t16) *reinterpret_cast<u32x2*>(t16 + 4 * i) = u32x2{pack2bf(tv[0], tv[1]), pack2bf(tv[2], tv[3])};     // the forward copy
  }
}

__global__ void cast_kernel(const float* __restrict__ x, uint16_t* __restrict__ y, long long n) {
  const long long n4 = n / 4;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(x + 4 * i);
    *reinterpret_cast<u32x2*>(y + 4 * i) = u32x2{pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
  }
}

// KRSC fp32 [cout][taps][cin] -> CRSK bf16 [cin][taps][cout_pad], out[ci][t][co] = w[co][t][ci]*scale[co]
__global__ void pack_dgrad_kernel(const float* __restrict__ w, const float* __restrict__ scale,
                                  uint16_t* __restrict__ out, int cout, int cout_pad, int taps, int cin) {
  __shared__ float tile[32][33];
  const int t = blockIdx.z;
  const int ci0 = blockIdx.x * 32, co0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
  for (int r = ty; r < 32; r += 8) {
    const int co = co0 + r, ci = ci0 + tx;
    float v = 0.f;
    if (co < cout && ci < cin) {
      v = w[((long long)co * taps + t) * cin + ci];
      if (scale) v *= scale[co];
    }
    tile[r][tx] = v;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int ci = ci0 + r, co = co0 + tx;
    if (ci < cin && co < cout_pad) out[((long long)ci * taps + t) * cout_pad + co] = f2bf(tile[tx][r]);
  }
}

// one 64x64 (cout x cin) tile of one tap per block: 256-byte fp32 rows in, 128-byte bf16 rows out
__global__ __launch_bounds__(256) void pack_dgrad_batched_kernel(const dsl_pack_item* __restrict__ items, int n) {
  __shared__ float tile[64][65];
  // locate this block's conv: binary search over the block_start prefix sums (a linear scan is a chain of up to
  // n dependent global loads per block)
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if ((int)blockIdx.x >= items[mid].block_start) lo = mid; else hi = mid - 1;
  }
  const dsl_pack_item I = items[lo];
  int b = blockIdx.x - I.block_start;
  const int tci = b % I.tiles_ci;
  b /= I.tiles_ci;
  const int tco = b % I.tiles_co;
  const int t = b / I.tiles_co;
  const int ci0 = tci * 64, co0 = tco * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  // tap selection (dsl_pack_item.tapmap): out tap t reads source tap st of a weight with staps taps; st >= staps = a zero tap
  const int staps = I.tapmap ? ((I.tapmap >> 16) & 0xff) : I.taps;
  const int st = I.tapmap ? ((I.tapmap >> (4 * t)) & 0xf) : t;
#pragma unroll 4
  for (int r = ty; r < 64; r += 4) {
    const int co = co0 + r, ci = ci0 + tx;
    float v = 0.f;
    if (co < I.cout && ci < I.cin && st < staps) {
      v = I.w[((long long)co * staps + st) * I.cin + ci];
      if (I.scale) v *= I.scale[co];
    }
    tile[r][tx] = v;
  }
  __syncthreads();
  uint16_t* out = (uint16_t*)I.out;
#pragma unroll 4
  for (int r = ty; r < 64; r += 4) {
    const int ci = ci0 + r, co = co0 + tx;
    if (ci < I.cin && co < I.cout_pad) out[((long long)ci * I.taps + t) * I.cout_pad + co] = f2bf(tile[tx][r]);
  }
}

int nblocks(long long n4, int cap) {
  long long b = (n4 + 255) / 256;
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return (int)b;
}

}  // namespace

extern "C" int dsl_sumsq(const float* x, long n, float* out, void* stream) {
  DSL_CHECK(x && out && n >= 0, "dsl_sumsq: bad arguments");
  hipLaunchKernelGGL(sumsq_kernel, dim3(nblocks(n / 4, 1024)), dim3(256), 0, (hipStream_t)stream, x, (long long)n, out);
  DSL_LAUNCH_CHECK("sumsq_kernel");
  return 0;
}

extern "C" int dsl_sumsq_det(const float* x, long n, float* out, float* workspace, void* stream) {
  DSL_CHECK(x && out && workspace && n >= 0, "dsl_sumsq_det: bad arguments");
  const int nb = nblocks(n / 4, 1024);
  hipLaunchKernelGGL(sumsq_part_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, x, (long long)n, workspace);
  hipLaunchKernelGGL(sumsq_fold_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float*)workspace, nb, out);
  DSL_LAUNCH_CHECK("sumsq_part_kernel");
  return 0;
}

extern "C" int dsl_sgd_step(float* p, const float* g, float* m, void* p16, const uint8_t* group, long n, float lr,
                            float momentum, float wd, float bias_lr_mult, float bias_decay_mult,
                            const float* gnorm_sq, float max_norm, int first_step, void* stream) {
  DSL_CHECK(p && g && m && group && n % 4 == 0, "dsl_sgd_step: bad arguments (n must be a multiple of 4)");
  hipLaunchKernelGGL(sgd_kernel, dim3(nblocks(n / 4, 4096)), dim3(256), 0, (hipStream_t)stream, p, g, m,
                     (uint16_t*)p16, group, (long long)n, lr, momentum, wd, bias_lr_mult, bias_decay_mult, gnorm_sq,
                     max_norm, first_step);
  DSL_LAUNCH_CHECK("sgd_kernel");
  return 0;
}

extern "C" int dsl_ema_lerp(float* teacher, const float* student, long n, float keep, void* stream) {
  DSL_CHECK(teacher && student && n % 4 == 0, "dsl_ema_lerp: bad arguments");
  hipLaunchKernelGGL(ema_kernel, dim3(nblocks(n / 4, 4096)), dim3(256), 0, (hipStream_t)stream, teacher, student,
                     (long long)n, keep, (uint16_t*)nullptr);
  DSL_LAUNCH_CHECK("ema_kernel");
  return 0;
}

extern "C" int dsl_ema_lerp_bf16(float* teacher, const float* student, void* teacher_bf16, long n, float keep, void* stream) {
  DSL_CHECK(teacher && student && teacher_bf16 && n % 4 == 0, "dsl_ema_lerp_bf16: bad arguments");
  hipLaunchKernelGGL(ema_kernel, dim3(nblocks(n / 4, 4096)), dim3(256), 0, (hipStream_t)stream, teacher, student,
                     (long long)n, keep, (uint16_t*)teacher_bf16);
  DSL_LAUNCH_CHECK("ema_kernel");
  return 0;
}

extern "C" int dsl_cast_bf16(const float* x, void* y, long n, void* stream) {
  DSL_CHECK(x && y && n % 4 == 0, "dsl_cast_bf16: bad arguments");
  hipLaunchKernelGGL(cast_kernel, dim3(nblocks(n / 4, 4096)), dim3(256), 0, (hipStream_t)stream, x, (uint16_t*)y,
                     (long long)n);
  DSL_LAUNCH_CHECK("cast_kernel");
  return 0;
}

// bf16 -> fp32 (the way back of a bf16 gradient bucket, dsl_amd/parallel.py grad_dtype='bf16')
__global__ void uncast_kernel(const uint16_t* __restrict__ x, float* __restrict__ y, long long n) {
  const long long n4 = n / 4;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    const u32x2 v = *reinterpret_cast<const u32x2*>(x + 4 * i);
    *reinterpret_cast<f32x4*>(y + 4 * i) = f32x4{bflo(v[0]), bfhi(v[0]), bflo(v[1]), bfhi(v[1])};
  }
}
extern "C" int dsl_cast_f32(const void* x_bf16, float* y, long n, void* stream) {
  DSL_CHECK(x_bf16 && y && n % 4 == 0, "dsl_cast_f32: bad arguments");
  hipLaunchKernelGGL(uncast_kernel, dim3(nblocks(n / 4, 4096)), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)x_bf16, y, (long long)n);
  DSL_LAUNCH_CHECK("uncast_kernel");
  return 0;
}

// Gradient norm in pieces (data parallel with clipping): dsl_sumsq_partial writes DSL_SUMSQ_PARTS block sums of x[0 .. n) to
// `partials` - one call per gradient bucket, on the stream that bucket's all-reduce completes on - and dsl_sumsq_fold adds
// n_partials of them up in index order; only the fold and the optimizer step remain behind the last bucket.  Fixed order, no atomics.
extern "C" int dsl_sumsq_partial(const float* x, long n, float* partials, void* stream) {
  DSL_CHECK(x && partials && n >= 0, "dsl_sumsq_partial: bad arguments");
  hipLaunchKernelGGL(sumsq_part_kernel, dim3(DSL_SUMSQ_PARTS), dim3(256), 0, (hipStream_t)stream, x, (long long)n, partials);
  DSL_LAUNCH_CHECK("sumsq_part_kernel");
  return 0;
}
extern "C" int dsl_sumsq_fold(const float* partials, int n_partials, float* out, void* stream) {
  DSL_CHECK(partials && out && n_partials > 0, "dsl_sumsq_fold: bad arguments");
  hipLaunchKernelGGL(sumsq_fold_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, partials, n_partials, out);
  DSL_LAUNCH_CHECK("sumsq_fold_kernel");
  return 0;
}

extern "C" int dsl_pack_dgrad(const float* w, const float* scale, void* out, int cout, int cout_pad, int taps, int cin,
                              void* stream) {
  DSL_CHECK(w && out && cout > 0 && cout_pad >= cout && taps > 0 && cin > 0, "dsl_pack_dgrad: bad arguments");
  dim3 grid((cin + 31) / 32, (cout_pad + 31) / 32, taps);
  hipLaunchKernelGGL(pack_dgrad_kernel, grid, dim3(256), 0, (hipStream_t)stream, w, scale, (uint16_t*)out, cout,
                     cout_pad, taps, cin);
  DSL_LAUNCH_CHECK("pack_dgrad_kernel");
  return 0;
}

extern "C" int dsl_pack_dgrad_batched(const dsl_pack_item* items_dev, int n, int total_blocks, void* stream) {
  DSL_CHECK(items_dev && n > 0 && total_blocks > 0, "dsl_pack_dgrad_batched: bad arguments");
  hipLaunchKernelGGL(pack_dgrad_batched_kernel, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, items_dev, n);
  DSL_LAUNCH_CHECK("pack_dgrad_batched_kernel");
  return 0;
}

// ---- fp8 (OCP e4m3) quantisation for the fp8 forward convolutions (conv.hip conv_f8_kernel) --------------------------------
namespace {
__global__ __launch_bounds__(256) void quant_fp8_kernel(const uint16_t* __restrict__ x, uint8_t* __restrict__ y, long long rows, int c16,
                                                        int ld_x, int c, float scale) {
  const long long total = rows * c16;       // 16 elements per thread-iteration: 32 bytes in, 16 bytes out
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long r = i / c16;
    const int ch = (int)(i - r * c16) * 16;
    const u32x4 v0 = *reinterpret_cast<const u32x4*>(x + r * ld_x + ch), v1 = *reinterpret_cast<const u32x4*>(x + r * ld_x + ch + 8);
    u32x4 o;
    o[0] = cvt4_fp8(bflo(v0[0]) * scale, bfhi(v0[0]) * scale, bflo(v0[1]) * scale, bfhi(v0[1]) * scale);
    o[1] = cvt4_fp8(bflo(v0[2]) * scale, bfhi(v0[2]) * scale, bflo(v0[3]) * scale, bfhi(v0[3]) * scale);
    o[2] = cvt4_fp8(bflo(v1[0]) * scale, bfhi(v1[0]) * scale, bflo(v1[1]) * scale, bfhi(v1[1]) * scale);
    o[3] = cvt4_fp8(bflo(v1[2]) * scale, bfhi(v1[2]) * scale, bflo(v1[3]) * scale, bfhi(v1[3]) * scale);
    *reinterpret_cast<u32x4*>(y + r * c + ch) = o;
  }
}
// one workgroup per output channel: row maximum, then the scaled row
__global__ __launch_bounds__(256) void quant_fp8_weights_kernel(const float* __restrict__ w, uint8_t* __restrict__ w8, float* __restrict__ comb,
                                                                const float* __restrict__ bn_scale, int cout, int k, float inv_act_scale) {
  __shared__ float sh[16];
  const int co = blockIdx.x;
  uint8_t* out = w8 + (long long)co * k;
  if (co >= cout) {
    for (int i = threadIdx.x * 4; i < k; i += blockDim.x * 4) *reinterpret_cast<uint32_t*>(out + i) = 0u;
    if (threadIdx.x == 0) comb[co] = 0.f;
    return;
  }
  const float* row = w + (long long)co * k;
  float m = 0.f;
  for (int i = threadIdx.x * 4; i < k; i += blockDim.x * 4) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(row + i);
    m = fmaxf(fmaxf(m, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = m;
  __syncthreads();
  m = fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
  const float s = m > 0.f ? 448.f / m : 1.f;
  for (int i = threadIdx.x * 4; i < k; i += blockDim.x * 4) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(row + i);
    *reinterpret_cast<uint32_t*>(out + i) = cvt4_fp8(v[0] * s, v[1] * s, v[2] * s, v[3] * s);
  }
  if (threadIdx.x == 0) comb[co] = inv_act_scale / s * (bn_scale ? bn_scale[co] : 1.f);
}
// per-block maximum of |x| over a bf16 [rows][ld_x] tensor (first c columns): partials[block]; no atomics, nothing to clear
__global__ __launch_bounds__(256) void absmax_kernel(const uint16_t* __restrict__ x, long long rows, int c8, int ld_x, float* __restrict__ partials) {
  __shared__ float sh[4];
  const long long total = rows * c8;
  float m = 0.f;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long r = i / c8;
    const u32x4 v = *reinterpret_cast<const u32x4*>(x + r * ld_x + (int)(i - r * c8) * 8);
#pragma unroll
    for (int e = 0; e < 4; ++e) m = fmaxf(m, fmaxf(fabsf(bflo(v[e])), fabsf(bfhi(v[e]))));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) partials[blockIdx.x] = fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
}
__device__ __forceinline__ float fold_absmax(const float* __restrict__ partials, int n, float* sh) {
  float m = 0.f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) m = fmaxf(m, partials[i]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = m;
  __syncthreads();
  m = sh[0];
  for (int w = 1; w < (int)(blockDim.x >> 6); ++w) m = fmaxf(m, sh[w]);
  return m;
}
// dynamic per-tensor scale: scale = 448 / max|x| of THIS tensor (from absmax_kernel's partials): nothing saturates
__global__ __launch_bounds__(256) void quant_fp8_dyn_kernel(const uint16_t* __restrict__ x, uint8_t* __restrict__ y, long long rows, int c16,
                                                            int ld_x, int c, const float* __restrict__ partials, int n_partials) {
  __shared__ float sh[4];
  const float amax = fold_absmax(partials, n_partials, sh);
  const float scale = amax > 0.f ? 448.f / amax : 1.f;
  const long long total = rows * c16;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long r = i / c16;
    const int ch = (int)(i - r * c16) * 16;
    const u32x4 v0 = *reinterpret_cast<const u32x4*>(x + r * ld_x + ch), v1 = *reinterpret_cast<const u32x4*>(x + r * ld_x + ch + 8);
    u32x4 o;
    o[0] = cvt4_fp8(bflo(v0[0]) * scale, bfhi(v0[0]) * scale, bflo(v0[1]) * scale, bfhi(v0[1]) * scale);
    o[1] = cvt4_fp8(bflo(v0[2]) * scale, bfhi(v0[2]) * scale, bflo(v0[3]) * scale, bfhi(v0[3]) * scale);
    o[2] = cvt4_fp8(bflo(v1[0]) * scale, bfhi(v1[0]) * scale, bflo(v1[1]) * scale, bfhi(v1[1]) * scale);
    o[3] = cvt4_fp8(bflo(v1[2]) * scale, bfhi(v1[2]) * scale, bflo(v1[3]) * scale, bfhi(v1[3]) * scale);
    *reinterpret_cast<u32x4*>(y + r * c + ch) = o;
  }
}
// the fp8 convolution's epilogue scale: comb[co] = winv[co] (1 / weight scale, x BatchNorm scale) x amax / 448 (1 / activation scale)
__global__ __launch_bounds__(256) void fp8_comb_kernel(const float* __restrict__ winv, float* __restrict__ comb, int n, const float* __restrict__ partials,
                                                       int n_partials) {
  __shared__ float sh[4];
  const float amax = fold_absmax(partials, n_partials, sh);
  const float inv = amax > 0.f ? amax / 448.f : 1.f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) comb[i] = winv[i] * inv;
}
// Delayed scaling: the weights of n convolutions, their epilogue scales and their inputs' quantisation scales in ONE launch per step
// (grid: cout_pad x items).  The input maximum is the previous step's (block maxima left by the producer's pass), widened by `margin`.
__global__ __launch_bounds__(256) void fp8_prep_kernel(const dsl_fp8_prep_item* __restrict__ items, int k, float margin) {
  __shared__ float sh[16];
  const dsl_fp8_prep_item it = items[blockIdx.y];
  const int co = blockIdx.x;
  const float amax = fold_absmax(it.amax, it.n_amax, sh) * margin;
  __syncthreads();
  if (co == 0 && threadIdx.x == 0) it.scale[0] = amax > 0.f ? 448.f / amax : 1.f;
  const float inv_act = amax > 0.f ? amax / 448.f : 1.f;
  uint8_t* out = (uint8_t*)it.w8 + (long long)co * k;
  if (co >= it.cout) {
    for (int i = threadIdx.x * 4; i < k; i += blockDim.x * 4) *reinterpret_cast<uint32_t*>(out + i) = 0u;
    if (threadIdx.x == 0) it.comb[co] = 0.f;
    return;
  }
  const float* row = it.w + (long long)co * k;
  float m = 0.f;
  for (int i = threadIdx.x * 4; i < k; i += blockDim.x * 4) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(row + i);
    m = fmaxf(fmaxf(m, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0) sh[8 + (threadIdx.x >> 6)] = m;
  __syncthreads();
  m = fmaxf(fmaxf(sh[8], sh[9]), fmaxf(sh[10], sh[11]));
  const float s = m > 0.f ? 448.f / m : 1.f;
  for (int i = threadIdx.x * 4; i < k; i += blockDim.x * 4) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(row + i);
    *reinterpret_cast<uint32_t*>(out + i) = cvt4_fp8(v[0] * s, v[1] * s, v[2] * s, v[3] * s);
  }
  if (threadIdx.x == 0) it.comb[co] = inv_act / s;
}
// one pass: y = e4m3(x * scale[0]) and this block's max|x| for the next step's scale
__global__ __launch_bounds__(256) void quant_fp8_delayed_kernel(const uint16_t* __restrict__ x, uint8_t* __restrict__ y, long long rows, int c16,
                                                                int ld_x, int c, const float* __restrict__ scale_dev, float* __restrict__ partials) {
  __shared__ float sh[4];
  const float scale = scale_dev[0];
  const long long total = rows * c16;
  float m = 0.f;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long r = i / c16;
    const int ch = (int)(i - r * c16) * 16;
    const u32x4 v0 = *reinterpret_cast<const u32x4*>(x + r * ld_x + ch), v1 = *reinterpret_cast<const u32x4*>(x + r * ld_x + ch + 8);
#pragma unroll
    for (int e = 0; e < 4; ++e)
      m = fmaxf(fmaxf(m, fmaxf(fabsf(bflo(v0[e])), fabsf(bfhi(v0[e])))), fmaxf(fabsf(bflo(v1[e])), fabsf(bfhi(v1[e]))));
    u32x4 o;
    o[0] = cvt4_fp8(bflo(v0[0]) * scale, bfhi(v0[0]) * scale, bflo(v0[1]) * scale, bfhi(v0[1]) * scale);
    o[1] = cvt4_fp8(bflo(v0[2]) * scale, bfhi(v0[2]) * scale, bflo(v0[3]) * scale, bfhi(v0[3]) * scale);
    o[2] = cvt4_fp8(bflo(v1[0]) * scale, bfhi(v1[0]) * scale, bflo(v1[1]) * scale, bfhi(v1[1]) * scale);
    o[3] = cvt4_fp8(bflo(v1[2]) * scale, bfhi(v1[2]) * scale, bflo(v1[3]) * scale, bfhi(v1[3]) * scale);
    *reinterpret_cast<u32x4*>(y + r * c + ch) = o;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) partials[blockIdx.x] = fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
}
}  // namespace

extern "C" int dsl_quant_fp8(const void* x, void* y, long rows, int c, int ld_x, float scale, void* stream) {
  DSL_CHECK(x && y && rows > 0 && c > 0 && c % 16 == 0 && ld_x >= c && ld_x % 8 == 0, "dsl_quant_fp8: bad arguments (c=%d ld_x=%d)", c, ld_x);
  const long long total = (long long)rows * (c / 16);
  hipLaunchKernelGGL(quant_fp8_kernel, dim3(nblocks(total, 4096)), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)x, (uint8_t*)y,
                     (long long)rows, c / 16, ld_x, c, scale);
  DSL_LAUNCH_CHECK("quant_fp8_kernel");
  return 0;
}

extern "C" int dsl_quant_fp8_weights(const float* w, void* w8, float* comb, const float* bn_scale, int cout, int cout_pad, int k,
                                     float inv_act_scale, void* stream) {
  DSL_CHECK(w && w8 && comb && cout > 0 && cout_pad >= cout && k > 0 && k % 4 == 0, "dsl_quant_fp8_weights: bad arguments");
  hipLaunchKernelGGL(quant_fp8_weights_kernel, dim3(cout_pad), dim3(256), 0, (hipStream_t)stream, w, (uint8_t*)w8, comb, bn_scale, cout, k,
                     inv_act_scale);
  DSL_LAUNCH_CHECK("quant_fp8_weights_kernel");
  return 0;
}

extern "C" int dsl_absmax(const void* x, long rows, int c, int ld_x, float* partials, int n_partials, void* stream) {
  DSL_CHECK(x && partials && rows > 0 && c > 0 && c % 8 == 0 && ld_x >= c && ld_x % 8 == 0 && n_partials > 0 && n_partials <= 4096,
            "dsl_absmax: bad arguments");
  hipLaunchKernelGGL(absmax_kernel, dim3(n_partials), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)x, (long long)rows, c / 8, ld_x, partials);
  DSL_LAUNCH_CHECK("absmax_kernel");
  return 0;
}

extern "C" int dsl_quant_fp8_dyn(const void* x, void* y, long rows, int c, int ld_x, const float* partials, int n_partials, void* stream) {
  DSL_CHECK(x && y && partials && rows > 0 && c > 0 && c % 16 == 0 && ld_x >= c && ld_x % 8 == 0 && n_partials > 0, "dsl_quant_fp8_dyn: bad arguments");
  const long long total = (long long)rows * (c / 16);
  hipLaunchKernelGGL(quant_fp8_dyn_kernel, dim3(nblocks(total, 4096)), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)x, (uint8_t*)y,
                     (long long)rows, c / 16, ld_x, c, partials, n_partials);
  DSL_LAUNCH_CHECK("quant_fp8_dyn_kernel");
  return 0;
}

extern "C" int dsl_fp8_comb(const float* winv, float* comb, int n, const float* partials, int n_partials, void* stream) {
  DSL_CHECK(winv && comb && partials && n > 0 && n_partials > 0, "dsl_fp8_comb: bad arguments");
  hipLaunchKernelGGL(fp8_comb_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, winv, comb, n, partials, n_partials);
  DSL_LAUNCH_CHECK("fp8_comb_kernel");
  return 0;
}


extern "C" int dsl_fp8_prep(const dsl_fp8_prep_item* items_dev, int n_items, int cout_pad, int k, float margin, void* stream) {
  DSL_CHECK(items_dev && n_items > 0 && n_items <= 65535 && cout_pad > 0 && k > 0 && k % 4 == 0 && margin >= 1.f,
            "dsl_fp8_prep: bad arguments (n_items=%d cout_pad=%d k=%d margin=%g)", n_items, cout_pad, k, (double)margin);
  hipLaunchKernelGGL(fp8_prep_kernel, dim3(cout_pad, n_items), dim3(256), 0, (hipStream_t)stream, items_dev, k, margin);
  DSL_LAUNCH_CHECK("fp8_prep_kernel");
  return 0;
}

extern "C" int dsl_quant_fp8_delayed(const void* x, void* y, long rows, int c, int ld_x, const float* scale_dev, float* partials,
                                     int n_partials, void* stream) {
  DSL_CHECK(x && y && scale_dev && partials && rows > 0 && c > 0 && c % 16 == 0 && ld_x >= c && ld_x % 8 == 0 && n_partials > 0 &&
            n_partials <= 4096, "dsl_quant_fp8_delayed: bad arguments");
  hipLaunchKernelGGL(quant_fp8_delayed_kernel, dim3(n_partials), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)x, (uint8_t*)y,
                     (long long)rows, c / 16, ld_x, c, scale_dev, partials);
  DSL_LAUNCH_CHECK("quant_fp8_delayed_kernel");
  return 0;
}
