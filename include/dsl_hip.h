/* dsl_hip.h — C ABI of libdsl_hip.so: the MI355X (gfx950) kernels behind the FCOS R50-FPN
 * teacher–student training step of chenbinghui1/DSL.
 *
 * Nothing like this exists in the reference (it is pure Python over torch/cuDNN/mmcv ops); each
 * entry point below names the reference call it replaces (paths relative to /root/reference).
 *
 * Conventions:
 *   - every function returns 0 on success, <0 on error; dsl_last_error() gives a thread-local text
 *   - the caller owns every buffer; the library never allocates device memory (workspace and table
 *     sizes are queried: dsl_*_workspace_bytes, dsl_wgrad_pixtab_bytes).  What the library DOES own, per
 *     device and for the life of the process: the HIP streams it schedules its op lists on (12 candidates
 *     created at first use, dsl_streams_init), the named events of dsl_run_ops / dsl_stream_*_slot and the
 *     event pairs of dsl_prof_*; tests/test_abi_gpu.py holds hipMemGetInfo still over multi-scale steps
 *   - kernels are enqueued on the given hipStream_t (passed as void*), no implicit device sync
 *   - activations are NHWC bf16; tensors that span the 5 FPN levels are stored level-major:
 *     [level][image][y][x][channel] ("segments"), which is exactly the flattened location order
 *     of mmdet/models/dense_heads/fcos_head.py:239-258
 *   - weights: fp32 master in KRSC = [Cout][kh][kw][Cin]; bf16 packed copies in the same order
 *     ("fwd pack", rows padded to 64) and in CRSK = [Cin][kh][kw][CoutPad] ("dgrad pack")
 */
#ifndef DSL_HIP_H
#define DSL_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DSL_MAX_SEG 5
#define DSL_MAX_GROUP 8   /* convolutions per grouped weight-gradient launch */

int dsl_version(void);
const char* dsl_last_error(void);
/* Library options - the library reads no environment variable.  Names: "wgrad_slots" (default 128: workgroup budget of a
 * weight-gradient launch whose descriptor leaves `slots` 0), "stream_probe" (1; 0 = the library takes its streams as the runtime deals
 * them instead of probing for hardware queues of their own, dsl_streams_init), "debug_sync" (0; 1 = dsl_run_ops drains the device after every op and
 * names it on stderr), "skip_kinds" (0; timing-only ablation: bit mask of op kinds dsl_run_ops skips), "comm_queue" (1; which of the four
 * hardware queues the communication stream is placed on, see dsl_comm_stream_queue; set it before the streams exist).  Unknown name: -1. */
int dsl_set_option(const char* name, int value);
int dsl_get_option(const char* name, int* value);

/* ------------------------------------------------------------------------------------------
 * Implicit-GEMM convolution on MFMA (forward, data-gradient) — replaces F.conv2d / cuDNN in
 * mmdet/models/backbones/resnet.py:262-301,598-645, necks/fpn.py:150-202,
 * dense_heads/anchor_free_head.py:197-217, fcos_head.py:154-156 and their autograd backward.
 * One "compute grid" pixel (seg, n, y, x) gathers, per tap (r, s), a contiguous channel vector of
 * the source:   mode 0 (forward):   (sy, sx) = (y*stride + r - pad, x*stride + s - pad)
 *               mode 1 (transposed): (sy, sx) = ((y + pad - r)/stride, (x + pad - s)/stride) if divisible
 * and writes to destination pixel (y*os, x*os).
 * Epilogue: v = acc*scale[c] + bias[c]; [mask_first: v *= (mask>0)]; v += addend (same index, or
 * nearest-upsampled from (ah, aw) when DSL_CONV_ADD_UPSAMPLE); [mask_last: v *= (mask>0)];
 * [relu]; store bf16 or fp32.
 * ---------------------------------------------------------------------------------------- */
enum {
  DSL_CONV_RELU_OUT = 1,      /* ReLU in the epilogue */
  DSL_CONV_RELU_IN = 2,       /* ReLU applied to the gathered source (FPN P7 = conv(relu(P6))) */
  DSL_CONV_OUT_F32 = 4,       /* destination is fp32 (head logits) instead of bf16 */
  DSL_CONV_MASK_FIRST = 8,    /* v = acc*(mask>0) + addend */
  DSL_CONV_MASK_LAST = 16,    /* v = (acc + addend)*(mask>0) */
  DSL_CONV_ADD_UPSAMPLE = 32, /* addend is nearest-upsampled (fpn.py:163-172) */
  DSL_CONV_SMALL_C = 64,      /* source has 8 channels (stem, image packed NHWC8) */
  DSL_CONV_FP8 = 128,         /* src and wgt are OCP fp8 e4m3 (dsl_quant_fp8 / dsl_quant_fp8_weights; forward mode, cs % 128 == 0,
                               * cs / lds count bytes): MX-scaled fp8 MFMA, `scale` carries 1 / (weight scale x activation scale) */
  /* bits 8-11: tile configuration, bits 12-15: split-K factor (test hooks; 0 = the library's own choice) */
  DSL_CONV_EPI_STAGED = 1 << 16   /* test hook: the general staged fp32 epilogue even where an in-register one applies (same bits) */
};

typedef struct dsl_conv_desc {
  int32_t nseg, n;                                   /* level segments, images */
  int32_t gh[DSL_MAX_SEG], gw[DSL_MAX_SEG];          /* compute grid per segment */
  int32_t sh[DSL_MAX_SEG], sw[DSL_MAX_SEG];          /* source spatial size */
  int32_t dh[DSL_MAX_SEG], dw[DSL_MAX_SEG];          /* destination spatial size */
  int32_t ah[DSL_MAX_SEG], aw[DSL_MAX_SEG];          /* addend spatial size (ADD_UPSAMPLE) */
  int32_t cs;                                        /* source channels (GEMM-K per tap) */
  int32_t cd;                                        /* real destination channels */
  int32_t cd_pad;                                    /* weight rows (multiple of 64) */
  int32_t ldd, lda, ldm;                             /* row strides (elements) of dst/addend/mask */
  int32_t kh, kw, stride, pad, mode, os, flags;
  const void* src;                                   /* bf16 */
  const void* wgt;                                   /* bf16 [cd_pad][kh*kw*cs] */
  void* dst;                                         /* bf16 or fp32 */
  const float* scale;                                /* [cd] or NULL (=1) */
  const float* bias;                                 /* [cd] or NULL (=0) */
  const void* addend;                                /* bf16 or NULL */
  const void* mask;                                  /* bf16 or NULL */
  void* workspace;                                   /* optional fp32 scratch for split-K (small-M, large-K convs) */
  size_t workspace_bytes;                            /* NULL/0: never split */
  int32_t cs_real;                                   /* 0 = cs; else the source's real channel count when cs is padded (the
                                                      * 80- / 5-channel predictor gradients stored 128 / 64 wide): only the
                                                      * algorithmic FLOP / byte counts of dsl_prof_* use it */
  int32_t lds;                                       /* 0 = cs; else the source's pixel stride in elements (>= cs, multiple of
                                                      * 8): the source is a channel slice [0, cs) of wider rows */
  void* gn_ws;                                       /* NULL, or the `workspace` of the dsl_gn_desc that normalises this output
                                                      * (8 channels per group, conv_stats = 1): the epilogue leaves the per-tile
                                                      * statistics records there and dsl_groupnorm_relu_fwd skips its own
                                                      * statistics pass (ConvModule conv -> GN -> ReLU, anchor_free_head.py:104-133);
                                                      * legal only where dsl_conv2d_gn_fusable() says 1 */
  const void* gn_x;                                  /* NULL: gn_ws takes the forward records (sum, sum of squares).  Else this launch is
                                                      * the data gradient that produces dY of that GroupNorm + ReLU, gn_x its bf16 input
                                                      * of the forward pass (same rows as dst, ldd == cd): gn_ws takes the BACKWARD
                                                      * records and dsl_groupnorm_relu_bwd (conv_stats = 1) skips its reduction pass */
  const float* gn_gamma;                             /* with gn_x: the norm's weight, bias and the forward statistics (dsl_gn_desc.stats) */
  const float* gn_beta;
  const float* gn_stats;
} dsl_conv_desc;

/* 1 if a launch of `d` can write GroupNorm records (pipelined kernel, bf16 output on its own pixel grid, no split-K) */
int dsl_conv2d_gn_fusable(const dsl_conv_desc* d);

/* bytes of split-K scratch this conv would like (0 if it will not split); any smaller buffer is legal */
size_t dsl_conv2d_workspace_bytes(const dsl_conv_desc* d);
int dsl_conv2d(const dsl_conv_desc* d, void* stream);

/* Weight gradient: dW[co][r][s][ci] = scale[co] * sum_p dY[p][co] * X[p@(r,s)][ci]  (fp32, KRSC).
 * Split-K over pixels into a caller-owned workspace, then a reduce pass.  Optionally also
 * db[co] = sum_p dY[p][co].  Replaces the autograd backward of the same F.conv2d calls. */
typedef struct dsl_wgrad_desc {
  int32_t nseg, n;
  int32_t gh[DSL_MAX_SEG], gw[DSL_MAX_SEG];          /* dY spatial size (conv output grid) */
  int32_t sh[DSL_MAX_SEG], sw[DSL_MAX_SEG];          /* X spatial size (conv input) */
  int32_t cs;                                        /* X channels (Cin) */
  int32_t cy;                                        /* dY channels = row stride (multiple of 64) */
  int32_t cd;                                        /* real Cout (rows of dW written) */
  int32_t kh, kw, stride, pad;
  int32_t splits;                                    /* from dsl_wgrad_splits() */
  const void* dy;                                    /* bf16 */
  const void* x;                                     /* bf16 */
  const float* scale;                                /* [cd] or NULL */
  float* dw;                                         /* fp32 [cd][kh*kw*cs] */
  float* db;                                         /* fp32 [cd] or NULL */
  void* workspace;
  size_t workspace_bytes;
  int32_t ldx;                                       /* 0 = cs; else X's pixel stride in elements (X is a channel slice of wider rows) */
  int32_t shared;                                    /* group launches: 1 = the members are applications of ONE convolution (same dw):
                                                      * their gradients are summed into dw (RLA's recurrent / conv_out layers) */
  int32_t slots;                                     /* 0 = default (option "wgrad_slots", 128); else the workgroup budget the split factor
                                                      * is chosen for: launches that run beside the caller's stream leave it CUs, the
                                                      * ones at the very end of a pass can take the chip (group launches: descs[0]'s) */
  int32_t pad_;
  const void* pixtab;                                /* the geometry's pixel descriptor table: caller-owned device memory of
                                                      * >= dsl_wgrad_pixtab_bytes(d), written ONCE per geometry by dsl_wgrad_pixtab_fill
                                                      * (any descriptor of the same nseg / n / grid / source sizes / kh / kw / stride /
                                                      * pad may share it).  May be NULL when dsl_wgrad_pixtab_bytes(d) is 0 */
  size_t pixtab_bytes;
} dsl_wgrad_desc;

int dsl_wgrad_splits(const dsl_wgrad_desc* d);
size_t dsl_wgrad_workspace_bytes(const dsl_wgrad_desc* d);
/* The pipelined weight-gradient kernels read their gather addresses (source pixel of tap (0, 0), tap validity bits) from a table of
 * 8 bytes per output pixel.  It is the CALLER's memory like every other buffer: query the size (0 = this geometry's kernel needs
 * none), fill it with one small kernel on `stream`, hand it over in dsl_wgrad_desc.pixtab.  The library keeps no device allocation
 * of its own (until round 6 it cached these tables in hipMalloc'd memory for the life of the process). */
size_t dsl_wgrad_pixtab_bytes(const dsl_wgrad_desc* d);
int dsl_wgrad_pixtab_fill(const dsl_wgrad_desc* d, void* table, size_t bytes, void* stream);
int dsl_conv2d_wgrad(const dsl_wgrad_desc* d, void* stream);
/* The weight gradients of `count` (<= DSL_MAX_GROUP) convolutions that share one geometry - every descriptor
 * field except dy, x, scale, dw, db - as ONE launch: the workgroups that fill the chip come from `count` times
 * more output tiles, so `count` times fewer pixel splits and fp32 partial tiles are needed (ResNet blocks of
 * one stage, the FCOS tower layers).  Uses descs[0].workspace (>= dsl_wgrad_group_workspace_bytes); the
 * members' `splits` fields are ignored. */
size_t dsl_wgrad_group_workspace_bytes(const dsl_wgrad_desc* descs, int count);
int dsl_conv2d_wgrad_group(const dsl_wgrad_desc* descs, int count, void* stream);

/* Multi launch: the weight gradients of up to DSL_MAX_MULTI sub-launches (sub-launch s = counts[s] consecutive descriptors
 * of one geometry, as in dsl_conv2d_wgrad_group) that share one tile configuration (dsl_wgrad_multi_config: 1..4; 0 = has
 * no multi form) as ONE grid and ONE reduce grid: the backward pass of a whole FPN or ResNet stage instead of one launch
 * pair per layer.  Split factors are chosen for the launch as a whole (every workgroup gets about 1/wgrad_slots of its
 * K iterations), sub-launches that end up with one split write dW directly and have no partial tiles or reduce pass.
 * The launch plan is a table the caller owns: dsl_wgrad_multi_build fills `table_host` (dsl_wgrad_multi_table_bytes()
 * bytes of host memory), the caller copies those bytes to device memory once and passes both to
 * dsl_conv2d_wgrad_multi.  The table holds the descriptors' pointers and `workspace`
 * (>= dsl_wgrad_multi_workspace_bytes): rebuild it when they change.  Results are bit-identical to the same sub-launches
 * run through dsl_conv2d_wgrad_group with the same split factors; with different split factors they differ by fp32
 * summation order only. */
#define DSL_MAX_MULTI 16
int dsl_wgrad_multi_config(const dsl_wgrad_desc* d);
size_t dsl_wgrad_multi_table_bytes(void);
size_t dsl_wgrad_multi_workspace_bytes(const dsl_wgrad_desc* descs, const int* counts, int nsub);
int dsl_wgrad_multi_build(const dsl_wgrad_desc* descs, const int* counts, int nsub, void* workspace, size_t workspace_bytes,
                          void* table_host, size_t table_bytes);
int dsl_conv2d_wgrad_multi(const void* table_host, const void* table_dev, void* stream);
int dsl_wgrad_multi_info(const void* table_host, double* flops, double* bytes, int* blocks, int* red_blocks, int* nsub);
/* The launch planner behind dsl_wgrad_multi_build, callable without a device (tests, tools): sub-launch s has stages[s] K stages
 * (32 pixels each), tiles[s] output tiles over all its members and tile_elems[s] fp32 elements per split (may be NULL);
 * cfg = tile configuration 1..3, cap = workgroups of the persistent grid (a multiple of 8).  The planner picks the split
 * factors by simulating the grid: items go longest first to the least loaded workgroup of their XCD class, cost = the longest
 * workgroup's stages (+ a fixed cost per item) + the reduce pass the partial tiles need.  splits_out[nsub]; info[5] =
 * {grid, makespan in stages, work items, makespan of round 3's rule (one target length, stride walk), its work items}. */
int dsl_wgrad_plan_probe(const int* stages, const int* tiles, const long long* tile_elems, int nsub, int cfg, int cap,
                         int* splits_out, int* info);

/* fp8 forward path (BASELINE.json configs[4], first slice; the reference trains in fp32 and has no such path).
 * dsl_quant_fp8: y[r][c] = e4m3(clamp(x[r][c] * scale, +-448)) for a bf16 [rows][ld_x] tensor -> fp8 [rows][c] (c % 16 == 0).
 * dsl_quant_fp8_weights: per output channel co < cout: s = 448 / max|w[co][:]| (1 if the row is zero), w8[co][k] = e4m3(w * s),
 * comb[co] = inv_act_scale / s  - the `scale` vector of the fp8 convolution's epilogue (x BatchNorm scale if bn_scale != NULL);
 * rows cout .. cout_pad are zero-filled with comb 0. */
int dsl_quant_fp8(const void* x_bf16, void* y_fp8, long rows, int c, int ld_x, float scale, void* stream);
/* Dynamic per-tensor activation scale (what the engine uses): dsl_absmax writes n_partials block maxima of |x|; dsl_quant_fp8_dyn
 * quantises with scale = 448 / max(partials) (the tensor's own maximum: nothing saturates); dsl_fp8_comb makes the convolution's
 * epilogue scale comb[i] = winv[i] * max(partials) / 448 from the weights' inverse scales (dsl_quant_fp8_weights with
 * inv_act_scale = 1).  No atomics, nothing to clear between steps. */
int dsl_absmax(const void* x_bf16, long rows, int c, int ld_x, float* partials, int n_partials, void* stream);
int dsl_quant_fp8_dyn(const void* x_bf16, void* y_fp8, long rows, int c, int ld_x, const float* partials, int n_partials, void* stream);
int dsl_fp8_comb(const float* winv, float* comb, int n, const float* partials, int n_partials, void* stream);
int dsl_quant_fp8_weights(const float* w, void* w8, float* comb, const float* bn_scale, int cout, int cout_pad, int k,
                          float inv_act_scale, void* stream);
/* Delayed scaling (round 6; what the engine uses for the towers' fp8 forward): an activation is quantised by its PRODUCER's pass with
 * the scale of the previous step's maximum, and leaves this step's block maxima for the next one - no pass of its own.
 * dsl_fp8_prep, one launch per step for n_items convolutions that share cout_pad and k: per item and output channel co < cout the
 *   weight row is quantised as dsl_quant_fp8_weights does (s_w = 448 / max|w[co][:]|); a = margin * max(amax[0 .. n_amax)) is the
 *   input tensor's expected maximum; scale[0] = 448 / a (1 when a == 0: nothing recorded yet - run the forward pass once to record)
 *   is what the input's producer multiplies by, comb[co] = a / 448 / s_w[co] the convolution's epilogue scale; rows co >= cout are 0.
 * dsl_quant_fp8_delayed: y = e4m3(clamp(x * scale[0], +-448)) for a tensor nobody's epilogue can quantise (the FPN outputs),
 *   partials[b] = block b's max|x| (n_partials blocks, every one written). */
typedef struct dsl_fp8_prep_item {
  const float* w;       /* fp32 [cout][k] */
  void* w8;             /* e4m3 [cout_pad][k] */
  float* comb;          /* [cout_pad] */
  const float* amax;    /* block maxima of the convolution's input, left by the previous step */
  float* scale;         /* [1] */
  int32_t n_amax, cout;
} dsl_fp8_prep_item;
int dsl_fp8_prep(const dsl_fp8_prep_item* items_dev, int n_items, int cout_pad, int k, float margin, void* stream);
int dsl_quant_fp8_delayed(const void* x_bf16, void* y_fp8, long rows, int c, int ld_x, const float* scale_dev, float* partials,
                          int n_partials, void* stream);

/* A whole trained bottleneck's forward pass as ONE launch (csrc/bneck.hip; mmdet/models/backbones/resnet.py:262-301 Bottleneck.forward,
 * caffe style: stride on conv1; eval-mode BatchNorms folded to (scale, bias)):
 *   a1 = relu(bn1(conv1_1x1/stride(x))), a2 = relu(bn2(conv2_3x3(a1))), out = relu(bn3(conv3_1x1(a2)) + identity)
 * x [n][hin][win] rows of ldx elements (cin read), a1 / a2 [n][h][w][planes] (written: the weight gradients read them), identity and out
 * [n][h][w] rows of ldi / ldo elements (4 planes channels); weights bf16 in the forward layout of dsl_conv2d: w1 [planes][cin],
 * w2 [planes][3][3][planes], w3 [4 planes][planes].  planes = 128 or 256 (ResNet-50's layer2 / layer3), stride 1 or 2.  Bit-identical to
 * the three dsl_conv2d launches it replaces (same K order, same rounding points). */
typedef struct dsl_bneck_desc {
  const void* x; const void* w1; const void* w2; const void* w3; const void* idt;
  const float* s1; const float* b1; const float* s2; const float* b2; const float* s3; const float* b3;
  void* a1; void* a2; void* out;
  int32_t n, hin, win, h, w;
  int32_t planes, cin, ldx, stride, ldi, ldo;
} dsl_bneck_desc;
int dsl_bottleneck_fwd_supported(const dsl_bneck_desc* d);      /* 1 if dsl_bottleneck_fwd takes this shape */
int dsl_bottleneck_fwd(const dsl_bneck_desc* d, void* stream);

/* ------------------------------------------------------------------------------------------
 * GPU data path
 * ---------------------------------------------------------------------------------------- */
/* One image of a batch: decoded uint8 HWC (BGR, as mmcv.imread yields) on the device and the parameters the reference's CPU
 * pipeline draws per sample - Resize(keep_ratio) target size, PatchShuffle mode / split, RandomFlip, Normalize
 * (mmdet/datasets/pipelines/transforms.py:218-247, 2143-2248, 334-470, 652-690). */
typedef struct dsl_image_prep_item {
  const unsigned char* src;      /* [src_h][src_w][3] */
  int32_t src_h, src_w;
  int32_t new_h, new_w;          /* size after Resize (= src size: no resampling) */
  int32_t flip;                  /* horizontal */
  int32_t ps_mode, ps_crop;      /* PatchShuffle: 0 none, 1 'flip' (columns [crop, w) first), 2 'flop' (rows [crop, h) first) */
  int32_t to_rgb;
  float mean[3], inv_std[3];     /* in output channel order */
} dsl_image_prep_item;
/* dst[n][3][hc][wc] fp32 <- resize (OpenCV's 8-bit fixed-point bilinear) -> PatchShuffle -> flip -> normalise, zero padded to the
 * canvas (Pad(size_divisor) and the loader's merge/pad, datasets/builder.py:236-267).  items_dev: device array of n items. */
int dsl_image_prep(const dsl_image_prep_item* items_dev, int n, float* dst, int hc, int wc, void* stream);
/* The same up to (not including) Normalize: the uint8 BGR images [n][hc][wc][3] (zero outside an image's new_h x new_w) that the
 * unlabeled stream's augmentations work on. */
int dsl_image_prep_u8(const dsl_image_prep_item* items_dev, int n, unsigned char* dst_u8, int hc, int wc, void* stream);

/* One augmentation pass over a batch of uint8 canvases (src -> dst, different buffers): per image `kind` selects the pass.
 * RandomAugmentBBox_Fast(aug_type='affine') (mmdet/datasets/pipelines/semi_aug.py:344-531: imgaug Affine, whole image or inside
 * one box) and UBAug (transforms.py:2098-2140: torchvision ColorJitter in its random order, RandomGrayscale, GaussianBlur,
 * three RandomErasing).  The host draws the parameters and transforms the boxes.  The colour, grayscale and blur passes are
 * Pillow's 8-bit arithmetic (ImageEnhance / Blend.c, Convert.c, BoxBlur.c) bit for bit - pinned against Pillow's own outputs
 * (tests/golden/ubaug_pil.npz); the stored channel order plays Pillow's (R, G, B), as the reference hands mmcv's BGR array to
 * ToPILImage unchanged.  imgaug is not available to this build: AFFINE follows its documented convention, parity UNPINNED.
 * ERASE writes its own N(0, 1) stream through torchvision's value mapping (mul(255).byte(): truncate, wrap). */
enum { DSL_AUG_COPY = 0, DSL_AUG_AFFINE = 1, DSL_AUG_BRIGHTNESS = 2, DSL_AUG_CONTRAST = 3, DSL_AUG_SATURATION = 4, DSL_AUG_HUE = 5,
       DSL_AUG_GRAY = 6, DSL_AUG_BLUR_H = 7, DSL_AUG_BLUR_V = 8, DSL_AUG_ERASE = 9,
       /* RandAug's ops on an image without boxes (mmdet/datasets/pipelines/autoaug_fast.py:219-224, 244-250, 371-372, 407, reached from
        * semi_aug.py:494-497): Pillow's ImageOps.autocontrast / equalize / solarize / posterize and ImageEnhance.Sharpness bit for
        * bit, pinned against Pillow's own outputs (tests/golden/randaug_pil.npz) */
       DSL_AUG_AUTOCONTRAST = 10, DSL_AUG_EQUALIZE = 11, DSL_AUG_SOLARIZE = 12, DSL_AUG_POSTERIZE = 13, DSL_AUG_SHARPNESS = 14 };
typedef struct dsl_aug_item {
  int32_t h, w;                  /* the image inside its canvas */
  int32_t kind;
  int32_t order;                 /* AFFINE: 0 nearest, 1 bilinear */
  float m[6];                    /* AFFINE: output pixel (x, y) relative to roi's corner -> source position x_s = m0 x + m1 y + m2,
                                  * y_s = m3 x + m4 y + m5 inside the roi (the inverse of the drawn transform) */
  int32_t roi[4];                /* AFFINE: x0, y0, x1, y1 - the region replaced and sampled (whole image, or one box) */
  int32_t cval;                  /* AFFINE: fill value (125) */
  float f[3];                    /* f[0]: BRIGHTNESS / CONTRAST / SATURATION / SHARPNESS the enhancement factor; HUE int(factor * 255), the
                                  * 8-bit shift; BLUR_H / BLUR_V the fractional box radius of GaussianBlur(sigma) (BoxBlur.c); SOLARIZE the
                                  * threshold (0 .. 256); POSTERIZE the bits kept (1 .. 8) */
  int32_t rect[3][4];            /* ERASE: up to three rectangles x0, y0, x1, y1 (x1 <= x0: unused) */
  uint32_t seed;                 /* ERASE: noise stream */
} dsl_aug_item;
/* scratch: dsl_image_aug_scratch_bytes(n) bytes, 8-byte aligned (per image: the luma sum, the three 256-bin band histograms and
 * the tables derived from them); need_stats: bit 0 set when some item is a CONTRAST pass, bit 1 when some item is an AUTOCONTRAST or
 * EQUALIZE pass (0: scratch may be NULL). */
size_t dsl_image_aug_scratch_bytes(int n);
int dsl_image_aug(const dsl_aug_item* items_dev, int n, const unsigned char* src, unsigned char* dst, int hc, int wc,
                  void* scratch, int need_stats, void* stream);
/* Normalize + Pad of a uint8 canvas batch (the tail of dsl_image_prep): dst[n][3][hc][wc] fp32. */
int dsl_image_normalize(const unsigned char* src_u8, const dsl_image_prep_item* items_dev, int n, float* dst, int hc, int wc,
                        void* stream);

/* ------------------------------------------------------------------------------------------
 * Memory-bound fused layers
 * ---------------------------------------------------------------------------------------- */
/* NCHW fp32 image -> NHWC8 bf16 (channels 3..7 zero).  Replaces the implicit layout of
 * SingleStageDetector.extract_feat's input (detectors/single_stage.py:40-45). */
int dsl_pack_image(const float* img_nchw, void* out_nhwc8, int n, int h, int w, void* stream);

/* The frozen stem as one kernel: fp32 NCHW image -> conv1 7x7 / 2 (3 -> 64) + eval-mode BatchNorm (scale, bias) + ReLU + MaxPool
 * 3x3 / 2 -> NHWC bf16 [n][ph][pw] rows of ld_out elements (resnet.py:598-645 _make_stem_layer, :634-638).  w_groups: conv1's
 * weight as [22][64][8] bf16 - group g holds, for every cout, the values 8 (g % 3) .. + 8 of the 21 (kx, c) values of tap row
 * g / 3, zero beyond (ParamStore.stem_groups16).  Same arithmetic as dsl_pack_image + dsl_conv2d + dsl_maxpool3x3s2 up to the
 * fp32 summation order of the 147 products. */
int dsl_stem_pool(const float* img_nchw, const void* w_groups, const float* scale, const float* bias, void* out, int ld_out,
                  int n, int h, int w, void* stream);
/* The same with half_last = 1: img_nchw holds n - 1 images; image n - 1 is SemiEpochBasedRunner's scale-invariant copy of the
 * last one (semi_epoch_based_runner.py:186-204: F.interpolate(img[-1:], size = (h / 2, w / 2), mode = 'bilinear') pasted at the
 * top-left of a zero canvas), sampled from its source while the patch is loaded - no interpolate / zeros / cat passes, no
 * second copy of the batch; bit-identical to them (h, w even). */
int dsl_stem_pool_half(const float* img_nchw, const void* w_groups, const float* scale, const float* bias, void* out, int ld_out,
                       int n, int h, int w, int half_last, void* stream);

/* 3x3 stride-2 pad-1 max pool, NHWC bf16 (resnet.py:610,638). */
int dsl_maxpool3x3s2(const void* x, void* y, int n, int h, int w, int c, void* stream);
int dsl_maxpool3x3s2_ld(const void* x, void* y, int n, int h, int w, int c, int ldy, void* stream);   /* output row stride ldy >= c */

/* GroupNorm(32 groups, eps) + ReLU over level-major NHWC bf16 (mmcv ConvModule norm+act as used at
 * anchor_free_head.py:104-133).  stats = [nseg*n*groups][2] fp32 (mean, rstd), written by fwd.
 * Both directions are two passes with two-level, fixed-order reductions through `workspace` (block records,
 * no atomics, nothing to pre-zero): results are bit-identical from run to run. */
typedef struct dsl_gn_desc {
  int32_t nseg, n, c, groups;
  int32_t h[DSL_MAX_SEG], w[DSL_MAX_SEG];
  float eps;
  const void* x;          /* bf16 pre-norm conv output */
  void* y;                /* bf16 relu(gn(x)) */
  const float* gamma;
  const float* beta;
  float* stats;
  /* backward only */
  const void* dy;         /* bf16 grad wrt y */
  void* dx;               /* bf16 grad wrt x */
  float* dgamma;          /* fp32 [c], overwritten */
  float* dbeta;           /* fp32 [c], overwritten */
  float* dbias;           /* fp32 [c] or NULL: sum over pixels of dx = gradient of the bias of the conv that made x */
  void* workspace;        /* >= dsl_groupnorm_workspace_bytes(d); private to this call until it completes */
  size_t workspace_bytes;
  int32_t conv_stats;     /* 1 = a convolution already left this call's block records in `workspace` (dsl_conv_desc.gn_ws ==
                           * workspace): forward, the one that produced x; backward, the data gradient that produced dy (its
                           * gn_x = x).  One pass instead of two; needs c / groups == 8 */
  int32_t pad_;
  /* forward only, optional (all NULL: off) - the fp8 copy of y that an fp8 convolution reads next (DSL_CONV_FP8), written in the
   * same pass with a DELAYED scale: y8 = e4m3(clamp(bf16(y) * y8_scale[0], +-448)) [pixel][c]; y8_amax[(segment * n + image) * nblk +
   * block] = max of this call's bf16(y) over the block's pixels (nblk = ceil(max hw / 128); blocks outside a level never write:
   * clear the buffer once) - the maxima dsl_fp8_prep turns into the NEXT step's scale */
  void* y8;
  const float* y8_scale;
  float* y8_amax;
} dsl_gn_desc;
size_t dsl_groupnorm_workspace_bytes(const dsl_gn_desc* d);
int dsl_groupnorm_relu_fwd(const dsl_gn_desc* d, void* stream);
int dsl_groupnorm_relu_bwd(const dsl_gn_desc* d, void* stream);

/* ------------------------------------------------------------------------------------------
 * RLA_ResNet layers (mmdet/models/backbones/resnet_rla.py:71-137,289-327): explicit row strides (ld, elements) because the
 * block input cat(x, h) is ONE buffer [pixel][C + 128] = [x | h (32) | zeros]
 * ---------------------------------------------------------------------------------------- */
/* nn.AvgPool2d((2, 2), stride=(2, 2)) on h at a stage transition (:98-100,129-130) and its backward */
int dsl_avgpool2x2(const void* x, int ldx, void* y, int ldy, int n, int h, int w, int c, void* stream);
int dsl_avgpool2x2_bwd(const void* gy, int ldgy, void* gx, int ldgx, int n, int h, int w, int c, void* stream);
/* t = tanh(bn(u)), BN in eval mode folded to (scale, bias) (:318-320), and its backward: g_u, dgamma, dbeta (overwritten;
 * block records in `workspace`, fixed-order sums).  dgamma = dbeta = NULL: the records [ceil(rows / 256)][2 c] are left
 * in `workspace` and dsl_rec_sum_multi sums them - several row ranges of one tensor (image-split backward chains on
 * two streams) write their records side by side and are summed by one launch behind both. */
int dsl_bn_tanh_fwd(const void* u, int ldu, const float* scale, const float* bias, void* t, int ldt, long rows, int c,
                    void* stream);
size_t dsl_bn_tanh_bwd_workspace_bytes(long rows, int c);
int dsl_bn_tanh_bwd(const void* gt, int ldgt, const void* t, int ldt, const void* u, int ldu, const float* scale,
                    const float* mean, const float* var, float eps, void* gu, int ldgu, float* dgamma, float* dbeta,
                    void* workspace, long rows, int c, void* stream);
typedef struct dsl_rec_sum_item {
  const float* rec; float* out_a; float* out_b;      /* out_a[ch] = sum_r rec[r][ch], out_b[ch] = sum_r rec[r][c + ch] */
  int32_t nrec, pad_;
} dsl_rec_sum_item;
int dsl_rec_sum_multi(const dsl_rec_sum_item* items_dev, int n, int c, void* stream);      /* items: DEVICE array, one workgroup each */
/* eval-mode BatchNorm with TRAINABLE affine parameters (norm_eval freezes the statistics only, :380-388): per-step fold
 * scale = gamma / sqrt(var + eps), bias = beta - mean * scale over n channels */
int dsl_bn_fold(const float* gamma, const float* beta, const float* mean, const float* var, float eps, float* scale,
                float* bias, int n, void* stream);
/* (gamma, beta) gradients of conv -> BN(eval) pairs from the UNSCALED weight gradient dWu (dsl_conv2d_wgrad with scale =
 * NULL, db = dbeta): dgamma[c] = (<W[c], dWu[c]> - mean[c] dbeta[c]) / sqrt(var[c] + eps), then dWu[c] *= gamma[c] /
 * sqrt(var[c] + eps) in place.  `items` is a DEVICE array; one workgroup per weight row (row_start = prefix sum of rows). */
typedef struct dsl_bn_post_item {
  const float* w; float* dw; float* dgamma; const float* dbeta; const float* gamma; const float* mean; const float* var;
  int32_t rows, k, row_start, pad_;
} dsl_bn_post_item;
int dsl_bn_wgrad_post(const dsl_bn_post_item* items_dev, int n, int total_rows, float eps, void* stream);
/* The recurrent path of one RLA block as ONE launch (resnet_rla.py:125-136): u = h_in + conv_out(x) (1x1, c4 -> 32; w_conv_out rows of c4
 * elements, 32 used), t = tanh(u * bn_scale + bn_bias), h_out = recurrent_conv(t) (3x3, pad 1; w_recurrent [>= 32][3][3][tw], the first
 * 32 of every tw input columns used).  x [n*h*w][ldx], h_in [..][ldh], u [..][32], t [..][tw] (first 32 written), h_out [..][ldo] (32 written);
 * bf16.  Same rounding points as dsl_conv2d (addend h_in) -> dsl_bn_tanh_fwd -> dsl_conv2d, which it replaces. */
int dsl_rla_tail_fwd(const void* x, int ldx, const void* h_in, int ldh, const void* w_conv_out, int c4, const float* bn_scale,
                     const float* bn_bias, const void* w_recurrent, int tw, void* u, void* t, void* h_out, int ldo, int n, int h, int w,
                     void* stream);
/* ... and its backward tail as ONE launch: g_t = recurrent_conv^T(g_h) (the 3x3 data gradient; wT_recurrent = the data-gradient pack
 * [>= 32 ci][3][3][ldw co], 32 co used; g_h [n*h*w][ldgh], 32 used) followed by dsl_bn_tanh_bwd's arithmetic on it (t [..][ldt], u [..][32],
 * g_u [..][ldgu], 32 written; dgamma / dbeta fp32 [32] overwritten, or NULL: the tile records [tiles][64] stay in `workspace` for
 * dsl_rec_sum_multi).  workspace >= dsl_rla_tail_bwd_workspace_bytes(n, h, w).  g_t is never stored.  Replaces dsl_conv2d (mode 1) ->
 * dsl_bn_tanh_bwd; g_u equals theirs bit for bit, dgamma / dbeta differ by the summation order of their records only. */
size_t dsl_rla_tail_bwd_workspace_bytes(int n, int h, int w);
int dsl_rla_tail_bwd(const void* g_h, int ldgh, const void* wT_recurrent, int ldw, const void* t, int ldt, const void* u,
                     const float* scale, const float* mean, const float* var, float eps, void* g_u, int ldgu, float* dgamma, float* dbeta,
                     void* workspace, int n, int h, int w, void* stream);
/* one of the above as an op-list entry (DSL_OP_RLA): kind selects the call, the arguments are taken in declaration order
 * from p[] (pointers), i[] (ints: strides / sizes), f[] (eps), rows */
enum { DSL_RLA_AVGPOOL = 2, DSL_RLA_AVGPOOL_BWD = 3, DSL_RLA_BN_TANH = 4, DSL_RLA_BN_TANH_BWD = 5,
       DSL_RLA_BN_FOLD = 6, DSL_RLA_BN_POST = 7, DSL_RLA_REC_SUM = 8 /* p[0] = items, i[0] = n, i[1] = c */,
       DSL_RLA_TAIL_FWD = 9 /* p[] = x, h_in, w_conv_out, bn_scale, bn_bias, w_recurrent, u, t, h_out; i[] = ldx, ldh, c4, tw, ldo, n, h, w */,
       DSL_RLA_TAIL_BWD = 10 /* p[] = g_h, wT_recurrent, t, u, scale, mean, var, g_u, dgamma, dbeta, workspace; i[] = ldgh, ldw, ldt, ldgu, n, h, w;
                              * f[0] = eps */ };
typedef struct dsl_rla_desc {
  int32_t kind;
  int32_t i[8];
  float f[2];
  int64_t rows;
  void* p[12];
} dsl_rla_desc;
int dsl_rla_op(const dsl_rla_desc* d, void* stream);

/* out[n][y][x][c] = sum over the children of (y,x) in g (backward of the FPN nearest upsample,
 * fpn.py:163-172).  out is (h, w), g is (ch, cw) (normally 2h x 2w); bf16. */
int dsl_sum2x2(const void* g, void* out, int n, int h, int w, int ch, int cw, int c, void* stream);

/* per-channel column sum of a bf16 [rows][ld] matrix -> fp32 [c]  (conv bias gradient) */
int dsl_colsum(const void* x, float* out, long rows, int c, int ld, void* stream);

/* ------------------------------------------------------------------------------------------
 * FCOS targets and losses (fcos_head.py:170-338,550-726; losses/focal_loss.py:11-56;
 * losses/iou_loss.py:85-102; losses/cross_entropy_loss.py:73-112)
 * ---------------------------------------------------------------------------------------- */
typedef struct dsl_fcos_desc {
  int32_t nlvl, n;                     /* levels, images */
  int32_t h[DSL_MAX_SEG], w[DSL_MAX_SEG], stride[DSL_MAX_SEG];
  float range_lo[DSL_MAX_SEG], range_hi[DSL_MAX_SEG];
  float radius;                        /* center_sample_radius (1.5) */
  int32_t num_classes;                 /* 80 */
  /* ground truth, concatenated over images; gt_off[n+1] prefix offsets (device int32) */
  const float* gt_boxes;               /* [G][4] xyxy */
  const int64_t* gt_labels;            /* [G] */
  const int32_t* gt_off;               /* [n+1] */
  const float* ig_boxes;               /* ignore boxes or NULL */
  const int32_t* ig_off;               /* [n+1] or NULL */
  /* outputs of assign, level-major [lvl][img][y][x] */
  int64_t* labels;                     /* [M] in [0, num_classes] */
  float* bbox_targets;                 /* [M][4], already / stride (norm_on_bbox) */
  int32_t* assign_idx;                 /* [M] argmin gt index within the image, -1 = background */
  float* cls_weight;                   /* [M] ignore weight * stream weight */
  float* pos_weight;                   /* [M] stream weight (bbox / centerness) */
  float* stats;                        /* [8]: 0 num_pos, 1 sum ctr targets (this rank) ... */
  float loss_weight;                   /* unlabeled-stream weight (DSL) ; 1.0 = off */
  /* head outputs */
  const float* cls_logits;             /* [M][ld_cls] fp32 */
  const float* regctr;                 /* [M][ld_rc] fp32: raw conv_reg (4) + centerness logit (1) */
  int32_t ld_cls, ld_rc;
  const float* scales;                 /* [nlvl] Scale parameters */
  /* [0] = sum over ranks of num_pos, [1] = sum over ranks of sum(ctr targets); the kernel applies
   * max(x*inv_world, 1) and max(x*inv_world, 1e-6)  (reduce_mean, fcos_head.py:264-274) */
  const float* norm;
  /* gradients (bf16, padded rows) and loss sums */
  void* g_cls;  int32_t ld_gcls;       /* bf16 [M][ld_gcls] */
  void* g_rc;   int32_t ld_grc;        /* bf16 [M][ld_grc] */
  float* g_scales;                     /* fp32 [DSL_MAX_SEG], overwritten */
  float* losses;                       /* fp32 [4]: cls, bbox, centerness, sisoft (overwritten) */
  float soft_weight;                   /* effective sisoft weight (0 = off) */
  float grad_scale;                    /* d(total)/d(each loss), normally 1 */
  float inv_world;                     /* 1/world_size: norm[] holds the SUM over ranks of stats[0:2] */
  void* workspace;                     /* >= dsl_fcos_workspace_bytes(d): block records of the loss / num_pos sums, which */
  size_t workspace_bytes;              /* are added up in a fixed order (bit-identical results from run to run) */
  float* logvec;                       /* NULL, or fp32 [5]: cls, bbox, centerness, [sisoft if soft_weight != 0,] their sum - the log
                                        * vector of BaseDetector._parse_losses (detectors/base.py:175-208) without a framework op */
} dsl_fcos_desc;
size_t dsl_fcos_workspace_bytes(const dsl_fcos_desc* d);

int dsl_fcos_points(const dsl_fcos_desc* d, float* points /* [P][2] */, void* stream);
int dsl_fcos_assign(const dsl_fcos_desc* d, void* stream);
int dsl_fcos_loss(const dsl_fcos_desc* d, void* stream);

/* ------------------------------------------------------------------------------------------
 * Optimizer / EMA over flat fp32 buffers (mmcv OptimizerHook + torch SGD, wired at
 * apis/train.py:111,157-166; EMA at runner/hooks/semi_epoch_based_runner.py:368-409)
 * ---------------------------------------------------------------------------------------- */
int dsl_sumsq(const float* x, long n, float* out /* [1], accumulated */, void* stream);
/* the same sum, overwritten (not accumulated), summed in a fixed order: workspace >= 1024 floats */
int dsl_sumsq_det(const float* x, long n, float* out /* [1] */, float* workspace, void* stream);
/* p -= lr*lr_mult[i] * (m = mom*m + (g*clip + wd*wd_mult[i]*p)); also writes bf16(p) to p16.
 * clip = min(max_norm/(sqrt(*gnorm_sq)+1e-6), 1) when gnorm_sq != NULL.  group[i] in {0,1}:
 * 1 = bias group (lr*bias_lr_mult, wd*bias_decay_mult). */
int dsl_sgd_step(float* p, const float* g, float* m, void* p16, const uint8_t* group, long n,
                 float lr, float momentum, float wd, float bias_lr_mult, float bias_decay_mult,
                 const float* gnorm_sq, float max_norm, int first_step, void* stream);
int dsl_ema_lerp(float* teacher, const float* student, long n, float keep, void* stream);
/* the same, and the teacher's bf16 forward copy written in the same pass (what dsl_cast_bf16 would re-read the result for) */
int dsl_ema_lerp_bf16(float* teacher, const float* student, void* teacher_bf16, long n, float keep, void* stream);
int dsl_cast_bf16(const float* x, void* y, long n, void* stream);
/* bf16 -> fp32 (n % 4 == 0): the way back of a gradient bucket that crossed xGMI as bf16 (dsl_allreduce_bucket_bf16). */
int dsl_cast_f32(const void* x_bf16, float* y, long n, void* stream);
/* The clipping norm in pieces, for data parallel with grad_clip (every DSL config: mmcv OptimizerHook's clip_grad_norm_,
 * mmdet/apis/train.py:157-166): one dsl_sumsq_partial per gradient bucket as its all-reduce completes (DSL_SUMSQ_PARTS block
 * sums of x[0..n) -> partials), one dsl_sumsq_fold over all of them (index order) behind the last bucket.  No atomics. */
#define DSL_SUMSQ_PARTS 256
int dsl_sumsq_partial(const float* x, long n, float* partials /* [DSL_SUMSQ_PARTS] */, void* stream);
int dsl_sumsq_fold(const float* partials, int n_partials, float* out /* [1] */, void* stream);
/* KRSC fp32 -> CRSK bf16 ("dgrad pack"), optional per-cout scale fold. */
int dsl_pack_dgrad(const float* w, const float* scale, void* out, int cout, int cout_pad, int taps,
                   int cin, void* stream);
/* the same for many convs in ONE launch: `items` is a DEVICE array of n entries */
typedef struct dsl_pack_item {
  const float* w; const float* scale; void* out;
  int32_t cout, cout_pad, taps, cin;
  int32_t block_start, tiles_ci, tiles_co;             /* 64x64 (cout x cin) tiles; block_start: prefix sum of tiles_ci*tiles_co*taps */
  int32_t tapmap;   /* 0: out tap t = source tap t of `taps`.  Else a tap SELECTION (the parity-class packs of a stride-2 3x3 data
                     * gradient): bits 16-23 = taps of the source weight (9), nibble t (bits 4t..4t+3) = source tap of out tap t,
                     * 0xF = a zero tap; `taps` (<= 4) counts the OUT taps */
} dsl_pack_item;
int dsl_pack_dgrad_batched(const dsl_pack_item* items_dev, int n, int total_blocks, void* stream);

/* ------------------------------------------------------------------------------------------
 * Teacher sweep post-processing (fcos_head.py:406-548, core/post_processing/bbox_nms.py:7-94)
 * ---------------------------------------------------------------------------------------- */
typedef struct dsl_det_desc {
  int32_t nlvl, n;
  int32_t h[DSL_MAX_SEG], w[DSL_MAX_SEG], stride[DSL_MAX_SEG];
  int32_t num_classes, nms_pre, max_per_img;
  float score_thr, iou_thr;
  const float* cls_logits; int32_t ld_cls;
  const float* regctr; int32_t ld_rc;        /* raw conv_reg; scale, relu, *stride applied here */
  const float* scales;
  const float* img_shapes;                   /* [n][2] (h, w) clip bounds */
  const float* scale_factors;                /* [n][4] */
  float* dets;                               /* [n][max_per_img][5] */
  int64_t* det_labels;                       /* [n][max_per_img] */
  int32_t* det_count;                        /* [n] */
  void* workspace; size_t workspace_bytes;
} dsl_det_desc;
size_t dsl_detect_workspace_bytes(const dsl_det_desc* d);
int dsl_fcos_detect(const dsl_det_desc* d, void* stream);

/* The label-file step of the pseudo-label refresh (runner/hooks/unlabel_pred_hook.py:20-57,84-171 with
 * fuse_history=False) on the detections of dsl_fcos_detect, per image: keep score >= parse_thr, int()-truncate the
 * coordinates, round the score to 6 decimals, then per class 0..num_classes-1 mmcv.ops.nms(iou_thr, score_threshold =
 * nms_thr) on the truncated boxes.  Outputs are ordered class-ascending, score-descending:
 * out_boxes [n][max_per_img][4], out_scores [n][max_per_img], out_labels [n][max_per_img], out_count [n]. */
int dsl_pseudo_label_fuse(const float* dets, const int64_t* labels, const int32_t* count, int n, int max_per_img,
                          int num_classes, float parse_thr, float iou_thr, float nms_thr, float* out_boxes,
                          float* out_scores, int64_t* out_labels, int32_t* out_count, void* stream);

/* The same step with fuse_history=True (unlabel_pred_hook.py:131-141): the image's previous labels - old_boxes
 * [n][max_old][4], old_scores [n][max_old], old_labels [n][max_old] (class indices), old_count [n], taken as the label
 * file held them (no truncation, no parse_thr) - precede the new detections in each class's candidate list.
 * max_per_img + max_old <= 1024; the outputs have max_out >= max_per_img + max_old slots per image.  max_old = 0 (old
 * pointers may be NULL) is dsl_pseudo_label_fuse. */
int dsl_pseudo_label_fuse_history(const float* dets, const int64_t* labels, const int32_t* count, int n, int max_per_img,
                                  const float* old_boxes, const float* old_scores, const int64_t* old_labels,
                                  const int32_t* old_count, int max_old, int num_classes, float parse_thr, float iou_thr,
                                  float nms_thr, float* out_boxes, float* out_scores, int64_t* out_labels,
                                  int32_t* out_count, int max_out, void* stream);

/* ------------------------------------------------------------------------------------------
 * Data-parallel exchange (one process per GPU, RCCL over xGMI).  Stands where the reference has
 * MMDistributedDataParallel's gradient buckets (mmdet/apis/train.py:92-96) and reduce_mean
 * (mmdet/core/utils/dist_utils.py:63-69).  librccl.so.1 is bound at the first call (dlopen), not at
 * load time.  `comm` is an rcclComm_t (= ncclComm_t) passed as void*: one made here, or the caller's own.
 *   dsl_comm_unique_id   rank 0 fills 128 bytes (ncclGetUniqueId) and hands them to every rank by its own means
 *   dsl_comm_init_rank   ncclCommInitRank on the calling thread's current device (collective over the ranks)
 *   dsl_comm_size        number of ranks of the communicator
 *   dsl_allreduce_bucket in-place fp32 sum of buf[0..count) over the ranks, queued on `stream`; the 1 / world
 *                        averaging of the gradients is folded into the loss kernel's gradient scale
 *   dsl_allreduce_buckets  n ranges in one RCCL group call (one launch for several small buckets)
 * ---------------------------------------------------------------------------------------- */
int dsl_comm_unique_id(void* id128);
int dsl_comm_init_rank(void** comm, int nranks, const void* id128, int rank);
int dsl_comm_size(void* comm);
int dsl_comm_destroy(void* comm);
int dsl_allreduce_bucket(void* comm, float* buf, size_t count, void* stream);
int dsl_allreduce_buckets(void* comm, float* const* bufs, const size_t* counts, int n, void* stream);
/* The same exchange on a bf16 copy of the bucket (half the bytes over xGMI; north_star allows bf16 gradients on the wire): in-place
 * bf16 sum of buf[0..count).  The caller casts fp32 -> bf16 before (dsl_cast_bf16) and back after (dsl_cast_f32); the master gradient,
 * the clipping norm and the optimizer stay fp32. */
int dsl_allreduce_bucket_bf16(void* comm, void* buf_bf16, size_t count, void* stream);

/* ------------------------------------------------------------------------------------------
 * Op-list executor: run a prebuilt sequence of the ops above with one call (keeps the per-step
 * host cost of ~400 launches out of Python).
 * ---------------------------------------------------------------------------------------- */
enum { DSL_OP_CONV = 1, DSL_OP_WGRAD = 2, DSL_OP_GN_FWD = 3, DSL_OP_GN_BWD = 4, DSL_OP_MAXPOOL = 5,
       DSL_OP_SUM2X2 = 6, DSL_OP_COLSUM = 7, DSL_OP_MEMSET = 8, DSL_OP_PACK_IMAGE = 9,
       DSL_OP_ASSIGN = 10, DSL_OP_LOSS = 11,   /* MAXPOOL: i[4] = output row stride (0 = c) */
       DSL_OP_FORK = 12,   /* side stream i[0] (default 1) waits for everything queued so far on stream i[1] (default 0 = caller's) */
       DSL_OP_JOIN = 13,   /* stream i[1] (default 0 = caller's) waits for everything queued so far on side stream i[0] (default 1) */
       DSL_OP_WGRAD_GROUP = 14,  /* desc = dsl_wgrad_desc[i[0]] -> dsl_conv2d_wgrad_group */
       DSL_OP_RLA = 17,    /* desc = dsl_rla_desc -> dsl_rla_op */
       DSL_OP_RECORD = 15, /* mark "everything queued so far on stream i[0]" in named event slot i[1] (0..15); survives the call */
       DSL_OP_WAIT = 16,   /* stream i[0] waits for named event slot i[1] (no-op if the slot was never recorded) */
       DSL_OP_PACK_DGRAD = 18, /* dsl_pack_dgrad_batched(p[0] = item table, i[0] = items, i[1] = blocks) */
       DSL_OP_WGRAD_MULTI = 19, /* dsl_conv2d_wgrad_multi(p[0] = table_host, p[1] = table_dev) */
       DSL_OP_QUANT_FP8 = 22,  /* p[2] == NULL: dsl_quant_fp8(p[0] = x, p[1] = y, l[0] = rows, i[0] = c, i[1] = ld_x, scale = the float whose bits are
                                * l[1]); else dsl_absmax(.., p[2] = partials, i[2] = n_partials) + dsl_quant_fp8_dyn(..) */
       DSL_OP_FP8_PREP = 27,   /* dsl_fp8_prep(p[0] = items_dev, i[0] = n_items, i[1] = cout_pad, i[2] = k, margin = the float whose bits are l[1]) */
       DSL_OP_QUANT_FP8_DELAYED = 28, /* dsl_quant_fp8_delayed(p[0] = x, p[1] = y, p[2] = partials, p[3] = scale_dev, l[0] = rows, i[0] = c, i[1] = ld_x,
                                       * i[2] = n_partials) */
       DSL_OP_FP8_COMB = 24,   /* dsl_fp8_comb(p[0] = winv, p[1] = comb, i[0] = n, p[2] = partials, i[2] = n_partials) */
       DSL_OP_QUANT_FP8_W = 23, /* dsl_quant_fp8_weights(p[0] = w, p[1] = w8, p[2] = comb, p[3] = bn_scale, i[0] = cout, i[1] = cout_pad, i[2] = k,
                                * inv_act_scale = the float whose bits are l[1]) */
       DSL_OP_BNECK = 26,      /* desc = dsl_bneck_desc -> dsl_bottleneck_fwd */
       DSL_OP_STEM_POOL = 25,  /* dsl_stem_pool(p[0] = img, p[1] = w_groups, l[0] / l[1] = scale / bias pointers, p[2] = out, i[0] = ld_out,
                                * i[1..3] = n, h, w, i[4] = half_last: dsl_stem_pool_half) */
       DSL_OP_PROF = 21 };     /* phase mark (dsl_prof_enable(3) only, else a no-op): i[0] = class >= 4, i[1] = 0 begin | 1 end, l[0] / l[1] =
                                * algorithmic FLOPs / bytes of the phase as IEEE doubles' bit patterns (begin only) */
typedef struct dsl_op {
  int32_t kind;
  int32_t i[7];            /* small integer arguments for the simple ops; i[6] = s > 0: run this op on the library's
                            * side stream s (1..3; independent work, e.g. weight gradients, overlapping the main chain) */
  const void* desc;        /* pointer to the op's descriptor (host memory, must stay alive) */
  void* p[4];              /* device pointers for the simple ops */
  int64_t l[2];
} dsl_op;
int dsl_run_ops(const dsl_op* ops, int n_ops, void* stream);
/* `stream` waits for named event slot `slot` (a DSL_OP_RECORD of an earlier dsl_run_ops): lets a communication stream
 * start a gradient bucket's all-reduce when the side stream has finished that bucket's weight gradients, whatever the
 * caller's compute stream is doing.  Returns 1 (and does nothing) if the slot was never recorded. */
int dsl_stream_wait_slot(int slot, void* stream);
/* Record named event slot `slot` (0..15) on `stream`: the counterpart of dsl_stream_wait_slot / DSL_OP_WAIT for a stream
 * the caller owns (the optimizer's stream marks "bucket updated"; the next forward list waits for it where it first reads
 * that bucket's weights). */
int dsl_stream_record_slot(int slot, void* stream);
/* The library's side stream `id` (1..3: 1 carries the weight gradients) of the current device as a hipStream_t, for work the
 * caller must queue in order with it (the optimizer step of a bucket whose last weight gradients run there). */
int dsl_side_stream(int id, void** stream_out);
/* Create the library's streams of the current device now and pick them so that the three that carry concurrent work (weight gradients,
 * second chain, frozen prefix) sit on hardware queues of their own, none of them `caller_stream`'s - measured with a 300 us spin kernel,
 * whatever other streams the process has created (api.hip side_init).  Optional (the first dsl_run_ops does it).  *distinct_out: how
 * many of the three were found (3 = all, -1 = option "stream_probe" is 0). */
int dsl_streams_init(void* caller_stream, int* distinct_out);
/* Which hardware queue the communication stream (dsl_side_stream(5)) shares - it is PLACED, not dealt (option "comm_queue", default 1):
 * 1 = the weight-gradient stream's, 2 = the second chain's, 3 = the frozen prefix's, 4 = the caller's; 0 = not placed (probe off or no
 * candidate found), -1 = the streams do not exist yet.  Side ids 2 and 3 are ONE stream (they only add ordering). */
int dsl_comm_stream_queue(void);
/* Device-side stand-in for a ring all-reduce of buf[0, n) (fp32, 16-byte aligned) on `stream`: `wgs` workgroups (RCCL: one per
 * channel) make `passes` value-preserving read-modify-write passes over the bucket - the HBM traffic and the CUs the collective's
 * kernels hold beside the backward pass, on a box with one GPU (bench.py extra.comm_proxy: mmdet/apis/train.py:92-96's DDP
 * all-reduce priced without a peer).  Not a collective: no data leaves the device. */
int dsl_comm_proxy(float* buf, long long n, int wgs, int passes, void* stream);

/* ------------------------------------------------------------------------------------------
 * Live kernel timing with HIP events (bench.py roofline): when enabled, each launch of the MFMA
 * conv kernels is bracketed by a hipEvent pair on the launch stream.  Classes: 0 = conv_pipe_kernel
 * <128,128,2,4,2> (forward + data gradient; the kernel with the largest share of the step), 1 = conv_pipe_kernel
 * <256,192,4,2,2> (the head / FPN tile), 2 = every other forward/dgrad conv kernel instance, 3 = the
 * weight-gradient kernels.
 * dsl_prof_enable(1) brackets only class 0 (cheap enough for a timed region), dsl_prof_enable(2) every class
 * (event pairs on concurrently running streams perturb the overlap).
 * dsl_prof_enable(3) brackets no kernel, only PHASES: DSL_OP_PROF ops in an op list mark begin / end of a stretch of the caller's
 * stream (classes 4.. : 4 = head forward - both towers, their GroupNorms and predictors, on two streams -, 5 = head backward data
 * path); with two streams running the same kernel class side by side, a launch's own duration says little about the rate the
 * phase achieves.
 * dsl_prof_read synchronises the events and returns per class: launches, total ms, algorithmic FLOPs.
 * ---------------------------------------------------------------------------------------- */
#define DSL_PROF_CLASSES 8
int dsl_prof_enable(int on);
int dsl_prof_reset(void);
int dsl_prof_read(int64_t* launches, double* ms, double* flops);
/* the same plus the algorithmic HBM bytes of the bracketed launches (tensors read + written once, real channels) */
int dsl_prof_read2(int64_t* launches, double* ms, double* flops, double* bytes);

/* hardware probes used by the tests */
int dsl_probe_tr16(const uint16_t* lds_image /* 4096 u16 */, const int32_t* lane_off /* 64 u16-offsets */,
                   uint16_t* out /* [64][4] */, void* stream);
/* xcc_of_block[b] = HW_REG_XCC_ID of workgroup b; every workgroup adds 1.0 to acc[xcc][0..255] (8 x 256 floats,
 * zeroed by the caller) with workgroup-scope atomics */
int dsl_probe_xcc(int32_t* xcc_of_block, float* acc, int nblocks, void* stream);
/* Runs nblocks one-per-CU workgroups on a temporary stream created with the given CU mask (NULL: unrestricted) and
 * returns per workgroup out[2b] = XCC id, out[2b+1] = HW_ID register (SE/SH/CU fields); synchronous. */
int dsl_probe_cu_mask(const uint32_t* mask, int nwords, int32_t* out /* [nblocks][2] */, int nblocks);

#ifdef __cplusplus
}
#endif
#endif
