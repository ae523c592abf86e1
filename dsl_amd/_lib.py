"""ctypes binding of libdsl_hip.so (include/dsl_hip.h).  There is NO fallback: if the HIP library is
missing, importing this module raises."""
import ctypes as C
import os

import torch  # noqa: F401  - FIRST: the library must bind to the HIP runtime torch ships (torch/lib/libamdhip64), not pull the system's
#                             copy into the process ahead of it: two runtimes in one process leave the second without a device

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('DSL_HIP_LIB') or os.path.join(HERE, 'lib', 'libdsl_hip.so')   # override: kernel ablation builds (tools/)
MAX_SEG = 5

if not os.path.exists(LIB_PATH):
    raise RuntimeError(
        f'{LIB_PATH} not found: the MI355X kernels are not built. Run `python -m dsl_amd.build` '
        '(needs hipcc); dsl_amd has no CPU or PyTorch fallback for its hot path.')
lib = C.CDLL(LIB_PATH)

I5 = C.c_int32 * MAX_SEG
F5 = C.c_float * MAX_SEG

# conv flags / op kinds (mirror the enums of dsl_hip.h)
CONV_RELU_OUT, CONV_RELU_IN, CONV_OUT_F32, CONV_MASK_FIRST, CONV_MASK_LAST, CONV_ADD_UPSAMPLE, \
    CONV_SMALL_C = 1, 2, 4, 8, 16, 32, 64
CONV_FP8 = 128
CONV_EPI_STAGED = 1 << 16
(OP_CONV, OP_WGRAD, OP_GN_FWD, OP_GN_BWD, OP_MAXPOOL, OP_SUM2X2, OP_COLSUM, OP_MEMSET, OP_PACK_IMAGE,
 OP_ASSIGN, OP_LOSS, OP_FORK, OP_JOIN, OP_WGRAD_GROUP, OP_RECORD, OP_WAIT) = range(1, 17)
OP_RLA = 17
OP_PACK_DGRAD = 18
OP_WGRAD_MULTI = 19
OP_PROF = 21
OP_QUANT_FP8, OP_QUANT_FP8_W, OP_FP8_COMB = 22, 23, 24
OP_FP8_PREP, OP_QUANT_FP8_DELAYED = 27, 28
OP_STEM_POOL = 25
OP_BNECK = 26
PROF_CLASSES = 8
MAX_MULTI = 16
SLOT_TAIL, SLOT_PREFIX = 13, 14     # pipelined frozen prefix: 'previous backward's data-gradient chain done', 'prefix of this step done'
SLOT_PACKS = 15      # named event: the data-gradient weight packs of the last optimizer step are complete
SUMSQ_PARTS = 256
SLOT_UPD = 8         # named events 8 .. 11: gradient bucket 0 .. 3 of the last optimizer step is updated (late exchange)
SLOT_HEADW = 12      # named event: the head + FPN bucket of the last optimizer step is updated (deferred head update)
(RLA_AVGPOOL, RLA_AVGPOOL_BWD, RLA_BN_TANH, RLA_BN_TANH_BWD, RLA_BN_FOLD, RLA_BN_POST, RLA_REC_SUM, RLA_TAIL_FWD, RLA_TAIL_BWD) = range(2, 11)
MAX_GROUP = 8


class ConvDesc(C.Structure):
    _fields_ = [('nseg', C.c_int32), ('n', C.c_int32),
                ('gh', I5), ('gw', I5), ('sh', I5), ('sw', I5), ('dh', I5), ('dw', I5), ('ah', I5), ('aw', I5),
                ('cs', C.c_int32), ('cd', C.c_int32), ('cd_pad', C.c_int32),
                ('ldd', C.c_int32), ('lda', C.c_int32), ('ldm', C.c_int32),
                ('kh', C.c_int32), ('kw', C.c_int32), ('stride', C.c_int32), ('pad', C.c_int32),
                ('mode', C.c_int32), ('os', C.c_int32), ('flags', C.c_int32),
                ('src', C.c_void_p), ('wgt', C.c_void_p), ('dst', C.c_void_p),
                ('scale', C.c_void_p), ('bias', C.c_void_p), ('addend', C.c_void_p), ('mask', C.c_void_p),
                ('workspace', C.c_void_p), ('workspace_bytes', C.c_size_t), ('cs_real', C.c_int32), ('lds', C.c_int32),
                ('gn_ws', C.c_void_p), ('gn_x', C.c_void_p), ('gn_gamma', C.c_void_p), ('gn_beta', C.c_void_p),
                ('gn_stats', C.c_void_p)]


class WgradDesc(C.Structure):
    _fields_ = [('nseg', C.c_int32), ('n', C.c_int32),
                ('gh', I5), ('gw', I5), ('sh', I5), ('sw', I5),
                ('cs', C.c_int32), ('cy', C.c_int32), ('cd', C.c_int32),
                ('kh', C.c_int32), ('kw', C.c_int32), ('stride', C.c_int32), ('pad', C.c_int32),
                ('splits', C.c_int32),
                ('dy', C.c_void_p), ('x', C.c_void_p), ('scale', C.c_void_p), ('dw', C.c_void_p),
                ('db', C.c_void_p), ('workspace', C.c_void_p), ('workspace_bytes', C.c_size_t),
                ('ldx', C.c_int32), ('shared', C.c_int32), ('slots', C.c_int32), ('pad_', C.c_int32),
                ('pixtab', C.c_void_p), ('pixtab_bytes', C.c_size_t)]


class GnDesc(C.Structure):
    _fields_ = [('nseg', C.c_int32), ('n', C.c_int32), ('c', C.c_int32), ('groups', C.c_int32),
                ('h', I5), ('w', I5), ('eps', C.c_float),
                ('x', C.c_void_p), ('y', C.c_void_p), ('gamma', C.c_void_p), ('beta', C.c_void_p),
                ('stats', C.c_void_p), ('dy', C.c_void_p), ('dx', C.c_void_p), ('dgamma', C.c_void_p),
                ('dbeta', C.c_void_p), ('dbias', C.c_void_p), ('workspace', C.c_void_p), ('workspace_bytes', C.c_size_t),
                ('conv_stats', C.c_int32), ('pad_', C.c_int32), ('y8', C.c_void_p), ('y8_scale', C.c_void_p), ('y8_amax', C.c_void_p)]


class Fp8PrepItem(C.Structure):
    _fields_ = [('w', C.c_void_p), ('w8', C.c_void_p), ('comb', C.c_void_p), ('amax', C.c_void_p), ('scale', C.c_void_p),
                ('n_amax', C.c_int32), ('cout', C.c_int32)]


class FcosDesc(C.Structure):
    _fields_ = [('nlvl', C.c_int32), ('n', C.c_int32),
                ('h', I5), ('w', I5), ('stride', I5), ('range_lo', F5), ('range_hi', F5),
                ('radius', C.c_float), ('num_classes', C.c_int32),
                ('gt_boxes', C.c_void_p), ('gt_labels', C.c_void_p), ('gt_off', C.c_void_p),
                ('ig_boxes', C.c_void_p), ('ig_off', C.c_void_p),
                ('labels', C.c_void_p), ('bbox_targets', C.c_void_p), ('assign_idx', C.c_void_p),
                ('cls_weight', C.c_void_p), ('pos_weight', C.c_void_p), ('stats', C.c_void_p),
                ('loss_weight', C.c_float),
                ('cls_logits', C.c_void_p), ('regctr', C.c_void_p),
                ('ld_cls', C.c_int32), ('ld_rc', C.c_int32),
                ('scales', C.c_void_p), ('norm', C.c_void_p),
                ('g_cls', C.c_void_p), ('ld_gcls', C.c_int32), ('g_rc', C.c_void_p), ('ld_grc', C.c_int32),
                ('g_scales', C.c_void_p), ('losses', C.c_void_p),
                ('soft_weight', C.c_float), ('grad_scale', C.c_float), ('inv_world', C.c_float),
                ('workspace', C.c_void_p), ('workspace_bytes', C.c_size_t), ('logvec', C.c_void_p)]


class DetDesc(C.Structure):
    _fields_ = [('nlvl', C.c_int32), ('n', C.c_int32),
                ('h', I5), ('w', I5), ('stride', I5),
                ('num_classes', C.c_int32), ('nms_pre', C.c_int32), ('max_per_img', C.c_int32),
                ('score_thr', C.c_float), ('iou_thr', C.c_float),
                ('cls_logits', C.c_void_p), ('ld_cls', C.c_int32),
                ('regctr', C.c_void_p), ('ld_rc', C.c_int32),
                ('scales', C.c_void_p), ('img_shapes', C.c_void_p), ('scale_factors', C.c_void_p),
                ('dets', C.c_void_p), ('det_labels', C.c_void_p), ('det_count', C.c_void_p),
                ('workspace', C.c_void_p), ('workspace_bytes', C.c_size_t)]


class PackItem(C.Structure):
    _fields_ = [('w', C.c_void_p), ('scale', C.c_void_p), ('out', C.c_void_p),
                ('cout', C.c_int32), ('cout_pad', C.c_int32), ('taps', C.c_int32), ('cin', C.c_int32),
                ('block_start', C.c_int32), ('tiles_ci', C.c_int32), ('tiles_co', C.c_int32), ('tapmap', C.c_int32)]


class BneckDesc(C.Structure):
    _fields_ = [('x', C.c_void_p), ('w1', C.c_void_p), ('w2', C.c_void_p), ('w3', C.c_void_p), ('idt', C.c_void_p),
                ('s1', C.c_void_p), ('b1', C.c_void_p), ('s2', C.c_void_p), ('b2', C.c_void_p), ('s3', C.c_void_p), ('b3', C.c_void_p),
                ('a1', C.c_void_p), ('a2', C.c_void_p), ('out', C.c_void_p),
                ('n', C.c_int32), ('hin', C.c_int32), ('win', C.c_int32), ('h', C.c_int32), ('w', C.c_int32),
                ('planes', C.c_int32), ('cin', C.c_int32), ('ldx', C.c_int32), ('stride', C.c_int32), ('ldi', C.c_int32), ('ldo', C.c_int32)]


class RlaDesc(C.Structure):
    _fields_ = [('kind', C.c_int32), ('i', C.c_int32 * 8), ('f', C.c_float * 2), ('rows', C.c_int64), ('p', C.c_void_p * 12)]


class BnPostItem(C.Structure):
    _fields_ = [('w', C.c_void_p), ('dw', C.c_void_p), ('dgamma', C.c_void_p), ('dbeta', C.c_void_p), ('gamma', C.c_void_p),
                ('mean', C.c_void_p), ('var', C.c_void_p), ('rows', C.c_int32), ('k', C.c_int32), ('row_start', C.c_int32),
                ('pad_', C.c_int32)]


class RecSumItem(C.Structure):
    _fields_ = [('rec', C.c_void_p), ('out_a', C.c_void_p), ('out_b', C.c_void_p), ('nrec', C.c_int32), ('pad_', C.c_int32)]


class ImagePrepItem(C.Structure):
    _fields_ = [('src', C.c_void_p), ('src_h', C.c_int32), ('src_w', C.c_int32), ('new_h', C.c_int32), ('new_w', C.c_int32),
                ('flip', C.c_int32), ('ps_mode', C.c_int32), ('ps_crop', C.c_int32), ('to_rgb', C.c_int32),
                ('mean', C.c_float * 3), ('inv_std', C.c_float * 3)]


class AugItem(C.Structure):
    _fields_ = [('h', C.c_int32), ('w', C.c_int32), ('kind', C.c_int32), ('order', C.c_int32), ('m', C.c_float * 6),
                ('roi', C.c_int32 * 4), ('cval', C.c_int32), ('f', C.c_float * 3), ('rect', (C.c_int32 * 4) * 3), ('seed', C.c_uint32)]


(AUG_COPY, AUG_AFFINE, AUG_BRIGHTNESS, AUG_CONTRAST, AUG_SATURATION, AUG_HUE, AUG_GRAY, AUG_BLUR_H, AUG_BLUR_V, AUG_ERASE,
 AUG_AUTOCONTRAST, AUG_EQUALIZE, AUG_SOLARIZE, AUG_POSTERIZE, AUG_SHARPNESS) = range(15)


class Op(C.Structure):
    _fields_ = [('kind', C.c_int32), ('i', C.c_int32 * 7), ('desc', C.c_void_p),
                ('p', C.c_void_p * 4), ('l', C.c_int64 * 2)]


lib.dsl_last_error.restype = C.c_char_p
lib.dsl_wgrad_workspace_bytes.restype = C.c_size_t
if hasattr(lib, 'dsl_wgrad_pixtab_bytes'):
    lib.dsl_wgrad_pixtab_bytes.restype = C.c_size_t
if hasattr(lib, 'dsl_wgrad_group_workspace_bytes'):
    lib.dsl_wgrad_group_workspace_bytes.restype = C.c_size_t
lib.dsl_conv2d_workspace_bytes.restype = C.c_size_t
if hasattr(lib, 'dsl_image_aug_scratch_bytes'):
    lib.dsl_image_aug_scratch_bytes.restype = C.c_size_t
if hasattr(lib, 'dsl_rla_tail_bwd_workspace_bytes'):
    lib.dsl_rla_tail_bwd_workspace_bytes.restype = C.c_size_t
for _n in ('dsl_wgrad_multi_table_bytes', 'dsl_wgrad_multi_workspace_bytes'):
    if hasattr(lib, _n):
        getattr(lib, _n).restype = C.c_size_t
if hasattr(lib, 'dsl_bn_tanh_bwd_workspace_bytes'):
    lib.dsl_bn_tanh_bwd_workspace_bytes.restype = C.c_size_t
if hasattr(lib, 'dsl_fcos_workspace_bytes'):
    lib.dsl_fcos_workspace_bytes.restype = C.c_size_t
if hasattr(lib, 'dsl_groupnorm_workspace_bytes'):
    lib.dsl_groupnorm_workspace_bytes.restype = C.c_size_t
if hasattr(lib, 'dsl_detect_workspace_bytes'):
    lib.dsl_detect_workspace_bytes.restype = C.c_size_t
_vp, _i, _l, _f = C.c_void_p, C.c_int, C.c_long, C.c_float
_SIGS = {
    'dsl_conv2d': [_vp, _vp], 'dsl_conv2d_workspace_bytes': [_vp], 'dsl_conv2d_gn_fusable': [_vp], 'dsl_conv2d_wgrad': [_vp, _vp], 'dsl_wgrad_splits': [_vp],
    'dsl_wgrad_workspace_bytes': [_vp], 'dsl_wgrad_pixtab_bytes': [_vp], 'dsl_wgrad_pixtab_fill': [_vp, _vp, C.c_size_t, _vp], 'dsl_wgrad_group_workspace_bytes': [_vp, _i], 'dsl_conv2d_wgrad_group': [_vp, _i, _vp],
    'dsl_wgrad_multi_config': [_vp], 'dsl_wgrad_multi_table_bytes': [], 'dsl_wgrad_multi_workspace_bytes': [_vp, _vp, _i],
    'dsl_wgrad_multi_build': [_vp, _vp, _i, _vp, C.c_size_t, _vp, C.c_size_t], 'dsl_conv2d_wgrad_multi': [_vp, _vp, _vp], 'dsl_wgrad_multi_info': [_vp, _vp, _vp, _vp, _vp, _vp],
    'dsl_wgrad_plan_probe': [_vp, _vp, _vp, _i, _i, _i, _vp, _vp],
    'dsl_image_prep': [_vp, _i, _vp, _i, _i, _vp],
    'dsl_pack_image': [_vp, _vp, _i, _i, _i, _vp], 'dsl_stem_pool': [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp], 'dsl_stem_pool_half': [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp], 'dsl_maxpool3x3s2': [_vp, _vp, _i, _i, _i, _i, _vp],
    'dsl_maxpool3x3s2_ld': [_vp, _vp, _i, _i, _i, _i, _i, _vp],
    'dsl_avgpool2x2': [_vp, _i, _vp, _i, _i, _i, _i, _i, _vp],
    'dsl_avgpool2x2_bwd': [_vp, _i, _vp, _i, _i, _i, _i, _i, _vp],
    'dsl_bn_tanh_fwd': [_vp, _i, _vp, _vp, _vp, _i, _l, _i, _vp],
    'dsl_rla_tail_bwd': [_vp, _i, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _f, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _vp], 'dsl_rla_tail_bwd_workspace_bytes': [_i, _i, _i],
    'dsl_rla_tail_fwd': [_vp, _i, _vp, _i, _vp, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _vp], 'dsl_bn_tanh_bwd_workspace_bytes': [_l, _i],
    'dsl_bn_tanh_bwd': [_vp, _i, _vp, _i, _vp, _i, _vp, _vp, _vp, _f, _vp, _i, _vp, _vp, _vp, _l, _i, _vp],
    'dsl_image_prep_u8': [_vp, _i, _vp, _i, _i, _vp], 'dsl_image_aug': [_vp, _i, _vp, _vp, _i, _i, _vp, _i, _vp], 'dsl_image_aug_scratch_bytes': [_i],
    'dsl_image_normalize': [_vp, _vp, _i, _vp, _i, _i, _vp],
    'dsl_quant_fp8': [_vp, _vp, _l, _i, _i, _f, _vp], 'dsl_absmax': [_vp, _l, _i, _i, _vp, _i, _vp],
    'dsl_quant_fp8_dyn': [_vp, _vp, _l, _i, _i, _vp, _i, _vp], 'dsl_fp8_comb': [_vp, _vp, _i, _vp, _i, _vp], 'dsl_fp8_prep': [_vp, _i, _i, _i, _f, _vp],
    'dsl_quant_fp8_delayed': [_vp, _vp, _l, _i, _i, _vp, _vp, _i, _vp], 'dsl_quant_fp8_weights': [_vp, _vp, _vp, _vp, _i, _i, _i, _f, _vp],
    'dsl_bn_fold': [_vp, _vp, _vp, _vp, _f, _vp, _vp, _i, _vp], 'dsl_bn_wgrad_post': [_vp, _i, _i, _f, _vp], 'dsl_rla_op': [_vp, _vp],
    'dsl_rec_sum_multi': [_vp, _i, _i, _vp],
    'dsl_groupnorm_relu_fwd': [_vp, _vp], 'dsl_groupnorm_relu_bwd': [_vp, _vp], 'dsl_groupnorm_workspace_bytes': [_vp],
    'dsl_sum2x2': [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp], 'dsl_colsum': [_vp, _vp, _l, _i, _i, _vp],
    'dsl_fcos_points': [_vp, _vp, _vp], 'dsl_fcos_workspace_bytes': [_vp], 'dsl_fcos_assign': [_vp, _vp], 'dsl_fcos_loss': [_vp, _vp],
    'dsl_sumsq': [_vp, _l, _vp, _vp], 'dsl_sumsq_det': [_vp, _l, _vp, _vp, _vp],
    'dsl_sgd_step': [_vp, _vp, _vp, _vp, _vp, _l, _f, _f, _f, _f, _f, _vp, _f, _i, _vp],
    'dsl_ema_lerp': [_vp, _vp, _l, _f, _vp], 'dsl_ema_lerp_bf16': [_vp, _vp, _vp, _l, _f, _vp], 'dsl_cast_bf16': [_vp, _vp, _l, _vp],
    'dsl_pack_dgrad': [_vp, _vp, _vp, _i, _i, _i, _i, _vp], 'dsl_pack_dgrad_batched': [_vp, _i, _i, _vp],
    'dsl_detect_workspace_bytes': [_vp], 'dsl_fcos_detect': [_vp, _vp],
    'dsl_pseudo_label_fuse': [_vp, _vp, _vp, _i, _i, _i, _f, _f, _f, _vp, _vp, _vp, _vp, _vp],
    'dsl_comm_unique_id': [_vp],
    'dsl_comm_init_rank': [_vp, _i, _vp, _i],
    'dsl_comm_size': [_vp],
    'dsl_comm_destroy': [_vp],
    'dsl_allreduce_bucket': [_vp, _vp, C.c_size_t, _vp],
    'dsl_allreduce_buckets': [_vp, _vp, _vp, _i, _vp],
    'dsl_allreduce_bucket_bf16': [_vp, _vp, C.c_size_t, _vp],
    'dsl_cast_f32': [_vp, _vp, _l, _vp], 'dsl_sumsq_partial': [_vp, _l, _vp, _vp], 'dsl_sumsq_fold': [_vp, _i, _vp, _vp],
    'dsl_pseudo_label_fuse_history': [_vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _i, _i, _f, _f, _f, _vp, _vp, _vp, _vp, _i, _vp],
    'dsl_bottleneck_fwd': [_vp, _vp], 'dsl_bottleneck_fwd_supported': [_vp],
    'dsl_set_option': [C.c_char_p, _i], 'dsl_get_option': [C.c_char_p, _vp],
    'dsl_run_ops': [_vp, _i, _vp], 'dsl_stream_wait_slot': [_i, _vp], 'dsl_stream_record_slot': [_i, _vp], 'dsl_side_stream': [_i, _vp], 'dsl_streams_init': [_vp, _vp], 'dsl_comm_stream_queue': [], 'dsl_comm_proxy': [_vp, C.c_longlong, _i, _i, _vp], 'dsl_prof_enable': [_i], 'dsl_prof_reset': [], 'dsl_prof_read': [_vp, _vp, _vp], 'dsl_prof_read2': [_vp, _vp, _vp, _vp], 'dsl_probe_tr16': [_vp, _vp, _vp, _vp], 'dsl_probe_xcc': [_vp, _vp, _i, _vp], 'dsl_probe_cu_mask': [_vp, _i, _vp, _i],
}
MISSING = []
for _name, _args in _SIGS.items():
    try:
        getattr(lib, _name).argtypes = _args
    except AttributeError:      # reported by tests/test_abi.py; calling it raises AttributeError
        MISSING.append(_name)


if 'dsl_set_option' not in MISSING:
    from . import tuning as _tuning
    _tuning.push_lib_options(lib)       # DSL_TUNE's lib.* keys -> the library's option table (it reads no environment variable)


def check(rc, what=''):
    if rc != 0:
        raise RuntimeError(f'libdsl_hip {what} failed ({rc}): {lib.dsl_last_error().decode()}')


def seg5(vals):
    a = I5()
    for i, v in enumerate(vals):
        a[i] = int(v)
    return a


def stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    if t is None:
        return C.c_void_p(0)
    if isinstance(t, int):
        return C.c_void_p(t)
    return C.c_void_p(t.data_ptr())
