"""Registered detector classes with the reference's names and constructor arguments, so that
configs/fcos_semi/*.py build unmodified:

  FCOS / SingleStageDetector   mmdet/models/detectors/{fcos,single_stage,base}.py
  ResNet                        mmdet/models/backbones/resnet.py:304-656   (depth 50, caffe, frozen BN)
  FPN                           mmdet/models/necks/fpn.py:9-202
  FCOSHead                      mmdet/models/dense_heads/fcos_head.py:14-726
  FocalLoss / GIoULoss / CrossEntropyLoss   mmdet/models/losses/*.py

Unlike the reference these classes do not compute with torch ops: they validate the configuration
the HIP path implements, own the parameter store, and drive the prebuilt kernel lists of
dsl_amd.engine.  There is no CPU / PyTorch fallback: calling the hot path without the HIP library or
off-GPU raises.
"""
from collections import OrderedDict

import ctypes
import os

import torch
import torch.distributed as dist
import torch.nn as nn

C_void = ctypes.c_void_p

from .params import ParamStore
from .registry import BACKBONES, DETECTORS, HEADS, LOSSES, NECKS, build_backbone, build_head, build_loss, build_neck


# mmcv's model-zoo aliases used by configs/fcos_semi (mmcv/model_zoo/open_mmlab.json, mmcv 1.3.10): the file a
# torch.hub download would leave in the checkpoint cache
_ZOO_FILES = {'open-mmlab://detectron2/resnet50_caffe': 'resnet50_msra-5891d200.pth'}


def resolve_checkpoint(uri):
    """Local file for a `checkpoint=` value: a path, or a model-zoo alias looked up in $DSL_PRETRAINED_DIR and the torch
    hub cache (there is no network on the training boxes: the file has to be there already).  None when not found."""
    import os
    if os.path.exists(uri):
        return uri
    name = _ZOO_FILES.get(uri, os.path.basename(uri))
    for d in (os.environ.get('DSL_PRETRAINED_DIR'), os.path.join(torch.hub.get_dir(), 'checkpoints'),
              os.path.expanduser('~/.cache/torch/checkpoints')):
        if d and os.path.exists(os.path.join(d, name)):
            return os.path.join(d, name)
    return None


def load_backbone_checkpoint(det, uri, strict=False):
    """Backbone-only checkpoint (keys un-prefixed: conv1.weight, layer1.0.bn1.running_mean, ...; optional
    'state_dict' wrapper and 'module.' / 'backbone.' prefixes as mmcv's load_checkpoint strips them) into the flat
    store.  A checkpoint that cannot be found is NOT silently ignored: with a frozen stem / layer1 / BatchNorms the
    model would train on random frozen features."""
    import warnings
    path = resolve_checkpoint(uri)
    if path is None:
        warnings.warn(f'dsl_amd: pretrained backbone checkpoint {uri!r} not found (looked in $DSL_PRETRAINED_DIR and the '
                      'torch hub cache; no network here): the FROZEN stem/layer1/BatchNorm tensors keep their random '
                      'reference-style initialisation', RuntimeWarning, stacklevel=2)
        return False
    ck = torch.load(path, map_location='cpu')
    sd = ck.get('state_dict', ck.get('model', ck)) if isinstance(ck, dict) else ck
    views = det.store.named_views()
    out, unexpected = {}, []
    for k, v in sd.items():
        if not isinstance(v, torch.Tensor):
            v = torch.as_tensor(v)
        for pre in ('module.', 'backbone.'):
            if k.startswith(pre):
                k = k[len(pre):]
        kk = 'backbone.' + k
        if kk in views and tuple(views[kk].shape) == tuple(v.shape):
            out[kk] = v
        else:
            unexpected.append(k)
    missing = [k for k in views if k.startswith('backbone.') and k not in out and not k.endswith('num_batches_tracked')]
    if strict and (missing or unexpected):
        raise KeyError(f'backbone checkpoint mismatch: missing {missing[:5]}, unexpected {unexpected[:5]}')
    if not out:
        raise KeyError(f'{path}: no key of this checkpoint matches the backbone (first keys: {list(sd)[:5]})')
    full = {k: v for k, v in views.items()}
    full.update(out)
    det.store.load_named(full, strict=False)
    return True


def _expect(cond, msg):
    if not cond:
        raise NotImplementedError('dsl_amd hot path: ' + msg)


@BACKBONES.register_module()
class ResNet(nn.Module):
    def __init__(self, depth=50, num_stages=4, out_indices=(0, 1, 2, 3), frozen_stages=-1,
                 norm_cfg=dict(type='BN', requires_grad=True), norm_eval=True, style='pytorch', init_cfg=None,
                 pretrained=None, **kw):
        super().__init__()
        _expect(depth == 50 and num_stages == 4 and tuple(out_indices) == (0, 1, 2, 3), 'ResNet-50, 4 stages, all outputs')
        _expect(style == 'caffe', "style='caffe' (stride on the first 1x1)")
        _expect(frozen_stages == 1 and norm_eval and not norm_cfg.get('requires_grad', True),
                'frozen_stages=1, norm_eval=True, BN requires_grad=False (configs/fcos_semi/r50_caffe_*.py:4-15)')
        _expect(not kw.get('dcn') and not kw.get('plugins') and not kw.get('with_cp', False), 'no DCN/plugins/checkpointing')
        self.init_cfg = init_cfg
        self.pretrained_checkpoint = pretrained
        if isinstance(init_cfg, dict) and init_cfg.get('type') == 'Pretrained':
            self.pretrained_checkpoint = init_cfg['checkpoint']


@BACKBONES.register_module()
class RLA_ResNet(nn.Module):
    """mmdet/models/backbones/resnet_rla.py:140-388 with the arguments of configs/fcos_semi/RLA_*.py:3-13: layers
    [3, 4, 6, 3], rla_channel 32, no SE / ECA, frozen_stages 1, norm_eval, style 'pytorch' (the 3x3 convolutions stride).
    The computation is dsl_amd/engine_rla.py; this class validates the configuration and names the checkpoint."""
    backbone_kind = 'rla'

    def __init__(self, block=None, layers=(3, 4, 6, 3), num_classes=1000, rla_channel=32, SE=False, ECA=None,
                 frozen_stages=-1, norm_eval=True, style='pytorch', zero_init_last_bn=True, groups=1, width_per_group=64,
                 replace_stride_with_dilation=None, norm_layer=None, pretrained=None, init_cfg=None, **kw):
        super().__init__()
        _expect(block is None and list(layers) == [3, 4, 6, 3] and rla_channel == 32, 'RLA_Bottleneck, layers [3, 4, 6, 3], 32 RLA channels')
        _expect(not SE and ECA is None and groups == 1 and width_per_group == 64 and replace_stride_with_dilation is None
                and norm_layer is None, 'no SE / ECA, no groups, no dilation, BatchNorm2d')
        _expect(frozen_stages == 1 and norm_eval and style == 'pytorch', "frozen_stages=1, norm_eval=True, style='pytorch'")
        self.zero_init_last_bn = zero_init_last_bn
        self.pretrained_checkpoint = pretrained
        if isinstance(init_cfg, dict) and init_cfg.get('type') == 'Pretrained':
            self.pretrained_checkpoint = init_cfg['checkpoint']


@NECKS.register_module()
class FPN(nn.Module):
    def __init__(self, in_channels, out_channels, num_outs, start_level=0, end_level=-1, add_extra_convs=False,
                 relu_before_extra_convs=False, no_norm_on_lateral=False, conv_cfg=None, norm_cfg=None, act_cfg=None,
                 upsample_cfg=dict(mode='nearest'), init_cfg=None, **kw):
        super().__init__()
        _expect(list(in_channels) == [256, 512, 1024, 2048] and out_channels == 256 and num_outs == 5 and start_level == 1,
                'FPN 256ch, start_level=1, 5 outputs')
        _expect(add_extra_convs == 'on_output' and relu_before_extra_convs and norm_cfg is None and act_cfg is None,
                "add_extra_convs='on_output', relu_before_extra_convs=True, no norm/act")
        _expect(upsample_cfg.get('mode', 'nearest') == 'nearest' and 'scale_factor' not in upsample_cfg, 'nearest upsample')


class _LossCfg(nn.Module):
    def __init__(self, loss_weight=1.0, **kw):
        super().__init__()
        _expect(loss_weight == 1.0 and kw.get('reduction', 'mean') == 'mean', 'loss_weight=1.0, reduction=mean')
        self.loss_weight = loss_weight


@LOSSES.register_module()
class FocalLoss(_LossCfg):
    def __init__(self, use_sigmoid=True, gamma=2.0, alpha=0.25, **kw):
        super().__init__(**kw)
        _expect(use_sigmoid and gamma == 2.0 and alpha == 0.25, 'sigmoid focal loss, gamma 2, alpha 0.25')


@LOSSES.register_module()
class GIoULoss(_LossCfg):
    def __init__(self, eps=1e-6, **kw):
        super().__init__(**kw)
        _expect(eps == 1e-6, 'GIoU eps 1e-6')


@LOSSES.register_module()
class CrossEntropyLoss(_LossCfg):
    def __init__(self, use_sigmoid=False, use_mask=False, class_weight=None, **kw):
        super().__init__(**kw)
        _expect(use_sigmoid and not use_mask and class_weight is None, 'sigmoid BCE centerness loss')


@HEADS.register_module()
class FCOSHead(nn.Module):
    def __init__(self, num_classes, in_channels, regress_ranges=((-1, 64), (64, 128), (128, 256), (256, 512), (512, 1e8)),
                 center_sampling=False, center_sample_radius=1.5, norm_on_bbox=False, centerness_on_reg=False,
                 loss_weight=1.0, soft_weight=0.0, soft_warm_up=0, feat_channels=256, stacked_convs=4,
                 strides=(4, 8, 16, 32, 64), dcn_on_last_conv=False, conv_bias='auto',
                 loss_cls=dict(type='FocalLoss', use_sigmoid=True, gamma=2.0, alpha=0.25, loss_weight=1.0),
                 loss_bbox=dict(type='IoULoss', loss_weight=1.0),
                 loss_centerness=dict(type='CrossEntropyLoss', use_sigmoid=True, loss_weight=1.0),
                 norm_cfg=dict(type='GN', num_groups=32, requires_grad=True), train_cfg=None, test_cfg=None,
                 init_cfg=None, **kw):
        super().__init__()
        _expect(num_classes == 80 and in_channels == 256 and feat_channels == 256 and stacked_convs == 4, '80 classes, 256ch, 4 convs')
        _expect(list(strides) == [8, 16, 32, 64, 128], 'strides 8..128')
        _expect(center_sampling and norm_on_bbox and centerness_on_reg and not dcn_on_last_conv and conv_bias is True,
                'the fcos_semi "tricks" head: center_sampling, norm_on_bbox, centerness_on_reg, conv_bias=True')
        _expect(norm_cfg.get('type') == 'GN' and norm_cfg.get('num_groups') == 32, 'GN-32 towers')
        self.num_classes, self.strides = num_classes, tuple(strides)
        self.regress_ranges = tuple(tuple(r) for r in regress_ranges)
        self.center_sample_radius = center_sample_radius
        self.loss_weight, self.soft_weight, self.soft_warm_up = loss_weight, soft_weight, soft_warm_up
        self.cur_iter = 0                               # fcos_head.py:103
        self.loss_cls, self.loss_bbox = build_loss(loss_cls), build_loss(loss_bbox)
        self.loss_centerness = build_loss(loss_centerness)
        _expect(isinstance(self.loss_cls, FocalLoss) and isinstance(self.loss_bbox, GIoULoss)
                and isinstance(self.loss_centerness, CrossEntropyLoss), 'FocalLoss + GIoULoss + CrossEntropyLoss')
        self.train_cfg, self.test_cfg = train_cfg, test_cfg

    def effective_soft_weight(self, batch_size):
        """fcos_head.py:312-327: the sisoft term exists for odd batches with soft_weight != 0; while
        cur_iter <= soft_warm_up it is scaled by 1/1000 and the counter advances."""
        if batch_size % 2 == 0 or self.soft_weight == 0.0:
            return 0.0
        w = self.soft_weight * 1.0
        if self.soft_warm_up >= self.cur_iter:
            self.cur_iter += 1
            w = self.soft_weight / 1000.0
        return w


# One stream per ROLE and device for the whole process, not per model instance (round 4).  Streams are dealt onto four hardware
# queues in creation order; which streams share a queue moves the step by up to 20 % (350 vs 427 img/s for one stream
# created too early).  The first detector of a process creates the streams in the order the measured layout came about; every later
# instance (a second model in a notebook, bench.py's extra timings, student + teacher) must find the SAME streams instead of drawing
# new ones from torch's pool - measured: the same loop on a second model in one process ran at 433 or 362 img/s depending on which
# pool stream it happened to get (profiles/r04_td_first.txt).
_ROLE_STREAMS = {}
_ROLE_IDS = dict(prefix=4, comm=5, sweep=6, optimizer=1)      # the library's stream ids (csrc/api.hip side_init)


def role_stream(role, device=None):
    """The library's own stream for `role` on `device`: created and PICKED by the C library together with its side streams (csrc/api.hip
    side_init: a spin-kernel probe finds streams on hardware queues of their own) - not drawn from torch's pool, whose streams share
    hardware queues with whatever else the process created.  The optimizer's per-bucket updates run on the weight-gradient stream itself (id 1): each follows the weight
    gradients it waits for, which is where a stream of its own ended up anyway (one hardware queue with stream 1)."""
    import ctypes as C
    from . import _lib as L
    dev = torch.cuda.current_device() if device is None else torch.device(device).index
    key = (role, dev)
    st = _ROLE_STREAMS.get(key)
    if st is None:
        h = C.c_void_p()
        with torch.cuda.device(dev):
            L.check(L.lib.dsl_streams_init(L.stream_ptr(), None), 'dsl_streams_init')     # (picks the streams against the caller's: first call only)
            L.check(L.lib.dsl_side_stream(_ROLE_IDS[role], C.byref(h)), 'dsl_side_stream')
        st = torch.cuda.ExternalStream(h.value, device=dev)
        _ROLE_STREAMS[key] = st
    return st


def _check_grad_now(det, who='total'):
    """Should this backward call verify its incoming gradient (one host sync)?  Always when the step syncs anyway or the tuning key
    asks for it; otherwise on the first three backward calls of EACH autograd bridge (`who`: one backward pass goes through
    _TotalFn and _TrainStepFn - a budget shared between them was spent within two passes)."""
    from .tuning import tune
    if not det.lazy_log or tune('check_backward_grad') != '0':
        return True
    budget = det.__dict__.setdefault('_grad_checks_left', {})
    left = budget.get(who, 3)
    if left > 0:
        budget[who] = left - 1
        return True
    return False


class _LossDict(OrderedDict):
    """The loss dict of forward_train; `.vec` (optional) holds the same scalars as one graph-connected tensor, `.vec_total` the same
    with their sum appended."""
    vec = None
    vec_total = None


class _TotalFn(torch.autograd.Function):
    """total = the last element of the step's log vector (summed by the loss kernel), connected to the graph without a device op in
    either direction: backward hands the cached one-hot gradient on (a gradient other than 1 is rejected where the step reads
    values back anyway, see _TrainStepFn.backward)."""

    @staticmethod
    def forward(ctx, vec, det):
        ctx.det, ctx.n = det, vec.numel()
        return vec[ctx.n - 1].detach()

    @staticmethod
    def backward(ctx, g):
        det = ctx.det
        if _check_grad_now(det):
            if float(g) != 1.0:
                raise NotImplementedError(f'dsl_amd: loss.backward() with a gradient of {float(g)} for the total loss: scale through '
                                          'FCOS.loss_scale (folded into the loss kernel), not by scaling the loss tensor')
        key = (ctx.n, g.device)
        oh = det._onehot.get(key)
        if oh is None:
            oh = torch.zeros(ctx.n, dtype=torch.float32, device=g.device)
            oh[ctx.n - 1] = 1.0
            det._onehot[key] = oh
        return oh, None


class _TrainStepFn(torch.autograd.Function):
    """Bridges `loss.backward()` (mmcv OptimizerHook) to the hand-written backward kernel lists."""

    @staticmethod
    def forward(ctx, anchor, det, plan):
        ctx.det, ctx.plan, ctx.eager = det, plan, bool(getattr(det, '_eager_now', False))      # were the backward lists queued already?
        ctx.n_losses = 4 if plan.lossplan.desc.soft_weight != 0.0 else 3
        # the loss terms in log order and, last, their sum - written by the loss kernel's finalize pass (dsl_fcos_desc.logvec)
        return plan.lossplan.logvec[:ctx.n_losses + 1].clone()

    @staticmethod
    def backward(ctx, g):
        """The backward lists compute d(sum of the losses)/d(parameters): the loss kernel already produced the head
        gradients for d(total)/d(loss_k) = grad_scale (1/world).  Any other incoming gradient - loss scaling, loss / k for
        gradient accumulation, re-weighted loss keys - is NOT representable: it is rejected instead of silently ignored.
        The check reads g back (one host sync), so it runs whenever the step syncs anyway (lazy_log False), on a detector's first
        backward calls (a loop that scales its loss does so from the first step on) and on every step under the tuning key
        check_backward_grad=1; fold a constant factor into `FCOS.loss_scale` instead."""
        det = ctx.det
        if _check_grad_now(det, 'train_step'):
            k = ctx.n_losses
            gh = g.tolist()       # d(total)/d(term) = 1 for every term: either through the terms or through the kernel's own sum
            if not (all(v == 1 for v in gh[:k]) and gh[k] == 0) and not (all(v == 0 for v in gh[:k]) and gh[k] == 1):
                raise NotImplementedError(
                    'dsl_amd: loss.backward() reached the HIP step with a gradient != 1 for its loss terms '
                    f'({gh}): scale through FCOS.loss_scale (folded into the loss kernel), not by scaling the loss tensor')
        if not ctx.eager:                 # eager: the kernels were queued right behind the loss kernel
            det._run_backward(ctx.plan)
        return None, None, None


@DETECTORS.register_module()
class FCOS(nn.Module):
    """SingleStageDetector (detectors/single_stage.py:10-165) specialised by detectors/fcos.py:5-18."""

    def __init__(self, backbone, neck=None, bbox_head=None, train_cfg=None, test_cfg=None, pretrained=None,
                 init_cfg=None, fp8=None):
        """fp8 (not a key of the reference, whose training is fp32): dict(layers='towers') runs the FORWARD of the head towers' 3x3
        convolutions on the MX-scaled fp8 MFMA (OCP e4m3 activations with a dynamic per-tensor scale 448 / max|x| of the very
        tensor, per-output-channel weight scales re-made from the fp32 weights every step; the backward pass stays bf16 on the
        bf16 tensors - straight-through).  None (default): everything bf16."""
        super().__init__()
        self.backbone = build_backbone(backbone)
        self.neck = build_neck(neck)
        bbox_head = dict(bbox_head)
        bbox_head.update(train_cfg=train_cfg, test_cfg=test_cfg)
        self.bbox_head = build_head(bbox_head)
        self.train_cfg, self.test_cfg = train_cfg, test_cfg
        self.store = ParamStore(self.bbox_head.num_classes, 'cpu', backbone=getattr(self.backbone, 'backbone_kind', 'resnet'))
        self.store.init_reference_style(0)
        if fp8:
            assert str(fp8.get('layers', 'towers')) == 'towers', "fp8: only layers='towers' is built"
            self.store.fp8 = dict(fp8)
        self._params = None
        self._engine = None
        self._anchor = None
        self.dist_group = None        # set by the DDP wrapper
        self.world_size = 1
        self.CLASSES = None
        # Product defaults = what bench.py measures (round 4; both were opt-in and only bench.py set them):
        # lazy_log True: train_step's log_vars stay 0-dim device tensors; the host reads them when it logs (TextLoggerHook: every
        # `interval` iterations, float(v)) instead of once per iteration between the forward and the backward pass
        # (detectors/base.py:175-208 calls .item() per key per iteration).  False: python floats, one host sync per iteration.
        self.lazy_log = True
        # SemiEpochBasedRunner(scale_invariant=True) may hand the batch over WITHOUT its half-scale third image: the stem kernel reads
        # it out of the second one (forward_train(half_scale_copy=True)); set it False and the runner builds the copy with framework ops
        self.half_scale_in_stem = True
        self._onehot = {}          # cached gradient of the total loss w.r.t. the step's log vector (_TotalFn)
        # eager_backward True: INSIDE train_step the backward kernel lists are queued right behind the loss kernel (and the few
        # log-variable ops) instead of when `loss.backward()` reaches the autograd bridge.  The gradient of the summed loss is 1
        # either way - the bridge ignores its incoming gradient - so only the launch order changes.  For loops that call
        # loss.backward() once per train_step, as mmcv's OptimizerHook does (mmdet/apis/train.py:126-176); a caller that scales the
        # loss tensor instead of FCOS.loss_scale must set this False (tuning key check_backward_grad=1 checks the incoming gradient on every step; by default the first three are checked).
        # A direct forward_train() call (no train_step around it) stays lazy unless eager_backward == 'always'.
        self.eager_backward = True
        self._in_train_step, self._deferred_plan = False, None
        # The frozen prefix of the forward pass - image layout, stem, pool, layer1 - runs on its own stream; when the batch's image
        # tensor carries its producer's event (dsl_amd.data.mark_ready: resident inputs, a loader that renders on its own stream)
        # it starts under the tail of the PREVIOUS step's backward pass and optimizer step, otherwise it is ordered behind
        # everything queued on the caller's stream (always correct, no overlap)
        self.pipeline_prefix = True
        self._prefix_stream = None
        self.loss_scale = 1.0         # constant factor on every gradient (gradient accumulation: 1/k); the reported losses stay unscaled
        self._pending = []
        self._comm_stream = None
        self.rccl = None              # parallel.RcclComm: the exchanges go through the C-ABI's communicator instead of torch.distributed
        self.comm_trace = None        # set to [] (bench.py --gpus N): per step, per gradient bucket, timed events of its all-reduce
        # data-parallel options (set by HipDistributedDataParallel)
        self.grad_bf16 = False        # gradient buckets cross xGMI as bf16 copies (half the bytes); master gradient, norm, update stay fp32
        self.comm_off = False         # DSL_COMM=none / bench.py's attribution probe: every collective skipped (timing only - WRONG gradients)
        # Communication proxy (one-GPU measurement of what the collectives cost the DEVICE; bench.py extra.comm_proxy):
        # dict(carrier='lib' | 'torch', wgs=32, passes=2).  The step then runs its data-parallel schedule - bucket events, communication
        # stream, per-bucket optimizer steps behind each bucket's "all-reduce" - with dsl_comm_proxy (value-preserving passes over the
        # bucket on `wgs` workgroups) in place of RCCL; carrier 'lib' = the library's placed communication stream (what comm='rccl'
        # uses), 'torch' = a stream from torch's pool, as ProcessGroupNCCL runs its kernels on one of its own
        self.comm_proxy = None
        self._proxy_stream = None
        self.clip_partials = None     # [buckets * SUMSQ_PARTS] floats: FlatSGD with grad_clip asks for the norm in pieces, per bucket
        self._partials_valid = False
        self._g16 = None

    # ---- nn.Module surface redirected to the flat store ------------------------------------------
    def init_weights(self):
        """tools/train.py:162,171 calls this on student and teacher: reference-style random init, then the backbone's
        `init_cfg=dict(type='Pretrained', checkpoint=...)` / `pretrained=` checkpoint (mmcv BaseModule.init_weights ->
        load_checkpoint(strict=False) on the backbone, resnet.py:372-382 / resnet_rla.py:369-377)."""
        self.store.init_reference_style(0)
        ck = getattr(self.backbone, 'pretrained_checkpoint', None)
        if ck:
            load_backbone_checkpoint(self, ck)

    def _apply(self, fn, recurse=True):
        probe = fn(torch.zeros(1, device=self.store.device))
        if probe.device != self.store.device:
            self.store.to(probe.device)
            self._params = None
            self._engine = None
            self._anchor = None
        return self

    def state_dict(self, destination=None, prefix='', keep_vars=False):
        self.store.wait_pending()
        out = OrderedDict() if destination is None else destination
        for k, v in self.store.named_views().items():
            out[prefix + k] = v
        return out

    def load_state_dict(self, state_dict, strict=True):
        return self.store.load_named(state_dict, strict)

    def named_parameters(self, prefix='', recurse=True, remove_duplicate=True):
        if self._params is None:
            tv = self.store.named_views()
            gv = self.store.named_views(self.store.grad)
            self._params = OrderedDict()
            for k, v in tv.items():
                if not v.is_floating_point() or k.rsplit('.', 1)[-1] in ('running_mean', 'running_var'):
                    continue
                p = nn.Parameter(v, requires_grad=k in gv)
                if k in gv:
                    p.grad = gv[k]
                self._params[k] = p
        for k, p in self._params.items():
            yield prefix + ('.' if prefix else '') + k, p

    def parameters(self, recurse=True):
        for _, p in self.named_parameters():
            yield p

    def _rebind_grads(self):
        if self._params is not None:
            gv = self.store.named_views(self.store.grad)
            for k, p in self._params.items():
                if k in gv and p.grad is None:
                    p.grad = gv[k]

    # ---- hot path ----------------------------------------------------------------------------------
    def _get_engine(self):
        if self.store.device.type != 'cuda':
            raise RuntimeError('dsl_amd.FCOS runs only on an MI355X: move the model to cuda (no CPU fallback)')
        if self._engine is None:
            from .engine import Engine
            self._engine = Engine()
            self._anchor = torch.zeros(1, device=self.store.device, requires_grad=True)
        return self._engine

    def forward_train(self, img, img_metas, gt_bboxes, gt_labels, gt_bboxes_ignore=None, half_scale_copy=False):
        """single_stage.py:56-84 + base_dense_head.py:22-59.  Returns dict of scalar loss tensors.
        half_scale_copy: the batch has one more image than `img` holds - SemiEpochBasedRunner's scale-invariant half-size copy of
        img[-1] (semi_epoch_based_runner.py:186-204), whose metas / boxes are the lists' last entries; the stem kernel samples it
        from img[-1] (dsl_stem_pool_half), it is never written to memory."""
        eng = self._get_engine()
        N, _, H, W = img.shape
        N += 1 if half_scale_copy else 0
        assert len(img_metas) == N == len(gt_bboxes) == len(gt_labels)
        if self.comm_trace is not None and img.is_cuda:
            t0 = torch.cuda.Event(enable_timing=True)
            t0.record()
            self.comm_trace.append(dict(t0=t0, buckets=[]))
        plan = eng.plan(self.store, N, H, W, training=True)
        head = self.bbox_head
        if head.loss_weight != 1.0 and gt_bboxes_ignore is None:
            raise TypeError('loss_weight != 1.0 needs gt_bboxes_ignore (fcos_head.py:223 iterates ig_labels)')
        lp = plan.lossplan
        sw = head.effective_soft_weight(N)
        ws = self.world_size
        lp.configure(loss_weight=head.loss_weight, soft_weight=sw, grad_scale=self.loss_scale / ws, inv_world=1.0 / ws)
        pipe = self.pipeline_prefix and plan.prefix is not None
        if pipe:
            # frozen prefix of THIS step on its own stream: it waits for the previous backward's data-gradient chain only (named
            # event), not for that step's weight-gradient tail / optimizer step; layer1's output alternates between two buffers
            plan.set_parity(plan._parity ^ 1)
            plan.bind_image(img, half_last=half_scale_copy)
            if self._prefix_stream is None:
                self._prefix_stream = role_stream('prefix', self.store.device)
            if img.is_cuda:
                img.record_stream(self._prefix_stream)
            ready = getattr(img, '_dsl_ready', None)          # event of the image's producer (dsl_amd.data.mark_ready)
            if ready is not None:
                # one-shot: a loader that refills this tensor in place has to mark it again (a stale, long-fired event would let the
                # prefix start before the new contents are written)
                try:
                    del img._dsl_ready
                except AttributeError:
                    pass
            staged = plan._img_ref is None                     # bind_image copied the batch into plan.img on the CALLER's stream
            if getattr(self, '_prefix_plan', None) is not plan:
                # first use of this plan (or the first step at all): SLOT_TAIL was never recorded, so nothing orders the prefix
                # stream behind the caller's stream, which wrote the frozen packs (store.refresh) and possibly the image
                self._prefix_stream.wait_stream(torch.cuda.current_stream())
                self._prefix_plan = plan
            elif staged:
                # a CPU / non-fp32 / non-contiguous batch: the staging copy was queued on the caller's stream just now - whatever
                # the producer's event says, the prefix must not read plan.img before that copy has landed (round-3 advisor)
                ev = torch.cuda.Event()
                ev.record()
                self._prefix_stream.wait_event(ev)
            elif ready is not None:
                self._prefix_stream.wait_event(ready)
            elif img.is_cuda:
                # unknown producer: it can only have been queued on the caller's stream, behind the previous step - correct for
                # any caller, the prefix then simply runs after that step instead of under its tail
                ev = torch.cuda.Event()
                ev.record()
                self._prefix_stream.wait_event(ev)
            with torch.cuda.stream(self._prefix_stream):
                plan.prefix.run()
            fwd = plan.fwd_rest
        else:
            plan.bind_image(img, half_last=half_scale_copy)
            fwd = plan.fwd
        plan.fp8_warm(fwd)
        work = None
        if ws > 1:
            # target assignment first: the reduce_mean of (num_pos, sum centerness targets) - one 2-float all-reduce
            # (fcos_head.py:264-274) - then runs under the forward pass
            lp.set_targets(gt_bboxes, gt_labels, gt_bboxes_ignore)
            plan.assign_ops.run()
            work = None if self.comm_off else self._all_reduce_async(lp.stats[:2], after_current=True)
            fwd.run()
            if work is not None:
                work.wait()
        else:
            # one process: target upload + assignment go behind the forward pass on the caller's stream, into the time it
            # would otherwise spend waiting for the regression tower on the side stream
            fwd.run()
            lp.set_targets(gt_bboxes, gt_labels, gt_bboxes_ignore)
            plan.assign_ops.run()
        plan.loss_ops.run()
        eager_now = False
        if self.eager_backward and torch.is_grad_enabled():
            if self._in_train_step:       # train_step queues it behind its log-variable ops (they only need the loss kernel)
                self._deferred_plan = plan
                eager_now = True
            elif self.eager_backward == 'always':
                self._run_backward(plan)
                eager_now = True
        self._eager_now = eager_now
        out = _TrainStepFn.apply(self._anchor, self, plan)
        losses = _LossDict(loss_cls=out[0], loss_bbox=out[1], loss_centerness=out[2])
        if sw != 0.0:
            losses['loss_sisoft'] = out[3]
        losses.vec = out[:len(losses)]      # the same scalars as ONE tensor: lets _parse_losses avoid per-key device ops
        losses.vec_total = out              # ... and with their sum (the loss kernel's) as the last element: no device op at all
        return losses

    def _all_reduce_async(self, t, after_current=False):
        """Sum `t` over the ranks beside the caller's stream; returns an object whose wait() orders the current stream behind it.
        after_current: the collective must see everything queued on the current stream so far (torch's process group
        orders its own stream that way by itself)."""
        if self.rccl is None or not t.is_cuda:
            return dist.all_reduce(t, group=self.dist_group, async_op=True)
        from .parallel import StreamWork
        if self._comm_stream is None:
            self._comm_stream = role_stream('comm', self.store.device)
        cs = self._comm_stream
        if after_current:
            cs.wait_stream(torch.cuda.current_stream())
        t.record_stream(cs)
        self.rccl.all_reduce(t, cs)
        return StreamWork(cs)

    def _run_backward(self, plan):
        """Hand-written backward.  Data parallel: gradient bucket s (head+FPN, layer4, layer3, layer2) is all-reduced as
        soon as the side stream has finished its weight gradients - the communication stream waits for the named event
        the segment recorded (dsl_stream_wait_slot), not for the caller's stream, which is already running the next
        segment's data-gradient chain - so the bulk RCCL traffic overlaps the remaining backward (what torch DDP's bucket
        hooks do at mmdet/apis/train.py:92-96); only the last, smallest bucket (layer2, 5 MB) is exposed."""
        self._pending = []
        self._last_bwd_infos = [info for _, info in plan.bwd_segments]      # bucket ranges / event slots, for the optimizer
        proxy = self.comm_proxy if (self.comm_proxy and self.world_size == 1 and self.store.grad.is_cuda) else None
        ddp = (self.world_size > 1 and not self.comm_off) or proxy is not None
        on_gpu = self.store.grad.is_cuda
        if ddp and on_gpu and self._comm_stream is None:
            self._comm_stream = role_stream('comm', self.store.device)
        if proxy is not None and proxy.get('carrier', 'lib') == 'torch' and self._proxy_stream is None:
            self._proxy_stream = torch.cuda.Stream()
        self._partials_valid = False
        # Late exchange (round 6; set by an optimizer that updates bucket by bucket without a global norm, FlatSGD._sync_defer): the
        # collectives are queued BEHIND the whole backward pass, in the order the next forward pass needs the parameters (layer2's
        # bucket first, head + FPN last), and nothing waits for them at the end of the step - the per-bucket updates record named
        # events SLOT_UPD + k, the next step's forward list waits for them stage by stage (engine.Plan._fwd_resnet).  Queued
        # bucket by bucket behind each segment ("eager") the traffic shared the tail of the backward pass with the weight
        # gradients, the next step's frozen prefix and the step boundary: the one-GPU proxy priced that at 10 % of the step
        # whatever queue carried it, the late exchange on the weight-gradient queue at 5.7 % (profiles/r06_comm_queue_sweep.txt).
        late = bool(ddp and on_gpu and getattr(self, 'late_exchange', False) and self.clip_partials is None
                    and self.store.backbone != 'rla' and not getattr(plan, 'defer', False))
        nbc = [0]

        def exchange(info):
            lo, hi = info['bucket']
            if not on_gpu:            # host tensors (the gloo unit test of the bucket order): nothing to order against
                self._pending.append(dist.all_reduce(self.store.grad[lo:hi], group=self.dist_group, async_op=True))
                return
            from . import _lib as L
            from .parallel import StreamWork
            cs = self._proxy_stream if (proxy is not None and proxy.get('carrier', 'lib') == 'torch') else self._comm_stream
            csp = C_void(cs.cuda_stream)
            L.check(min(L.lib.dsl_stream_wait_slot(info['slot'], csp), 0), 'dsl_stream_wait_slot')
            if info['main']:
                cs.wait_stream(torch.cuda.current_stream())
            g = self.store.grad[lo:hi]
            if proxy is not None:
                # the bucket's stand-in collective (no peer: the values stay): same stream, same event, same optimizer hand-over
                L.check(L.lib.dsl_comm_proxy(L.ptr(g), (hi - lo) // 4 * 4, int(proxy.get('wgs', 32)), int(proxy.get('passes', 2)), csp),
                        'dsl_comm_proxy')
                self._pending.append(StreamWork(cs))
                nbc[0] += 1
                return
            with torch.cuda.stream(cs):
                if self.comm_trace:
                    es = torch.cuda.Event(enable_timing=True)
                    es.record()            # the bucket's named event has fired and the previous bucket's traffic is queued
                    self.comm_trace[-1]['buckets'].append(dict(mb=(hi - lo) * (2 if self.grad_bf16 else 4) / 1e6, start=es, done=None))
                work = None
                if self.grad_bf16:
                    # the bucket crosses xGMI as a bf16 copy (RCCL's own ring then adds in bf16 per hop; the fp32 master gradient
                    # gets the result back): cast -> all-reduce -> cast back, all on the communication stream
                    if self._g16 is None or self._g16.device != g.device:
                        self._g16 = torch.empty(self.store.grad.numel(), dtype=torch.bfloat16, device=g.device)
                    g16 = self._g16[lo:hi]
                    L.check(L.lib.dsl_cast_bf16(L.ptr(g), L.ptr(g16), hi - lo, csp), 'dsl_cast_bf16')
                    if self.rccl is not None:
                        L.check(L.lib.dsl_allreduce_bucket_bf16(self.rccl.comm, L.ptr(g16), hi - lo, csp), 'dsl_allreduce_bucket_bf16')
                    elif dist.get_backend(self.dist_group) == 'gloo':
                        # (the CPU-side test carrier has no bf16 sum: the bf16-rounded values are added in fp32 and rounded back)
                        t32 = g16.float()
                        dist.all_reduce(t32, group=self.dist_group, async_op=True).wait()
                        g16.copy_(t32)
                    else:
                        dist.all_reduce(g16, group=self.dist_group, async_op=True).wait()      # (orders cs behind the collective)
                    L.check(L.lib.dsl_cast_f32(L.ptr(g16), L.ptr(g), hi - lo, csp), 'dsl_cast_f32')
                elif self.rccl is not None:
                    self.rccl.all_reduce(g, cs)
                else:
                    work = dist.all_reduce(g, group=self.dist_group, async_op=True)
                if self.clip_partials is not None:
                    # the clipping norm of the REDUCED bucket, as soon as it has arrived: only the fold + the update stay behind the
                    # last bucket (FlatSGD.step)
                    if work is not None:
                        work.wait()
                        work = None
                    L.check(L.lib.dsl_sumsq_partial(L.ptr(g), hi - lo, C_void(self.clip_partials.data_ptr() + nbc[0] * L.SUMSQ_PARTS * 4), csp),
                            'dsl_sumsq_partial')
                self._pending.append(work if work is not None else StreamWork(cs))
            nbc[0] += 1

        todo = []
        for ol, info in plan.bwd_segments:
            ol.run()
            if not ddp or info['bucket'] is None:       # (bucket None: the deferred head update - its bucket completes with a later list)
                continue
            if late:
                todo.append(info)
            else:
                exchange(info)
        # late: nothing is queued here - the optimizer asks for each bucket's exchange right in front of that bucket's update
        # (exchange_late), so that on ONE hardware queue the order is exchange, update, exchange, update ... in the order the next
        # forward pass needs the parameters; queued all at once the first update sat behind every exchange (measured: 13 - 18 %)
        self._late_todo = todo[::-1] if late else []
        self._exchange_fn = exchange if late else None
        nb = nbc[0]
        self._partials_valid = bool(ddp and on_gpu and self.clip_partials is not None and 0 < nb * 256 <= self.clip_partials.numel())
        self._n_partials = nb * 256
        self._rebind_grads()

    def exchange_late(self):
        """Late exchange: queues the next bucket's collective (order: layer2, layer3, layer4, head + FPN) and returns (info, work), or
        None when every bucket of the last backward pass has been handed out."""
        todo = getattr(self, '_late_todo', None)
        if not todo:
            return None
        info = todo.pop(0)
        n = len(self._pending)
        self._exchange_fn(info)
        return info, self._pending[n]

    def wait_grads(self):
        while self.exchange_late() is not None:      # (a late exchange nobody asked for bucket by bucket: an optimizer with a global norm)
            pass
        for w in self._pending:
            w.wait()
        self._pending = []

    def forward(self, img, img_metas, return_loss=True, **kwargs):
        """detectors/base.py:155-173."""
        if return_loss:
            return self.forward_train(img, img_metas, **kwargs)
        return self.forward_test(img, img_metas, **kwargs)

    def forward_test(self, imgs, img_metas, **kwargs):
        if isinstance(imgs, (list, tuple)):
            assert len(imgs) == 1, 'test-time augmentation is out of scope'
            imgs, img_metas = imgs[0], img_metas[0]
        return self.simple_test(imgs, img_metas, **kwargs)

    def simple_test(self, img, img_metas, rescale=False):
        from .sweep import simple_test
        self.store.wait_pending()          # a deferred head update of the last optimizer step (the training forward waits in its op list)
        return simple_test(self, img, img_metas, rescale)

    def _parse_losses(self, losses):
        """detectors/base.py:175-208: total = sum of keys containing 'loss'; log vars averaged over ranks."""
        log_vars = OrderedDict()
        for name, value in losses.items():
            if isinstance(value, torch.Tensor):
                log_vars[name] = value if value.dim() == 0 else value.mean()     # the HIP head returns scalars
            else:
                log_vars[name] = sum(v.mean() for v in value)
        keys = list(log_vars.keys())
        vt = getattr(losses, 'vec_total', None)
        if vt is not None and vt.numel() == len(keys) + 1 and all('loss' in k for k in keys):
            # the HIP head: the loss kernel summed the terms itself, the total is the vector's last element
            loss = _TotalFn.apply(vt, self)
            log_vars['loss'] = loss
            keys.append('loss')
            # (the all-reduce below works in place: not on the storage `loss` is a view of)
            vec = vt.detach().clone() if (self.world_size > 1 and not self.comm_off) else vt.detach()
            return self._finish_log_vars(loss, vec, keys)
        # two device ops instead of one per key (stack + sum); the reference sums the keys containing 'loss'
        vec0 = getattr(losses, 'vec', None)
        stacked = vec0 if (vec0 is not None and vec0.numel() == len(keys)) else torch.stack([log_vars[k] for k in keys])
        if all('loss' in k for k in keys):
            loss = stacked.sum()
        else:
            loss = sum(v for k, v in log_vars.items() if 'loss' in k)
        log_vars['loss'] = loss
        keys.append('loss')
        vec = torch.cat([stacked.detach(), loss.detach().reshape(1)])
        return self._finish_log_vars(loss, vec, keys)

    def _finish_log_vars(self, loss, vec, keys):
        if self.world_size > 1 and not self.comm_off:      # ONE all-reduce for all log vars instead of one per key
            if self.rccl is not None and vec.is_cuda:
                self.rccl.all_reduce(vec)
            else:
                dist.all_reduce(vec, group=self.dist_group)
            vec = vec / self.world_size
        if self.lazy_log:
            log_vars = OrderedDict((k, vec[i]) for i, k in enumerate(keys))
        else:
            host = vec.tolist()
            log_vars = OrderedDict((k, host[i]) for i, k in enumerate(keys))
        return loss, log_vars

    def train_step(self, data, optimizer):
        """detectors/base.py:210-243."""
        self._in_train_step, self._deferred_plan = True, None
        try:
            losses = self(**data)
        finally:
            self._in_train_step = False
        loss, log_vars = self._parse_losses(losses)
        if self._deferred_plan is not None:
            # eager backward: its kernels go behind the handful of small device ops above instead of in front of them, where
            # those would sit between the last weight gradient and the optimizer step
            plan, self._deferred_plan = self._deferred_plan, None
            self._run_backward(plan)
        return dict(loss=loss, log_vars=log_vars, num_samples=len(data['img_metas']))

    def val_step(self, data, optimizer=None):
        return self.train_step(data, optimizer)
