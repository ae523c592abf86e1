"""Import harness for the *reference* hot-path modules (container-only tooling).

The reference (chenbinghui1/DSL, an mmdet-2.14 fork at /root/reference) cannot be imported as a
package here: `mmdet/__init__.py` needs mmcv-full 1.3.10, torchvision, cv2, pycocotools, none of
which are installed.  Its hot-path arithmetic is plain torch, however, so this harness
  1. installs a minimal stand-in for the handful of mmcv symbols those files touch (thin wrappers
     over torch.nn), and
  2. pre-seeds `sys.modules` with empty package objects whose `__path__` points at the real
     reference directories, then imports the leaf files *unmodified* from /root/reference.

It exists only so that `make_golden.py` can run the reference on CPU and dump input/output
vectors into tests/golden/*.npz.  Nothing here travels to the GPU box as a dependency of the
product or of the tests (the tests read the .npz files only).  It is NOT a copy of mmcv or mmdet.
"""
import importlib
import sys
import types

import torch
import torch.nn as nn

REF = '/root/reference'


class _Registry:
    def __init__(self, name, parent=None, **kw):
        self.name = name
        self._m = {} if parent is None else parent._m

    def register_module(self, name=None, force=False, module=None):
        def deco(cls):
            self._m[name or cls.__name__] = cls
            return cls
        return deco

    def get(self, key):
        return self._m.get(key)

    def build(self, cfg, default_args=None):
        cfg = dict(cfg)
        for k, v in (default_args or {}).items():
            cfg.setdefault(k, v)
        return self._m[cfg.pop('type')](**cfg)


def _build_from_cfg(cfg, registry, default_args=None):
    return registry.build(cfg, default_args)


class _BaseModule(nn.Module):
    def __init__(self, init_cfg=None):
        super().__init__()
        self.init_cfg = init_cfg

    def init_weights(self):
        pass


class _Sequential(_BaseModule, nn.Sequential):
    def __init__(self, *args, init_cfg=None):
        _BaseModule.__init__(self, init_cfg)
        nn.Sequential.__init__(self, *args)


def _identity_deco_factory(*a, **k):
    def deco(f):
        return f
    return deco


def _build_conv_layer(cfg, *args, **kwargs):
    assert cfg is None or cfg.get('type', 'Conv2d') in ('Conv2d', 'Conv')
    return nn.Conv2d(*args, **kwargs)


def _build_norm_layer(cfg, num_features, postfix=''):
    cfg = dict(cfg)
    t = cfg.pop('type')
    rg = cfg.pop('requires_grad', True)
    cfg.setdefault('eps', 1e-5)
    if t == 'BN':
        name, layer = 'bn', nn.BatchNorm2d(num_features, **cfg)
    elif t == 'GN':
        name, layer = 'gn', nn.GroupNorm(num_channels=num_features, **cfg)
    else:
        raise NotImplementedError(t)
    for p in layer.parameters():
        p.requires_grad = rg
    return name + str(postfix), layer


class _ConvModule(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1,
                 groups=1, bias='auto', conv_cfg=None, norm_cfg=None, act_cfg=dict(type='ReLU'),
                 inplace=True, **kw):
        super().__init__()
        self.with_norm = norm_cfg is not None
        self.with_activation = act_cfg is not None
        if bias == 'auto':
            bias = not self.with_norm
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, stride=stride,
                              padding=padding, dilation=dilation, groups=groups, bias=bias)
        if self.with_norm:
            self.norm_name, norm = _build_norm_layer(norm_cfg, out_channels)
            self.add_module(self.norm_name, norm)
        if self.with_activation:
            assert act_cfg['type'] == 'ReLU'
            self.activate = nn.ReLU(inplace=inplace)

    def forward(self, x):
        x = self.conv(x)
        if self.with_norm:
            x = getattr(self, self.norm_name)(x)
        if self.with_activation:
            x = self.activate(x)
        return x


class _Scale(nn.Module):
    def __init__(self, scale=1.0):
        super().__init__()
        self.scale = nn.Parameter(torch.tensor(scale, dtype=torch.float))

    def forward(self, x):
        return x * self.scale


def _greedy_nms(boxes, scores, thr):
    order = scores.argsort(descending=True)
    keep = []
    b = boxes
    area = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    sup = torch.zeros(len(b), dtype=torch.bool)
    for i in order.tolist():
        if sup[i]:
            continue
        keep.append(i)
        lt = torch.max(b[i, :2], b[:, :2])
        rb = torch.min(b[i, 2:], b[:, 2:])
        wh = (rb - lt).clamp(min=0)
        inter = wh[:, 0] * wh[:, 1]
        iou = inter / (area[i] + area - inter)
        sup |= iou > thr
    return torch.tensor(keep, dtype=torch.long)


def _nms(boxes, scores, iou_threshold, offset=0, score_threshold=0, max_num=-1):
    keep = _greedy_nms(boxes, scores, iou_threshold)
    return torch.cat([boxes[keep], scores[keep, None]], -1), keep


def _batched_nms(boxes, scores, idxs, nms_cfg, class_agnostic=False):
    nms_cfg = dict(nms_cfg)
    thr = nms_cfg.get('iou_threshold', nms_cfg.get('iou_thr'))
    if boxes.numel() == 0:
        return torch.cat([boxes, scores[:, None]], -1), torch.zeros(0, dtype=torch.long)
    max_coordinate = boxes.max()
    offsets = idxs.to(boxes) * (max_coordinate + torch.tensor(1).to(boxes))
    keep = _greedy_nms(boxes + offsets[:, None], scores, thr)
    return torch.cat([boxes[keep], scores[keep, None]], -1), keep


class AttrDict(dict):
    __getattr__ = dict.get


def install():
    if 'mmcv' in sys.modules:
        return
    mmcv = types.ModuleType('mmcv')
    mmcv.__version__ = '1.3.10'
    mmcv.jit = lambda *a, **k: (lambda f: f)
    utils = types.ModuleType('mmcv.utils')
    utils.Registry = _Registry
    utils.build_from_cfg = _build_from_cfg
    cnn = types.ModuleType('mmcv.cnn')
    cnn.MODELS = _Registry('model')
    cnn.build_conv_layer = _build_conv_layer
    cnn.build_norm_layer = _build_norm_layer
    cnn.build_plugin_layer = None
    cnn.ConvModule = _ConvModule
    cnn.Scale = _Scale
    runner = types.ModuleType('mmcv.runner')
    runner.BaseModule = _BaseModule
    runner.Sequential = _Sequential
    runner.force_fp32 = _identity_deco_factory
    runner.auto_fp16 = _identity_deco_factory
    runner.load_checkpoint = runner.load_state_dict = None
    runner.OptimizerHook = type('OptimizerHook', (), {})
    ops = types.ModuleType('mmcv.ops')
    ops.sigmoid_focal_loss = None
    ops_nms = types.ModuleType('mmcv.ops.nms')
    ops_nms.batched_nms = _batched_nms
    ops_nms.nms = _nms
    ops.nms = ops_nms
    mmcv.utils, mmcv.cnn, mmcv.runner, mmcv.ops = utils, cnn, runner, ops
    for m in (mmcv, utils, cnn, runner, ops, ops_nms):
        sys.modules[m.__name__] = m

    def pkg(name):
        m = types.ModuleType(name)
        m.__path__ = [REF + '/' + name.replace('.', '/')]
        sys.modules[name] = m
        return m
    for n in ['mmdet', 'mmdet.core', 'mmdet.core.bbox', 'mmdet.core.bbox.iou_calculators',
              'mmdet.core.utils', 'mmdet.core.post_processing', 'mmdet.core.mask',
              'mmdet.core.export', 'mmdet.core.visualization', 'mmdet.utils', 'mmdet.models',
              'mmdet.models.dense_heads', 'mmdet.models.losses', 'mmdet.models.detectors',
              'mmdet.models.backbones', 'mmdet.models.necks', 'mmdet.models.utils']:
        pkg(n)
    ms = types.ModuleType('mmdet.core.mask.structures')
    ms.BitmapMasks = type('BitmapMasks', (), {})
    ms.PolygonMasks = type('PolygonMasks', (), {})
    sys.modules[ms.__name__] = ms
    sys.modules['mmdet.core.visualization'].imshow_det_bboxes = None
    sys.modules['mmdet.core.export'].get_k_for_topk = lambda k, size: k if 0 < k < size else -1
    sys.modules['mmdet.utils'].get_root_logger = lambda *a, **k: None

    imp = importlib.import_module
    core = sys.modules['mmdet.core']
    tr = imp('mmdet.core.bbox.transforms')
    iou = imp('mmdet.core.bbox.iou_calculators.iou2d_calculator')
    sys.modules['mmdet.core.bbox.iou_calculators'].bbox_overlaps = iou.bbox_overlaps
    sys.modules['mmdet.core.bbox'].bbox_overlaps = iou.bbox_overlaps
    misc = imp('mmdet.core.utils.misc')
    du = imp('mmdet.core.utils.dist_utils')
    core.distance2bbox, core.bbox2result = tr.distance2bbox, tr.bbox2result
    core.bbox_mapping_back = tr.bbox_mapping_back
    core.merge_aug_proposals = None
    core.multi_apply, core.reduce_mean = misc.multi_apply, du.reduce_mean
    core.bbox_overlaps = iou.bbox_overlaps
    nms = imp('mmdet.core.post_processing.bbox_nms')
    core.multiclass_nms = nms.multiclass_nms
    b = imp('mmdet.models.builder')
    models = sys.modules['mmdet.models']
    for k in dir(b):
        if not k.startswith('_'):
            setattr(models, k, getattr(b, k))
    rl = imp('mmdet.models.utils.res_layer')
    sys.modules['mmdet.models.utils'].ResLayer = rl.ResLayer
    for leaf in ['mmdet.models.losses.utils', 'mmdet.models.losses.focal_loss',
                 'mmdet.models.losses.iou_loss', 'mmdet.models.losses.cross_entropy_loss',
                 'mmdet.models.backbones.resnet', 'mmdet.models.necks.fpn',
                 'mmdet.models.dense_heads.base_dense_head',
                 'mmdet.models.dense_heads.dense_test_mixins',
                 'mmdet.models.dense_heads.anchor_free_head',
                 'mmdet.models.dense_heads.fcos_head', 'mmdet.models.detectors.base',
                 'mmdet.models.detectors.single_stage', 'mmdet.models.detectors.fcos']:
        imp(leaf)
    return b


def load_cfg(path):
    g = {}
    exec(open(path).read(), g)
    return g


def build_fcos(cfg_path, **head_overrides):
    """Build the reference FCOS detector from a reference config file (model dict only)."""
    b = install() or sys.modules['mmdet.models.builder']
    cfg = load_cfg(cfg_path)
    model = dict(cfg['model'])
    model['backbone'] = dict(model['backbone'])
    model['backbone'].pop('init_cfg', None)
    model['bbox_head'] = dict(model['bbox_head'], **head_overrides)
    model['test_cfg'] = AttrDict({k: (AttrDict(v) if isinstance(v, dict) else v)
                                  for k, v in model['test_cfg'].items()})
    model['train_cfg'] = AttrDict(model['train_cfg'])
    return b.build_detector(model)
