"""CPU oracle for the FCOS R50-FPN teacher-student training step  (TEST INFRASTRUCTURE ONLY).

This file restates, in plain torch-on-CPU (fp32), the arithmetic of the reference hot path
(chenbinghui1/DSL @ /root/reference, an mmdetection-2.14 fork).  It is the *checker* for the HIP
kernels in dsl_amd/csrc: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import it.  The product (dsl_amd/) never imports it and fails loudly without its HIP library.

Pinning: tests/golden/*.npz were produced by running the reference's own python files in this
container (tests/golden/make_golden.py); tests/test_oracle_golden.py checks every function below
against them, plus the reference's own known-answer tests (GIoU triple, distance2bbox cases).
mmcv-full 1.3.10 is absent from /root/reference; ConvModule/Scale are restated as conv->norm->relu
and x*scalar (their documented behaviour), `mmcv.ops.sigmoid_focal_loss` through the in-tree
py_sigmoid_focal_loss formula, `mmcv.ops.nms` as offset-0 greedy NMS with IoU > thr suppression.

Every function cites the reference file:line it follows (paths relative to /root/reference).
`emulate_bf16=True` additionally rounds tensors to bf16 at the points where the HIP path stores
bf16 (weights, every activation written to HBM, every activation-gradient written to HBM); with
it off the oracle is the fp32 reference arithmetic.
"""
import math

import torch
import torch.nn.functional as F

INF = 1e8  # mmdet/models/dense_heads/fcos_head.py:11
STRIDES = (8, 16, 32, 64, 128)
REGRESS_RANGES = ((-1, 64), (64, 128), (128, 256), (256, 512), (512, INF))  # fcos_head.py:61-62
NUM_CLASSES = 80


# ----------------------------------------------------------------------------------------------
# bf16 storage emulation
# ----------------------------------------------------------------------------------------------
class _RoundBoth(torch.autograd.Function):
    """y = bf16(x) in forward, g = bf16(g) in backward (what a bf16 HBM round trip does)."""

    @staticmethod
    def forward(ctx, x):
        return x.bfloat16().float()

    @staticmethod
    def backward(ctx, g):
        return g.bfloat16().float()


class _RoundFwd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.bfloat16().float()

    @staticmethod
    def backward(ctx, g):
        return g


class _JitterRound(torch.autograd.Function):
    """x*(1 + eps*n), n ~ N(0,1), both ways, followed (rnd) by the bf16 round trip: a DIFFERENT realisation of the same
    storage rounding - what another fp32 summation order in front of the store does (an fp32-ulp-sized change flips the
    bf16 rounding of the ~1 % of values that sit next to a rounding boundary).  Without rnd it is the bare fp32-level
    perturbation: the sensitivity probe of tests/golden/make_trajectory.py."""

    @staticmethod
    def forward(ctx, x, eps, gen, rnd):
        ctx.eps, ctx.gen, ctx.rnd = eps, gen, rnd
        y = x * (1 + eps * torch.randn(x.shape, generator=gen))
        return y.bfloat16().float() if rnd else y

    @staticmethod
    def backward(ctx, g):
        y = g * (1 + ctx.eps * torch.randn(g.shape, generator=ctx.gen))
        return (y.bfloat16().float() if ctx.rnd else y), None, None, None


class Quant:
    def __init__(self, emulate_bf16=False, jitter=0.0, seed=0, fp32_grad_tags=()):
        self.on = emulate_bf16
        self.jitter = jitter
        self.gen = torch.Generator().manual_seed(seed) if jitter else None
        self.fp32_grad_tags = set(fp32_grad_tags)      # storage points whose GRADIENT tensor is kept in fp32

    def act(self, x, tag=None):      # activation stored in HBM as bf16, its gradient too
        if self.on and tag in self.fp32_grad_tags:
            return _RoundFwd.apply(x)
        if self.jitter:
            return _JitterRound.apply(x, self.jitter, self.gen, self.on)
        return _RoundBoth.apply(x) if self.on else x

    def wt(self, w):       # bf16 packed weight copy; gradient stays fp32 (wgrad writes fp32)
        return _RoundFwd.apply(w) if self.on else w


# ----------------------------------------------------------------------------------------------
# parameters: synthetic, key-addressed, so the reference model, the oracle and the HIP model can be
# given bit-identical weights without shipping a 128 MB checkpoint
# ----------------------------------------------------------------------------------------------
def r50_fcos_param_shapes(num_classes=NUM_CLASSES):
    """(name -> shape) for the 377 state_dict entries of FCOS R50-caffe + FPN + FCOSHead
    (mmdet/models/backbones/resnet.py:304-656, necks/fpn.py:61-148,
    dense_heads/anchor_free_head.py:89-139, fcos_head.py:112-116)."""
    sh = {}

    def bn(prefix, c):
        sh[prefix + '.weight'] = (c,)
        sh[prefix + '.bias'] = (c,)
        sh[prefix + '.running_mean'] = (c,)
        sh[prefix + '.running_var'] = (c,)
        sh[prefix + '.num_batches_tracked'] = ()

    sh['backbone.conv1.weight'] = (64, 3, 7, 7)
    bn('backbone.bn1', 64)
    inpl = 64
    for li, (planes, blocks) in enumerate(zip((64, 128, 256, 512), (3, 4, 6, 3))):
        for b in range(blocks):
            p = f'backbone.layer{li + 1}.{b}'
            sh[p + '.conv1.weight'] = (planes, inpl, 1, 1)
            bn(p + '.bn1', planes)
            sh[p + '.conv2.weight'] = (planes, planes, 3, 3)
            bn(p + '.bn2', planes)
            sh[p + '.conv3.weight'] = (planes * 4, planes, 1, 1)
            bn(p + '.bn3', planes * 4)
            if b == 0:
                sh[p + '.downsample.0.weight'] = (planes * 4, inpl, 1, 1)
                bn(p + '.downsample.1', planes * 4)
            inpl = planes * 4
    for i, c in enumerate((512, 1024, 2048)):
        sh[f'neck.lateral_convs.{i}.conv.weight'] = (256, c, 1, 1)
        sh[f'neck.lateral_convs.{i}.conv.bias'] = (256,)
    for i in range(5):
        sh[f'neck.fpn_convs.{i}.conv.weight'] = (256, 256, 3, 3)
        sh[f'neck.fpn_convs.{i}.conv.bias'] = (256,)
    for tower in ('cls_convs', 'reg_convs'):
        for i in range(4):
            sh[f'bbox_head.{tower}.{i}.conv.weight'] = (256, 256, 3, 3)
            sh[f'bbox_head.{tower}.{i}.conv.bias'] = (256,)
            sh[f'bbox_head.{tower}.{i}.gn.weight'] = (256,)
            sh[f'bbox_head.{tower}.{i}.gn.bias'] = (256,)
    for name, c in (('conv_cls', num_classes), ('conv_reg', 4), ('conv_centerness', 1)):
        sh[f'bbox_head.{name}.weight'] = (c, 256, 3, 3)
        sh[f'bbox_head.{name}.bias'] = (c,)
    for i in range(5):
        sh[f'bbox_head.scales.{i}.scale'] = ()
    return sh


def synth_state_dict(seed=0, num_classes=NUM_CLASSES):
    """Deterministic, well-conditioned synthetic weights addressed by key name.

    Conv weights ~ N(0, gain/fan_in) so activations neither vanish nor explode through 50 layers,
    BN/GN affine terms and running stats are non-trivial so that folding bugs show, conv_cls bias is
    the reference's focal prior -log((1-0.01)/0.01) (fcos_head.py:83-91)."""
    import zlib
    sd = {}
    for k, shape in r50_fcos_param_shapes(num_classes).items():
        g = torch.Generator().manual_seed((zlib.crc32(k.encode()) + 7919 * seed) & 0x7FFFFFFF)
        leaf = k.rsplit('.', 1)[-1]
        if leaf == 'num_batches_tracked':
            t = torch.tensor(0, dtype=torch.long)
        elif leaf == 'scale':
            t = torch.tensor(1.0 + 0.1 * torch.randn((), generator=g).item())
        elif leaf == 'running_mean':
            t = 0.1 * torch.randn(shape, generator=g)
        elif leaf == 'running_var':
            t = 0.5 + torch.rand(shape, generator=g)
        elif len(shape) == 4:
            fan_in = shape[1] * shape[2] * shape[3]
            gain = 2.0
            if '.conv3.' in k or 'downsample.0' in k:
                gain = 0.5      # keep the residual sum from doubling the variance each block
            if k.startswith('bbox_head.conv_'):
                gain = 0.05
            t = torch.randn(shape, generator=g) * math.sqrt(gain / fan_in)
        elif leaf == 'weight':      # bn / gn gamma
            t = 1.0 + 0.1 * torch.randn(shape, generator=g)
        elif leaf == 'bias':
            if k == 'bbox_head.conv_cls.bias':
                t = torch.full(shape, -math.log((1 - 0.01) / 0.01))
            elif k.startswith('bbox_head.conv_reg'):
                t = 1.0 + 0.05 * torch.randn(shape, generator=g)   # keep ReLU(bbox_pred) alive
            else:
                t = 0.05 * torch.randn(shape, generator=g)
        else:
            raise KeyError(k)
        sd[k] = t
    return sd


def trainable_keys(sd):
    """Parameters that receive gradients under the supervised config: BN frozen everywhere
    (norm_cfg requires_grad=False), stem + layer1 frozen (frozen_stages=1)
    (configs/fcos_semi/r50_caffe_mslonger_tricks_0.Xdata.py:4-15; resnet.py:612-628)."""
    out = []
    for k in sd:
        leaf = k.rsplit('.', 1)[-1]
        if leaf in ('running_mean', 'running_var', 'num_batches_tracked'):
            continue
        if k.startswith('backbone.'):
            if '.bn' in k or 'downsample.1' in k or k.startswith('backbone.bn1'):
                continue
            if k.startswith('backbone.conv1') or k.startswith('backbone.layer1.'):
                continue
        out.append(k)
    return out


# ----------------------------------------------------------------------------------------------
# a1  ResNet-50 caffe, frozen BN   (mmdet/models/backbones/resnet.py:262-301,598-645)
# ----------------------------------------------------------------------------------------------
def _bn_eval(sd, p, x):
    return F.batch_norm(x, sd[p + '.running_mean'], sd[p + '.running_var'], sd[p + '.weight'],
                        sd[p + '.bias'], training=False, eps=1e-5)


def resnet50_forward(sd, x, q):
    # stem: 7x7 s2 p3 conv -> BN(eval) -> ReLU -> maxpool 3x3 s2 p1   (resnet.py:598-610,635-638)
    x = F.conv2d(x, q.wt(sd['backbone.conv1.weight']), None, 2, 3)
    x = q.act(F.relu(_bn_eval(sd, 'backbone.bn1', x)))
    x = q.act(F.max_pool2d(x, 3, 2, 1))
    outs = []
    for li, blocks in enumerate((3, 4, 6, 3)):
        for b in range(blocks):
            p = f'backbone.layer{li + 1}.{b}'
            # caffe style: the stride sits on the first 1x1 (resnet.py:153-158)
            s = 2 if (b == 0 and li > 0) else 1
            idt = x
            o = F.conv2d(x, q.wt(sd[p + '.conv1.weight']), None, s, 0)
            o = q.act(F.relu(_bn_eval(sd, p + '.bn1', o)))
            o = F.conv2d(o, q.wt(sd[p + '.conv2.weight']), None, 1, 1)
            o = q.act(F.relu(_bn_eval(sd, p + '.bn2', o)))
            o = F.conv2d(o, q.wt(sd[p + '.conv3.weight']), None, 1, 0)
            o = _bn_eval(sd, p + '.bn3', o)
            if b == 0:      # res_layer.py:39-60
                idt = F.conv2d(x, q.wt(sd[p + '.downsample.0.weight']), None, s, 0)
                idt = q.act(_bn_eval(sd, p + '.downsample.1', idt))
            x = q.act(F.relu(o + idt))     # resnet.py:293-299
        outs.append(x)
    return outs


# ----------------------------------------------------------------------------------------------
# a2  FPN, start_level=1, add_extra_convs='on_output', relu_before_extra_convs  (necks/fpn.py:150-202)
# ----------------------------------------------------------------------------------------------
def fpn_forward(sd, feats, q):
    lat = [F.conv2d(feats[i + 1], q.wt(sd[f'neck.lateral_convs.{i}.conv.weight']),
                    sd[f'neck.lateral_convs.{i}.conv.bias']) for i in range(3)]
    lat[2] = q.act(lat[2])
    for i in (2, 1):     # fpn.py:163-172
        lat[i - 1] = q.act(lat[i - 1] + F.interpolate(lat[i], size=lat[i - 1].shape[2:],
                                                      mode='nearest'))
    outs = [q.act(F.conv2d(lat[i], q.wt(sd[f'neck.fpn_convs.{i}.conv.weight']),
                           sd[f'neck.fpn_convs.{i}.conv.bias'], 1, 1)) for i in range(3)]
    outs.append(q.act(F.conv2d(outs[-1], q.wt(sd['neck.fpn_convs.3.conv.weight']),
                               sd['neck.fpn_convs.3.conv.bias'], 2, 1)))          # P6, no relu
    outs.append(q.act(F.conv2d(F.relu(outs[-1]), q.wt(sd['neck.fpn_convs.4.conv.weight']),
                               sd['neck.fpn_convs.4.conv.bias'], 2, 1)))          # P7 on relu(P6)
    return outs


# ----------------------------------------------------------------------------------------------
# a3  FCOS head  (anchor_free_head.py:197-217, fcos_head.py:136-168)
# ----------------------------------------------------------------------------------------------
def head_forward(sd, feats, q, training=True):
    cls_scores, bbox_preds, ctrs = [], [], []
    for lvl, x in enumerate(feats):
        cf = rf = x
        for i in range(4):
            cf = F.conv2d(cf, q.wt(sd[f'bbox_head.cls_convs.{i}.conv.weight']),
                          sd[f'bbox_head.cls_convs.{i}.conv.bias'], 1, 1)
            cf = q.act(cf, 'tower_pre')      # the HIP path stores the pre-GN conv output as bf16
            cf = q.act(F.relu(F.group_norm(cf, 32, sd[f'bbox_head.cls_convs.{i}.gn.weight'],
                                           sd[f'bbox_head.cls_convs.{i}.gn.bias'], 1e-5)), 'tower_act')
            rf = F.conv2d(rf, q.wt(sd[f'bbox_head.reg_convs.{i}.conv.weight']),
                          sd[f'bbox_head.reg_convs.{i}.conv.bias'], 1, 1)
            rf = q.act(rf, 'tower_pre')
            rf = q.act(F.relu(F.group_norm(rf, 32, sd[f'bbox_head.reg_convs.{i}.gn.weight'],
                                           sd[f'bbox_head.reg_convs.{i}.gn.bias'], 1e-5)), 'tower_act')
        cls = F.conv2d(cf, q.wt(sd['bbox_head.conv_cls.weight']), sd['bbox_head.conv_cls.bias'], 1, 1)
        reg = F.conv2d(rf, q.wt(sd['bbox_head.conv_reg.weight']), sd['bbox_head.conv_reg.bias'], 1, 1)
        ctr = F.conv2d(rf, q.wt(sd['bbox_head.conv_centerness.weight']),
                       sd['bbox_head.conv_centerness.bias'], 1, 1)   # centerness_on_reg
        reg = F.relu(reg * sd[f'bbox_head.scales.{lvl}.scale'])       # norm_on_bbox (fcos_head.py:159-163)
        if not training:
            reg = reg * STRIDES[lvl]                                   # fcos_head.py:164-165
        cls_scores.append(cls)
        bbox_preds.append(reg)
        ctrs.append(ctr)
    return cls_scores, bbox_preds, ctrs


def extract_and_head(sd, img, q, training=True, debug=None):
    """single_stage.py:40-45 + base_dense_head.py:49.  `debug` (dict) receives the backbone / FPN
    feature maps with retain_grad() so tests can compare intermediate gradients."""
    x = q.act(img) if q.on else img
    c = resnet50_forward(sd, x, q)
    f = fpn_forward(sd, c, q)
    if debug is not None:
        for i, t in enumerate(c):
            if t.requires_grad:
                t.retain_grad()
            debug[f'c{i + 2}'] = t
        for i, t in enumerate(f):
            if t.requires_grad:
                t.retain_grad()
            debug[f'p{i + 3}'] = t
    return head_forward(sd, f, q, training)


# ----------------------------------------------------------------------------------------------
# a4  points   (anchor_free_head.py:287-321, fcos_head.py:550-560)
# ----------------------------------------------------------------------------------------------
def get_points(featmap_sizes, strides=STRIDES):
    pts = []
    for (h, w), s in zip(featmap_sizes, strides):
        ys, xs = torch.meshgrid(torch.arange(h, dtype=torch.float32),
                                torch.arange(w, dtype=torch.float32), indexing='ij')
        pts.append(torch.stack((xs.reshape(-1) * s, ys.reshape(-1) * s), -1) + s // 2)
    return pts


# ----------------------------------------------------------------------------------------------
# a5  target assignment for one image   (fcos_head.py:623-705), center_sampling, radius 1.5
# returns labels (P,), ltrb (P,4) un-normalised, argmin index (P,) (-1 where background)
# ----------------------------------------------------------------------------------------------
def assign_single(points_per_lvl, gt_bboxes, gt_labels, strides=STRIDES,
                  regress_ranges=REGRESS_RANGES, radius=1.5, num_classes=NUM_CLASSES):
    pts = torch.cat(points_per_lvl)
    P, G = pts.shape[0], gt_bboxes.shape[0]
    if G == 0:     # fcos_head.py:628-630
        return (torch.full((P,), num_classes, dtype=torch.long), torch.zeros(P, 4),
                torch.full((P,), -1, dtype=torch.long))
    lo = torch.cat([torch.full((p.shape[0],), float(r[0])) for p, r in zip(points_per_lvl, regress_ranges)])
    hi = torch.cat([torch.full((p.shape[0],), float(r[1])) for p, r in zip(points_per_lvl, regress_ranges)])
    rs = torch.cat([torch.full((p.shape[0],), float(s) * radius) for p, s in zip(points_per_lvl, strides)])
    x, y = pts[:, 0:1], pts[:, 1:2]
    x1, y1, x2, y2 = (gt_bboxes[:, i][None] for i in range(4))
    area = ((x2 - x1) * (y2 - y1)).repeat(P, 1)
    l, t, r, b = x - x1, y - y1, x2 - x, y2 - y
    ltrb = torch.stack((l, t, r, b), -1)
    cx, cy = (x1 + x2) / 2, (y1 + y2) / 2
    rr = rs[:, None]
    xmin, ymin, xmax, ymax = cx - rr, cy - rr, cx + rr, cy + rr
    c0 = torch.where(xmin > x1, xmin, x1.expand_as(xmin))     # fcos_head.py:669-676
    c1 = torch.where(ymin > y1, ymin, y1.expand_as(ymin))
    c2 = torch.where(xmax > x2, x2.expand_as(xmax), xmax)
    c3 = torch.where(ymax > y2, y2.expand_as(ymax), ymax)
    inside = torch.stack((x - c0, y - c1, c2 - x, c3 - y), -1).min(-1)[0] > 0
    mx = ltrb.max(-1)[0]
    in_range = (mx >= lo[:, None]) & (mx <= hi[:, None])
    area[~inside] = INF
    area[~in_range] = INF
    min_area, idx = area.min(dim=1)
    labels = gt_labels[idx].clone()
    labels[min_area == INF] = num_classes
    tgt = ltrb[torch.arange(P), idx]
    idx = idx.clone()
    idx[min_area == INF] = -1
    return labels, tgt, idx


def get_targets(points_per_lvl, gt_bboxes_list, gt_labels_list, strides=STRIDES, norm_on_bbox=True,
                **kw):
    """fcos_head.py:562-621.  Returns per-level lists, each concatenated over images (image-major),
    plus the per-level argmin indices (the 'assignment indices' of the parity bar)."""
    npl = [p.shape[0] for p in points_per_lvl]
    per_img = [assign_single(points_per_lvl, b, l, strides=strides, **kw)
               for b, l in zip(gt_bboxes_list, gt_labels_list)]
    labels, tgts, idxs = [], [], []
    for i in range(len(npl)):
        labels.append(torch.cat([r[0].split(npl)[i] for r in per_img]))
        t = torch.cat([r[1].split(npl)[i] for r in per_img])
        if norm_on_bbox:
            t = t / strides[i]      # fcos_head.py:618-619
        tgts.append(t)
        idxs.append(torch.cat([r[2].split(npl)[i] for r in per_img]))
    return labels, tgts, idxs


# ----------------------------------------------------------------------------------------------
# a6, a8, a9, a10  loss primitives
# ----------------------------------------------------------------------------------------------
def centerness_target(t):      # fcos_head.py:707-726
    lr, tb = t[:, [0, 2]], t[:, [1, 3]]
    if len(lr) == 0:
        return lr[..., 0]
    return torch.sqrt((lr.min(-1)[0] / lr.max(-1)[0]) * (tb.min(-1)[0] / tb.max(-1)[0]))


def distance2bbox(points, distance, max_shape=None):      # core/bbox/transforms.py:119-162
    x1 = points[..., 0] - distance[..., 0]
    y1 = points[..., 1] - distance[..., 1]
    x2 = points[..., 0] + distance[..., 2]
    y2 = points[..., 1] + distance[..., 3]
    bboxes = torch.stack([x1, y1, x2, y2], -1)
    if max_shape is not None:
        if not isinstance(max_shape, torch.Tensor):
            max_shape = x1.new_tensor(max_shape)
        max_shape = max_shape[..., :2].type_as(x1)
        if max_shape.ndim == 2:
            max_shape = max_shape.unsqueeze(1)
        min_xy = x1.new_tensor(0)
        max_xy = torch.cat([max_shape, max_shape], dim=-1).flip(-1)
        bboxes = torch.where(bboxes < min_xy, min_xy, bboxes)
        bboxes = torch.where(bboxes > max_xy, max_xy, bboxes)
    return bboxes


def giou_aligned(a, b, eps=1e-6):      # core/bbox/iou_calculators/iou2d_calculator.py:212-260
    area1 = (a[..., 2] - a[..., 0]) * (a[..., 3] - a[..., 1])
    area2 = (b[..., 2] - b[..., 0]) * (b[..., 3] - b[..., 1])
    lt = torch.max(a[..., :2], b[..., :2])
    rb = torch.min(a[..., 2:], b[..., 2:])
    wh = (rb - lt).clamp(min=0)
    overlap = wh[..., 0] * wh[..., 1]
    union = torch.max(area1 + area2 - overlap, a.new_tensor(eps))
    ious = overlap / union
    elt = torch.min(a[..., :2], b[..., :2])
    erb = torch.max(a[..., 2:], b[..., 2:])
    ewh = (erb - elt).clamp(min=0)
    earea = torch.max(ewh[..., 0] * ewh[..., 1], a.new_tensor(eps))
    return ious - (earea - union) / earea


def focal_loss_elem(pred, labels, num_classes=NUM_CLASSES, gamma=2.0, alpha=0.25):
    """losses/focal_loss.py:11-56 with the one-hot of :165-168 (label == C -> all-zero row)."""
    t = F.one_hot(labels, num_classes + 1)[:, :num_classes].type_as(pred)
    p = pred.sigmoid()
    pt = (1 - p) * t + p * (1 - t)
    fw = (alpha * t + (1 - alpha) * (1 - t)) * pt.pow(gamma)
    return F.binary_cross_entropy_with_logits(pred, t, reduction='none') * fw


# ----------------------------------------------------------------------------------------------
# a7  FCOSHead.loss   (fcos_head.py:170-338), including the DSL branches
# ----------------------------------------------------------------------------------------------
def fcos_loss(cls_scores, bbox_preds, centernesses, gt_bboxes, gt_labels, gt_bboxes_ignore=None,
              loss_weight=1.0, soft_weight=0.0, soft_scale=1.0, num_classes=NUM_CLASSES,
              strides=STRIDES, world_mean=lambda t: t, return_aux=False):
    """`soft_scale` is the warm-up factor the reference keeps as mutable state
    (1/1000 while cur_iter <= soft_warm_up, else 1; fcos_head.py:323-327) - the caller owns the
    counter.  `world_mean` is reduce_mean (core/utils/dist_utils.py:63-69)."""
    B = cls_scores[0].shape[0]
    sizes = [c.shape[-2:] for c in cls_scores]
    pts = get_points(sizes, strides)
    labels, tgts, idxs = get_targets(pts, gt_bboxes, gt_labels, strides=strides,
                                     num_classes=num_classes)
    ig_labels = None
    if gt_bboxes_ignore is not None:       # fcos_head.py:207-215
        ig_lab = [torch.full((b.shape[0],), num_classes - 1, dtype=torch.long) for b in gt_bboxes_ignore]
        ig_labels, _, _ = get_targets(pts, gt_bboxes_ignore, ig_lab, strides=strides,
                                      num_classes=num_classes)
    stream_w = None
    if loss_weight != 1.0:                 # fcos_head.py:217-234
        stream_w = []
        for lab in labels:
            w = torch.ones(lab.shape[0])
            n = lab.shape[0]
            cut = int(n / 2) if B % 2 == 0 else int(n / B * (B - 1) / 2)
            w[cut:] *= loss_weight
            stream_w.append(w)
        stream_w = torch.cat(stream_w)
    fc = torch.cat([c.permute(0, 2, 3, 1).reshape(-1, num_classes) for c in cls_scores])
    fb = torch.cat([b.permute(0, 2, 3, 1).reshape(-1, 4) for b in bbox_preds])
    fctr = torch.cat([c.permute(0, 2, 3, 1).reshape(-1) for c in centernesses])
    fl = torch.cat(labels)
    ft = torch.cat(tgts)
    fp = torch.cat([p.repeat(B, 1) for p in pts])
    pos = ((fl >= 0) & (fl < num_classes)).nonzero().reshape(-1)
    num_pos = max(float(world_mean(torch.tensor(float(len(pos))))), 1.0)
    pb, pc, pt_ = fb[pos], fctr[pos], ft[pos]
    ctr_t = centerness_target(pt_)
    denorm = max(float(world_mean(ctr_t.sum().detach())), 1e-6)
    if len(pos) > 0:
        pp = fp[pos]
        w = torch.ones_like(ctr_t)
        if stream_w is not None:
            w = w * stream_w[pos]
        wb = ctr_t * w
        if not torch.any(wb > 0):       # losses/iou_loss.py:345-348
            loss_bbox = (distance2bbox(pp, pb) * wb[:, None]).sum()
        else:
            g = giou_aligned(distance2bbox(pp, pb), distance2bbox(pp, pt_), 1e-6)
            loss_bbox = ((1 - g) * wb).sum() / denorm
        loss_ctr = (F.binary_cross_entropy_with_logits(pc, ctr_t, reduction='none') * w).sum() / num_pos
    else:
        loss_bbox, loss_ctr = pb.sum(), pc.sum()
    weight = torch.ones(fl.shape[0])
    if ig_labels is not None:           # fcos_head.py:297-304
        fig = torch.cat(ig_labels).clone()
        inter = ((fig - num_classes) * (fl - num_classes)).nonzero().reshape(-1)
        fig[inter] = num_classes
        weight = fig.float() - num_classes + 1
    if stream_w is not None:
        weight = weight * stream_w
    loss_cls = (focal_loss_elem(fc, fl, num_classes) * weight[:, None]).sum() / num_pos
    out = dict(loss_cls=loss_cls, loss_bbox=loss_bbox, loss_centerness=loss_ctr)
    if B % 2 != 0 and soft_weight != 0.0:      # fcos_head.py:312-328
        s = 0.0
        for i in range(1, len(cls_scores)):
            h, w_ = cls_scores[i].shape[-2:]
            d = cls_scores[i][B - 2] - cls_scores[i - 1][B - 1][:, :h, :w_]
            s = s + (d * d).mean()
        out['loss_sisoft'] = s * (soft_weight * soft_scale)
    if return_aux:
        return out, dict(labels=fl, bbox_targets=ft, assign_idx=torch.cat(idxs), cls_weight=weight,
                         num_pos=num_pos, ctr_denorm=denorm, ctr_targets=ctr_t, pos_inds=pos)
    return out


# ----------------------------------------------------------------------------------------------
# a13  scale-invariant third image   (runner/hooks/semi_epoch_based_runner.py:186-204)
# ----------------------------------------------------------------------------------------------
def append_half_scale(img, gt_bboxes, gt_labels, gt_bboxes_ignore):
    B, _, H, W = img.shape
    small = F.interpolate(img[B - 1:], size=(int(H / 2), int(W / 2)), mode='bilinear')
    canvas = torch.zeros_like(img[B - 1:])
    canvas[:, :, :small.shape[2], :small.shape[3]] = small
    img = torch.cat([img, canvas])
    gt_bboxes = list(gt_bboxes) + [gt_bboxes[-1] / 2]
    gt_labels = list(gt_labels) + [gt_labels[-1]]
    if gt_bboxes_ignore is not None:
        gt_bboxes_ignore = list(gt_bboxes_ignore) + [gt_bboxes_ignore[-1] / 2]
    return img, gt_bboxes, gt_labels, gt_bboxes_ignore


# ----------------------------------------------------------------------------------------------
# whole step: forward + loss + backward   (detectors/single_stage.py:56-84, base.py:175-243)
# ----------------------------------------------------------------------------------------------
def train_step(sd, img, gt_bboxes, gt_labels, gt_bboxes_ignore=None, emulate_bf16=False,
               want_grads=True, debug=None, quant=None, **loss_kw):
    q = quant if quant is not None else Quant(emulate_bf16)
    tk = trainable_keys(sd)
    p = {k: (v.detach().clone().requires_grad_(k in tk) if v.is_floating_point() else v)
         for k, v in sd.items()}
    cls, reg, ctr = extract_and_head(p, img, q, training=True, debug=debug)
    for t in cls + reg + ctr:
        t.retain_grad()
    losses = fcos_loss(cls, reg, ctr, gt_bboxes, gt_labels, gt_bboxes_ignore, **loss_kw)
    total = sum(v for k, v in losses.items() if 'loss' in k)     # base.py:197-198
    grads = {}
    if want_grads:
        total.backward()
        grads = {k: p[k].grad for k in tk}
    return ({k: float(v) for k, v in losses.items()}, grads,
            dict(cls=cls, reg=reg, ctr=ctr, total=float(total)))


# ----------------------------------------------------------------------------------------------
# a12  SGD with momentum / weight decay / param-group rules, grad-norm clip
#      (mmcv DefaultOptimizerConstructor semantics, SURVEY.md §8a note; torch.optim.SGD formula)
# ----------------------------------------------------------------------------------------------
def param_group_rule(key, base_lr, base_wd, bias_lr_mult=2.0, bias_decay_mult=0.0):
    leaf = key.rsplit('.', 1)[-1]
    is_norm = ('.gn.' in key) or ('.bn' in key) or ('downsample.1' in key)
    lr, wd = base_lr, base_wd
    if leaf == 'bias' and not is_norm:
        lr, wd = base_lr * bias_lr_mult, base_wd * bias_decay_mult
    return lr, wd


def clip_coef(grads, max_norm):      # torch.nn.utils.clip_grad_norm_
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads.values())).float()
    return min(float(max_norm / (total + 1e-6)), 1.0), float(total)


def sgd_step(params, grads, bufs, base_lr=0.01, momentum=0.9, base_wd=1e-4, max_norm=None,
             first_step=False):
    coef = 1.0
    if max_norm is not None:
        coef, _ = clip_coef(grads, max_norm)
    for k, g in grads.items():
        lr, wd = param_group_rule(k, base_lr, base_wd)
        d = g * coef + wd * params[k]
        bufs[k] = d.clone() if first_step else momentum * bufs[k] + d
        params[k] = params[k] - lr * bufs[k]
    return params, bufs


# ----------------------------------------------------------------------------------------------
# a14  EMA teacher   (runner/hooks/semi_epoch_based_runner.py:368-409): state_dict lerp, every tensor
# ----------------------------------------------------------------------------------------------
def ema_update(teacher_sd, student_sd, keep_rate=0.99):
    out = {}
    for k, v in teacher_sd.items():
        s = student_sd[k]
        nv = v.float() * keep_rate + s.float() * (1 - keep_rate)
        out[k] = nv.to(v.dtype)      # load_state_dict copies back into int64 num_batches_tracked
    return out


# ----------------------------------------------------------------------------------------------
# a15  teacher sweep: get_bboxes + multiclass_nms   (fcos_head.py:406-548, bbox_nms.py:7-94)
# ----------------------------------------------------------------------------------------------
def nms_greedy(boxes, scores, thr):
    """mmcv.ops.nms semantics (1.3.10): sort by score desc, suppress IoU > thr, offset 0."""
    order = scores.argsort(descending=True, stable=True)
    area = (boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])
    sup = torch.zeros(len(boxes), dtype=torch.bool)
    keep = []
    for i in order.tolist():
        if sup[i]:
            continue
        keep.append(i)
        lt = torch.max(boxes[i, :2], boxes[:, :2])
        rb = torch.min(boxes[i, 2:], boxes[:, 2:])
        wh = (rb - lt).clamp(min=0)
        inter = wh[:, 0] * wh[:, 1]
        sup |= inter / (area[i] + area - inter) > thr
    return torch.tensor(keep, dtype=torch.long)


def multiclass_nms(bboxes, scores, score_thr, iou_thr, max_num, score_factors):
    """bbox_nms.py:7-94: valid = score > thr BEFORE the centerness factor; class-offset trick."""
    C = scores.shape[1] - 1
    b = bboxes[:, None].expand(scores.shape[0], C, 4)
    s = scores[:, :-1]
    labels = torch.arange(C)[None].expand_as(s)
    b, s, labels = b.reshape(-1, 4), s.reshape(-1), labels.reshape(-1)
    valid = s > score_thr
    s = (scores[:, :-1] * score_factors[:, None]).reshape(-1)
    inds = valid.nonzero().squeeze(1)
    b, s, labels = b[inds], s[inds], labels[inds]
    if b.numel() == 0:
        return torch.zeros(0, 5), torch.zeros(0, dtype=torch.long)
    off = labels.to(b) * (b.max() + 1)
    keep = nms_greedy(b + off[:, None], s, iou_thr)
    keep = keep[:max_num] if max_num > 0 else keep
    return torch.cat([b[keep], s[keep, None]], -1), labels[keep]


def get_bboxes(cls_scores, bbox_preds, centernesses, img_shapes, scale_factors, nms_pre=1000,
               score_thr=0.05, iou_thr=0.5, max_per_img=100, rescale=True, strides=STRIDES):
    B = cls_scores[0].shape[0]
    pts = get_points([c.shape[-2:] for c in cls_scores], strides)
    mb, ms, mc = [], [], []
    for cls, reg, ctr, p in zip(cls_scores, bbox_preds, centernesses, pts):
        C = cls.shape[1]
        sc = cls.permute(0, 2, 3, 1).reshape(B, -1, C).sigmoid()
        ce = ctr.permute(0, 2, 3, 1).reshape(B, -1).sigmoid()
        bp = reg.permute(0, 2, 3, 1).reshape(B, -1, 4)
        p = p.expand(B, -1, 2)
        if 0 < nms_pre < bp.shape[1]:          # core/export get_k_for_topk
            mxs, _ = (sc * ce[..., None]).max(-1)
            _, ti = mxs.topk(nms_pre)
            bi = torch.arange(B).view(-1, 1).expand_as(ti)
            p, bp, sc, ce = p[bi, ti], bp[bi, ti], sc[bi, ti], ce[bi, ti]
        mb.append(distance2bbox(p, bp, max_shape=img_shapes))
        ms.append(sc)
        mc.append(ce)
    bb = torch.cat(mb, 1)
    if rescale:
        bb = bb / bb.new_tensor(scale_factors).unsqueeze(1)
    ss = torch.cat(ms, 1)
    ss = torch.cat([ss, ss.new_zeros(B, ss.shape[1], 1)], -1)
    cc = torch.cat(mc, 1)
    return [multiclass_nms(bb[i], ss[i], score_thr, iou_thr, max_per_img, cc[i]) for i in range(B)]


# ----------------------------------------------------------------------------------------------
# synthetic COCO-shaped batch   - shared by tests, smoke and bench
# ----------------------------------------------------------------------------------------------
def synth_boxes(rng, n, H=800, W=1333, lo=16.0, hi=600.0):
    import numpy as np
    cx = rng.uniform(0, W, n)
    cy = rng.uniform(0, H, n)
    w = np.exp(rng.uniform(math.log(lo), math.log(min(hi, W)), n))
    h = np.exp(rng.uniform(math.log(lo), math.log(min(hi, H)), n))
    b = np.stack([np.clip(cx - w / 2, 0, W), np.clip(cy - h / 2, 0, H),
                  np.clip(cx + w / 2, 0, W), np.clip(cy + h / 2, 0, H)], 1).astype('float32')
    keep = ((b[:, 2] - b[:, 0]) >= 1) & ((b[:, 3] - b[:, 1]) >= 1)
    return b[keep]
