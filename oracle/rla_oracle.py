"""CPU oracle of the RLA_ResNet backbone - TEST INFRASTRUCTURE, like fcos_oracle.py: only
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it; the product path (dsl_amd/) never does.

Restates mmdet/models/backbones/resnet_rla.py of the reference in plain torch:
  RLA_Bottleneck.forward :103-136   x <- cat(x, h); conv1-bn1-relu, conv2(stride)-bn2-relu, conv3-bn3; identity =
                                    downsample(x_in); h <- AvgPool2d(2,2)(h) when the block strides; out = relu(. + identity);
                                    the returned y aliases out (in-place add and ReLU): y == out
  RLA_ResNet._forward_impl :289-327 stem, then per block: x, y, h = block(x, h); h = recurrent_conv(tanh(bn_i(h + conv_out(y))))
  _freeze_stages / train :343-388   stem + stage 0 (blocks, stage_bns[0], conv_outs[0], recurrent_convs[0]) and
                                    stage_bns[3][2] have requires_grad False; every BatchNorm runs in eval mode (norm_eval),
                                    the affine parameters of the other BatchNorms stay TRAINABLE
Pinned: tests/golden/rla_tiny.npz holds the reference's own stage outputs and parameter gradients on a seeded input
(tests/golden/make_golden.py rla), tests/test_oracle_golden.py compares.
"""
import math
import zlib

import torch
import torch.nn.functional as F

from . import fcos_oracle as O

LAYERS = (3, 4, 6, 3)
PLANES = (64, 128, 256, 512)
RLA_C = 32


def rla_param_shapes(num_classes=O.NUM_CLASSES):
    """name -> shape of FCOS with the RLA_ResNet backbone (neck / head entries as in r50_fcos_param_shapes)."""
    sh = {}

    def bn(prefix, c):
        for leaf in ('weight', 'bias', 'running_mean', 'running_var'):
            sh[f'{prefix}.{leaf}'] = (c,)
        sh[prefix + '.num_batches_tracked'] = ()

    sh['backbone.conv1.weight'] = (64, 3, 7, 7)
    bn('backbone.bn1', 64)
    inpl = 64
    for s, (planes, blocks) in enumerate(zip(PLANES, LAYERS)):
        sh[f'backbone.conv_outs.{s}.weight'] = (RLA_C, planes * 4, 1, 1)
        sh[f'backbone.recurrent_convs.{s}.weight'] = (RLA_C, RLA_C, 3, 3)
        for b in range(blocks):
            p = f'backbone.stages.{s}.{b}'
            sh[p + '.conv1.weight'] = (planes, inpl + RLA_C, 1, 1)
            bn(p + '.bn1', planes)
            sh[p + '.conv2.weight'] = (planes, planes, 3, 3)
            bn(p + '.bn2', planes)
            sh[p + '.conv3.weight'] = (planes * 4, planes, 1, 1)
            bn(p + '.bn3', planes * 4)
            if b == 0:
                sh[p + '.downsample.0.weight'] = (planes * 4, inpl, 1, 1)
                bn(p + '.downsample.1', planes * 4)
            inpl = planes * 4
            bn(f'backbone.stage_bns.{s}.{b}', RLA_C)
    for k, v in O.r50_fcos_param_shapes(num_classes).items():
        if not k.startswith('backbone.'):
            sh[k] = v
    return sh


def synth_state_dict(seed=0, num_classes=O.NUM_CLASSES):
    """Deterministic synthetic weights by key name (same recipe as fcos_oracle.synth_state_dict; bn3 gammas are NOT zero so
    that the residual branches carry signal - the zero_init_last_bn start is covered by its own test)."""
    sd = {}
    base = O.synth_state_dict(seed, num_classes)
    for k, shape in rla_param_shapes(num_classes).items():
        if not k.startswith('backbone.'):
            sd[k] = base[k]
            continue
        g = torch.Generator().manual_seed((zlib.crc32(k.encode()) + 7919 * seed) & 0x7FFFFFFF)
        leaf = k.rsplit('.', 1)[-1]
        if leaf == 'num_batches_tracked':
            t = torch.tensor(0, dtype=torch.long)
        elif leaf == 'running_mean':
            t = 0.1 * torch.randn(shape, generator=g)
        elif leaf == 'running_var':
            t = 0.5 + torch.rand(shape, generator=g)
        elif len(shape) == 4:
            fan_in = shape[1] * shape[2] * shape[3]
            gain = 2.0
            if '.conv3.' in k or 'downsample.0' in k:
                gain = 0.5
            if 'conv_outs' in k or 'recurrent_convs' in k:
                gain = 1.0
            t = torch.randn(shape, generator=g) * math.sqrt(gain / fan_in)
        elif leaf == 'weight':
            t = 1.0 + 0.1 * torch.randn(shape, generator=g)
        else:
            t = 0.05 * torch.randn(shape, generator=g)
        sd[k] = t
    return sd


def trainable_keys(sd):
    """requires_grad after RLA_ResNet._freeze_stages (frozen_stages=1): everything except the stem, stage 0 and its RLA
    layers, stage_bns.3.2, and buffers; neck / head as in fcos_oracle.trainable_keys."""
    out = []
    for k, v in sd.items():
        leaf = k.rsplit('.', 1)[-1]
        if leaf in ('running_mean', 'running_var', 'num_batches_tracked') or not v.is_floating_point():
            continue
        if k.startswith('backbone.'):
            if k.startswith(('backbone.conv1.', 'backbone.bn1.', 'backbone.stages.0.', 'backbone.stage_bns.0.',
                             'backbone.conv_outs.0.', 'backbone.recurrent_convs.0.', 'backbone.stage_bns.3.2.')):
                continue
        out.append(k)
    return out


def _bn(sd, p, x):
    return F.batch_norm(x, sd[p + '.running_mean'], sd[p + '.running_var'], sd[p + '.weight'], sd[p + '.bias'],
                        training=False, eps=1e-5)


def rla_resnet_forward(sd, x, q):
    """Returns the four stage outputs (resnet_rla.py:304-327; `outs.append(x)`, h is not part of the outputs).
    bf16 storage points (q.act) are the buffers the HIP path keeps: a1, a2, out, u, t, h."""
    x = F.conv2d(x, q.wt(sd['backbone.conv1.weight']), None, 2, 3)
    x = q.act(F.relu(_bn(sd, 'backbone.bn1', x)))
    x = q.act(F.max_pool2d(x, 3, 2, 1))
    h = torch.zeros(x.shape[0], RLA_C, x.shape[2], x.shape[3])
    outs = []
    for s, (planes, blocks) in enumerate(zip(PLANES, LAYERS)):
        for b in range(blocks):
            p = f'backbone.stages.{s}.{b}'
            stride = 2 if (b == 0 and s > 0) else 1            # style 'pytorch': the 3x3 strides (:84-86)
            identity = x
            o = F.conv2d(torch.cat((x, h), 1), q.wt(sd[p + '.conv1.weight']))
            o = q.act(F.relu(_bn(sd, p + '.bn1', o)))
            o = F.conv2d(o, q.wt(sd[p + '.conv2.weight']), None, stride, 1)
            o = q.act(F.relu(_bn(sd, p + '.bn2', o)))
            o = F.conv2d(o, q.wt(sd[p + '.conv3.weight']))
            o = _bn(sd, p + '.bn3', o)
            if b == 0:
                identity = F.conv2d(x, q.wt(sd[p + '.downsample.0.weight']), None, stride, 0)
                identity = q.act(_bn(sd, p + '.downsample.1', identity))
                if stride != 1:
                    h = q.act(F.avg_pool2d(h, 2, 2))
            x = q.act(F.relu(o + identity))
            # `y = out` at :123 ALIASES the tensor that `out += identity` (:132) and the in-place ReLU (:133, nn.ReLU(inplace=
            # True) at :91) then modify: what the block returns as y - and what conv_out sees - is the block OUTPUT, after
            # the residual add and the ReLU.  The oracle restates what the reference computes.
            y = x
            last = s == 3 and b == blocks - 1                   # h after the last block is never used (:322-327)
            if not last:
                u = q.act(h + F.conv2d(y, q.wt(sd[f'backbone.conv_outs.{s}.weight'])))
                t = q.act(torch.tanh(_bn(sd, f'backbone.stage_bns.{s}.{b}', u)))
                h = q.act(F.conv2d(t, q.wt(sd[f'backbone.recurrent_convs.{s}.weight']), None, 1, 1))
        outs.append(x)
    return outs


def extract_and_head(sd, img, q, training=True):
    feats = rla_resnet_forward(sd, img, q)
    fpn = O.fpn_forward(sd, feats, q)
    return O.head_forward(sd, fpn, q, training)


def train_step(sd, img, gt_bboxes, gt_labels, gt_bboxes_ignore=None, emulate_bf16=False, want_grads=True, quant=None,
               **loss_kw):
    """fcos_oracle.train_step with the RLA backbone."""
    q = quant if quant is not None else O.Quant(emulate_bf16)
    tk = trainable_keys(sd)
    p = {k: (v.detach().clone().requires_grad_(k in tk) if v.is_floating_point() else v) for k, v in sd.items()}
    cls, reg, ctr = extract_and_head(p, img, q, training=True)
    for t in cls + reg + ctr:
        t.retain_grad()
    losses = O.fcos_loss(cls, reg, ctr, gt_bboxes, gt_labels, gt_bboxes_ignore, **loss_kw)
    total = sum(v for k, v in losses.items() if 'loss' in k)
    grads = {}
    if want_grads:
        total.backward()
        grads = {k: p[k].grad for k in tk}
    return ({k: float(v.detach()) for k, v in losses.items()}, grads, dict(cls=cls, reg=reg, ctr=ctr, total=float(total.detach())))
