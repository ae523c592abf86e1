"""Parameter storage for the FCOS R50-caffe + FPN + FCOSHead detector, laid out for the HIP kernels.

The reference keeps 377 separate OIHW fp32 tensors.  Here every trainable
parameter lives in ONE flat fp32 buffer (`train`), with conv weights stored KRSC =
[Cout][kh][kw][Cin] so that
  * the weight-gradient kernel writes straight into the matching flat `grad` buffer,
  * SGD / EMA / gradient all-reduce are single passes over flat memory,
  * the bf16 forward pack is an element-wise cast of the same buffer (`train16`).
`state_dict()` keys, shapes and dtypes are exactly the reference's: each entry is a (permuted) view
into the flat buffers, so checkpoints and the EMA hook interoperate.

Frozen tensors (stem, layer1, every BatchNorm) live in a second flat buffer (`frozen`).
"""
import math

import torch

from . import _lib as L

STAGE_BLOCKS = (3, 4, 6, 3)
STAGE_PLANES = (64, 128, 256, 512)


def _round_up(x, m):
    return (x + m - 1) // m * m


class ConvSpec:
    def __init__(self, name, cin, cout, k, stride, pad, bn=None, bias=False, trainable=True, cin_store=None, bn_train=False,
                 stage=None, no_dgrad=False):
        """cin_store: input channels of the STORED weight rows (>= cin; the extra columns are zero and stay zero: their
        inputs are zero padding) - RLA's conv1 reads cat(x, h) from one [C + 128]-wide buffer.  bn_train: the BatchNorm
        behind the conv runs in eval mode but its affine parameters train (RLA_ResNet).  stage: backward segment
        (0..3 = backbone stages) the gradients belong to."""
        self.name, self.cin, self.cout, self.k, self.stride, self.pad = name, cin, cout, k, stride, pad
        self.bn, self.bias, self.trainable = bn, bias, trainable
        self.cout_pad = _round_up(cout, 64)
        self.cin_store = cin_store or cin
        self.bn_train = bn_train and trainable
        self.stage, self.no_dgrad = stage, no_dgrad


def backbone_specs():
    """ResNet-50, caffe style, frozen_stages=1, BN frozen (mmdet/models/backbones/resnet.py:304-656)."""
    specs = [ConvSpec('backbone.conv1', 3, 64, 7, 2, 3, bn='backbone.bn1', trainable=False)]
    inpl = 64
    for li, (planes, blocks) in enumerate(zip(STAGE_PLANES, STAGE_BLOCKS)):
        tr = li >= 1
        for b in range(blocks):
            p = f'backbone.layer{li + 1}.{b}'
            s = 2 if (b == 0 and li > 0) else 1      # caffe: stride on the first 1x1 (resnet.py:153-158)
            specs.append(ConvSpec(p + '.conv1', inpl, planes, 1, s, 0, bn=p + '.bn1', trainable=tr))
            specs.append(ConvSpec(p + '.conv2', planes, planes, 3, 1, 1, bn=p + '.bn2', trainable=tr))
            specs.append(ConvSpec(p + '.conv3', planes, planes * 4, 1, 1, 0, bn=p + '.bn3', trainable=tr))
            if b == 0:
                specs.append(ConvSpec(p + '.downsample.0', inpl, planes * 4, 1, s, 0, bn=p + '.downsample.1',
                                      trainable=tr))
            inpl = planes * 4
    return specs


RLA_C = 32          # channels of the recurrent state h (resnet_rla.py: rla_channel)
RLA_PAD = 128       # cat(x, h) is stored [x | h | zeros] with C + RLA_PAD channels (multiple of the 128-channel wgrad tile)


def rla_backbone_specs():
    """RLA_ResNet (mmdet/models/backbones/resnet_rla.py:140-287), layers [3, 4, 6, 3], style 'pytorch' (the 3x3 strides),
    frozen_stages=1: stem, stage 0 and its RLA layers frozen; the BatchNorms of stages 1-3 keep trainable affine parameters
    (eval-mode statistics).  Names are the reference's module names."""
    specs = [ConvSpec('backbone.conv1', 3, 64, 7, 2, 3, bn='backbone.bn1', trainable=False)]
    inpl = 64
    for s_, (planes, blocks) in enumerate(zip(STAGE_PLANES, STAGE_BLOCKS)):
        tr = s_ >= 1
        for b in range(blocks):
            p = f'backbone.stages.{s_}.{b}'
            st = 2 if (b == 0 and s_ > 0) else 1
            first_trainable = tr and s_ == 1 and b == 0        # fed by the frozen stage 0: no data gradient needed
            specs.append(ConvSpec(p + '.conv1', inpl + RLA_C, planes, 1, 1, 0, bn=p + '.bn1', trainable=tr, bn_train=True,
                                  cin_store=inpl + RLA_PAD, stage=s_, no_dgrad=first_trainable))
            specs.append(ConvSpec(p + '.conv2', planes, planes, 3, st, 1, bn=p + '.bn2', trainable=tr, bn_train=True, stage=s_))
            specs.append(ConvSpec(p + '.conv3', planes, planes * 4, 1, 1, 0, bn=p + '.bn3', trainable=tr, bn_train=True, stage=s_))
            if b == 0:
                specs.append(ConvSpec(p + '.downsample.0', inpl, planes * 4, 1, st, 0, bn=p + '.downsample.1', trainable=tr,
                                      bn_train=True, stage=s_, no_dgrad=first_trainable))
            inpl = planes * 4
        # the stage's shared RLA convolutions: conv_out 1x1 (4*planes -> 32), recurrent 3x3 (32 -> 32, stored 64 -> 32 over a
        # zero-padded source)
        specs.append(ConvSpec(f'backbone.conv_outs.{s_}', planes * 4, RLA_C, 1, 1, 0, trainable=tr, stage=s_))
        # (stored 128 wide where it trains: the weight-gradient kernel tiles input channels by 128)
        specs.append(ConvSpec(f'backbone.recurrent_convs.{s_}', RLA_C, RLA_C, 3, 1, 1, trainable=tr, cin_store=128 if tr else 64,
                              stage=s_))
    return specs


def rla_stage_bns():
    """(name, trainable) of the per-block BatchNorm(32) layers on the recurrent path (stage_bns, resnet_rla.py:236,283);
    stage_bns.3.2 is frozen by _freeze_stages (:364-366)."""
    return [(f'backbone.stage_bns.{s_}.{b}', s_ >= 1 and not (s_ == 3 and b == 2))
            for s_, blocks in enumerate(STAGE_BLOCKS) for b in range(blocks)]


def neck_specs():
    specs = [ConvSpec(f'neck.lateral_convs.{i}.conv', c, 256, 1, 1, 0, bias=True)
             for i, c in enumerate((512, 1024, 2048))]
    specs += [ConvSpec(f'neck.fpn_convs.{i}.conv', 256, 256, 3, 1 if i < 3 else 2, 1, bias=True) for i in range(5)]
    return specs


def head_specs():
    specs = []
    for tower in ('cls_convs', 'reg_convs'):
        specs += [ConvSpec(f'bbox_head.{tower}.{i}.conv', 256, 256, 3, 1, 1, bias=True) for i in range(4)]
    return specs


class ParamStore:
    """Flat buffers + named views.  One instance for the student, one for the EMA teacher."""

    def __init__(self, num_classes=80, device='cpu', backbone='resnet'):
        assert num_classes == 80 and backbone in ('resnet', 'rla')
        self.num_classes, self.backbone = num_classes, backbone
        self.defer_head = False       # deferred head update (FlatSGD._sync_defer decides; engine.Plan.defer reads it)
        self._pending_ev = None
        bspecs = backbone_specs() if backbone == 'resnet' else rla_backbone_specs()
        self.convs = {s.name: s for s in bspecs + neck_specs() + head_specs()}
        self.extra_bns = rla_stage_bns() if backbone == 'rla' else []        # BatchNorms that follow no convolution
        self.train_regions = {}     # name -> (offset, numel, shape)   (shape = storage shape)
        self.frozen_regions = {}
        toff = foff = 0

        def add(regions, off, name, shape):
            n = _round_up(int(math.prod(shape)) if len(shape) else 1, 8)
            regions[name] = (off, n, tuple(shape))
            return off + n

        # BatchNorms with TRAINABLE affine parameters (eval-mode statistics; RLA_ResNet): every gamma in one block, every
        # beta in the next, at the START of the flat buffer - i.e. inside the gradient bucket that completes last - and the
        # statistics in the same order in the frozen buffer: the per-step fold to (scale, bias) is one launch
        self.bn_train = [(s.bn, s.cout) for s in self.convs.values() if s.bn and s.bn_train] + \
                        [(n, RLA_C) for n, tr in self.extra_bns if tr]
        self.bn_train_off = {}
        if self.bn_train:
            ctot = 0
            for n, c in self.bn_train:
                self.bn_train_off[n] = (ctot, c)
                ctot += c
            self.bn_train_ch = ctot
            toff = add(self.train_regions, toff, 'bn_train.weight', (ctot,))
            toff = add(self.train_regions, toff, 'bn_train.bias', (ctot,))
            foff = add(self.frozen_regions, foff, 'bn_train.running_mean', (ctot,))
            foff = add(self.frozen_regions, foff, 'bn_train.running_var', (ctot,))
        for s in self.convs.values():
            shape = (s.cout, s.k, s.k, s.cin_store)
            if s.trainable:
                toff = add(self.train_regions, toff, s.name + '.weight', shape)
                if s.bias:
                    toff = add(self.train_regions, toff, s.name + '.bias', (s.cout,))
            else:
                foff = add(self.frozen_regions, foff, s.name + '.weight', shape)
            if s.bn and not s.bn_train:
                for leaf in ('weight', 'bias', 'running_mean', 'running_var'):
                    foff = add(self.frozen_regions, foff, f'{s.bn}.{leaf}', (s.cout,))
        for n, tr in self.extra_bns:
            if not tr:
                for leaf in ('weight', 'bias', 'running_mean', 'running_var'):
                    foff = add(self.frozen_regions, foff, f'{n}.{leaf}', (RLA_C,))
        # the conv kernels fetch whole weight tiles: cout_pad rows, of which a 32-channel conv stores 32.  Inside the buffers
        # the extra rows are whatever follows (their outputs are dropped); the LAST conv of the frozen buffer would read past
        # the allocation (RLA's recurrent_convs.0 did, when the next page happened to be unmapped): zero rows behind it
        over = max([(s.cout_pad - s.cout) * s.k * s.k * s.cin_store for s in self.convs.values() if not s.trainable] + [0])
        if over:
            foff = add(self.frozen_regions, foff, '_tail_pad', (over,))
        for tower in ('cls_convs', 'reg_convs'):
            for i in range(4):
                toff = add(self.train_regions, toff, f'bbox_head.{tower}.{i}.gn.weight', (256,))
                toff = add(self.train_regions, toff, f'bbox_head.{tower}.{i}.gn.bias', (256,))
        # predictors: rows padded to the kernel tile; conv_reg (4) + conv_centerness (1) share one region
        toff = add(self.train_regions, toff, 'head.cls_w', (128, 3, 3, 256))
        toff = add(self.train_regions, toff, 'head.cls_b', (128,))
        toff = add(self.train_regions, toff, 'head.regctr_w', (64, 3, 3, 256))
        toff = add(self.train_regions, toff, 'head.regctr_b', (64,))
        toff = add(self.train_regions, toff, 'head.scales', (8,))
        self.n_train, self.n_frozen = toff, foff
        self.device = torch.device(device)
        self.train = torch.zeros(toff, dtype=torch.float32, device=device)
        self.frozen = torch.zeros(foff, dtype=torch.float32, device=device)
        self.grad = torch.zeros(toff, dtype=torch.float32, device=device)
        self.nbt = {s.bn: torch.zeros((), dtype=torch.long, device=device) for s in self.convs.values() if s.bn}
        self.nbt.update({n: torch.zeros((), dtype=torch.long, device=device) for n, _ in self.extra_bns})
        # bias group mask for the optimizer's paramwise rules (bias_lr_mult / bias_decay_mult apply to
        # conv biases, not to norm layers: mmcv DefaultOptimizerConstructor, SURVEY.md §8a note)
        grp = torch.zeros(toff, dtype=torch.uint8)
        for name, (off, n, shape) in self.train_regions.items():
            if (name.endswith('.conv.bias') or name in ('head.cls_b', 'head.regctr_b')):
                grp[off:off + n] = 1
        self.group = grp.to(device)
        # derived device-side packs (built by refresh())
        self.train16 = None
        self.frozen16 = None
        self.wT16 = None
        self.bn_scale = None
        self.bn_bias = None
        self.stem16 = None
        self._views = None
        self.dirty = True

    # -- flat <-> named ---------------------------------------------------------------------------
    def tview(self, name, buf=None):
        off, n, shape = self.train_regions[name]
        buf = self.train if buf is None else buf
        return buf[off:off + int(math.prod(shape))].view(shape)

    def fview(self, name):
        off, n, shape = self.frozen_regions[name]
        return self.frozen[off:off + int(math.prod(shape))].view(shape)

    def toff(self, name):
        return self.train_regions[name][0]

    def named_views(self, buf=None):
        """OrderedDict key -> tensor view with the reference's names/shapes (OIHW for conv weights).
        `buf` selects which flat buffer the trainable views come from (train / grad)."""
        out = {}
        tb = self.train if buf is None else buf

        def conv_w(s):
            v = self.tview(s.name + '.weight', tb) if s.trainable else (self.fview(s.name + '.weight') if buf is None else None)
            if v is None:
                return None
            if s.cin_store != s.cin:      # stored rows carry zero columns behind the real input channels
                v = v[..., :s.cin]
            return v.permute(0, 3, 1, 2)

        def bn(name):
            if name in self.bn_train_off:          # trainable affine parameters: slices of the gamma / beta blocks
                o, c = self.bn_train_off[name]
                out[f'{name}.weight'] = self.tview('bn_train.weight', tb)[o:o + c]
                out[f'{name}.bias'] = self.tview('bn_train.bias', tb)[o:o + c]
                if buf is None:
                    out[f'{name}.running_mean'] = self.fview('bn_train.running_mean')[o:o + c]
                    out[f'{name}.running_var'] = self.fview('bn_train.running_var')[o:o + c]
                    out[f'{name}.num_batches_tracked'] = self.nbt[name]
                return
            if buf is not None:
                return
            for leaf in ('weight', 'bias', 'running_mean', 'running_var'):
                out[f'{name}.{leaf}'] = self.fview(f'{name}.{leaf}')
            out[f'{name}.num_batches_tracked'] = self.nbt[name]

        for s in self.convs.values():
            w = conv_w(s)
            if w is not None:
                out[s.name + '.weight'] = w
            if s.bias:
                out[s.name + '.bias'] = self.tview(s.name + '.bias', tb)
            if s.bn:
                bn(s.bn)
            if s.name.startswith('bbox_head.') and s.name.endswith('.conv'):
                base = s.name[:-5]
                out[base + '.gn.weight'] = self.tview(base + '.gn.weight', tb)
                out[base + '.gn.bias'] = self.tview(base + '.gn.bias', tb)
        for n, _ in self.extra_bns:
            bn(n)
        cw, cb = self.tview('head.cls_w', tb), self.tview('head.cls_b', tb)
        rw, rb = self.tview('head.regctr_w', tb), self.tview('head.regctr_b', tb)
        out['bbox_head.conv_cls.weight'] = cw[:80].permute(0, 3, 1, 2)
        out['bbox_head.conv_cls.bias'] = cb[:80]
        out['bbox_head.conv_reg.weight'] = rw[:4].permute(0, 3, 1, 2)
        out['bbox_head.conv_reg.bias'] = rb[:4]
        out['bbox_head.conv_centerness.weight'] = rw[4:5].permute(0, 3, 1, 2)
        out['bbox_head.conv_centerness.bias'] = rb[4:5]
        sc = self.tview('head.scales', tb)
        for i in range(5):
            out[f'bbox_head.scales.{i}.scale'] = sc[i]
        return out

    def wait_pending(self):
        """Deferred head update (FlatSGD, Plan.defer): the last optimizer step may still be updating the head + FPN bucket on the
        optimizer's stream.  Training forwards wait for it inside their op lists; every other reader or writer of the parameters
        on the current stream (state_dict, EMA, evaluation, loading) calls this first."""
        ev = getattr(self, '_pending_ev', None)
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)
            self._pending_ev = None

    def load_named(self, sd, strict=True):
        self.wait_pending()
        views = self.named_views()
        missing = [k for k in views if k not in sd]
        unexpected = [k for k in sd if k not in views]
        if strict and (missing or unexpected):
            raise KeyError(f'state_dict mismatch: missing {missing[:5]}, unexpected {unexpected[:5]}')
        with torch.no_grad():
            for k, v in views.items():
                if k in sd:
                    v.copy_(sd[k].to(v.device))
        self.dirty = True
        return missing, unexpected

    def to(self, device):
        device = torch.device(device)
        for a in ('train', 'frozen', 'grad', 'group'):
            setattr(self, a, getattr(self, a).to(device))
        self.nbt = {k: v.to(device) for k, v in self.nbt.items()}
        self.device = device
        self.dirty = True
        return self

    # -- device-side derived packs ------------------------------------------------------------------
    def wT_layout(self):
        """Offsets (bf16 elements) of the dgrad packs [Cin][kh][kw][CoutPad] of every conv that needs a
        data gradient."""
        if getattr(self, '_wT', None) is None:
            off, lay = 0, {}
            for s in self.convs.values():
                if not s.trainable or s.no_dgrad:
                    continue
                if s.name in ('backbone.layer2.0.conv1', 'backbone.layer2.0.downsample.0'):
                    continue      # fed by frozen layer1: no data gradient needed
                lay[s.name] = (off, s.cin_store * s.k * s.k * s.cout_pad)
                off += _round_up(lay[s.name][1], 8)
                if s.k == 3 and s.stride == 2 and self.backbone == 'rla' and s.name.startswith('backbone.'):
                    # (RLA_ResNet's stage-entry 3x3 / 2 convolutions; the FPN's P6 / P7 convolutions keep the general strided gather)
                    # + the four parity-class packs of its data gradient (ops.dgrad_s2_descs): [Cin][k_c * k_c][CoutPad], k_c = 1 for
                    # the even-even pixels, 2 for the rest
                    from .ops import s2_class
                    for py in (0, 1):
                        for px in (0, 1):
                            k_c = s2_class(py, px)[0]
                            lay[f'{s.name}#s2{py}{px}'] = (off, s.cin_store * k_c * k_c * s.cout_pad)
                            off += _round_up(lay[f'{s.name}#s2{py}{px}'][1], 8)
            lay['head.cls'] = (off, 256 * 9 * 128)
            off += 256 * 9 * 128
            lay['head.regctr'] = (off, 256 * 9 * 64)
            off += 256 * 9 * 64
            self._wT, self._wT_total = lay, off
        return self._wT, self._wT_total

    def refresh_frozen(self):
        """Fold the frozen BatchNorms into per-channel (scale, bias) and build the bf16 packs of the frozen
        convs.  y = gamma*(x-mean)/sqrt(var+eps)+beta = x*scale + bias  (BN eval mode, resnet.py:647-656)."""
        dev = self.device
        scs, bis, self.bn_off = [], [], {}
        off = 0
        frozen_bns = [(s.bn, s.cout) for s in self.convs.values() if s.bn and not s.bn_train] + \
                     [(n, RLA_C) for n, tr in self.extra_bns if not tr]
        for name, c in frozen_bns:
            g, b = self.fview(name + '.weight'), self.fview(name + '.bias')
            m, v = self.fview(name + '.running_mean'), self.fview(name + '.running_var')
            sc = g / torch.sqrt(v + 1e-5)
            scs.append(sc)
            bis.append(b - m * sc)
            self.bn_off[name] = off
            off += c
        # trainable-affine BatchNorms: the tail of the same arrays, re-folded after every optimizer step (refold_bn)
        self.bn_train_base = off
        if self.bn_train:
            for name, (o, c) in self.bn_train_off.items():
                self.bn_off[name] = off + o
            g, b = self.tview('bn_train.weight'), self.tview('bn_train.bias')
            m, v = self.fview('bn_train.running_mean'), self.fview('bn_train.running_var')
            sc = g / torch.sqrt(v + 1e-5)
            scs.append(sc)
            bis.append(b - m * sc)
        wp = torch.zeros(64, 7 * 64, device=dev)
        w = self.fview('backbone.conv1.weight')           # [64][7][7][3] -> [64][448], k = tap*8 + c
        wp[:, :392] = torch.cat([w, torch.zeros(64, 7, 7, 5, device=dev)], -1).reshape(64, 392)
        # the same weight for dsl_stem_pool: [22 groups][64][8], group g = the (kx, c) values 8 (g % 3) .. + 8 of tap row g / 3
        wg = torch.zeros(64, 7, 24, device=dev)
        wg[:, :, :21] = w.reshape(64, 7, 21)
        wg = torch.cat([wg.reshape(64, 21, 8).permute(1, 0, 2), torch.zeros(1, 64, 8, device=dev)], 0)
        # Execution plans bake the device addresses of these four tensors into their kernel descriptors: refresh IN
        # PLACE whenever the buffers already exist on this device (load_state_dict into a model that has already run)
        for name, val in (('bn_scale', torch.cat(scs)), ('bn_bias', torch.cat(bis)), ('frozen16', self.frozen.bfloat16()),
                          ('stem16', wp.bfloat16()), ('stem_groups16', wg.bfloat16())):
            cur = getattr(self, name, None)
            if cur is not None and cur.device == val.device and cur.shape == val.shape:
                cur.copy_(val)
            else:
                setattr(self, name, val.contiguous())
                self._pack_tab = None        # holds pointers into bn_scale
                self.generation = getattr(self, 'generation', 0) + 1    # plans built against older buffers are stale

    def refold_bn(self, sp=None):
        """Eval-mode BatchNorms whose affine parameters train (RLA_ResNet): scale = gamma / sqrt(var + eps), bias = beta -
        mean * scale, re-made on the device after every optimizer step (one launch over all their channels)."""
        if not self.bn_train:
            return
        o = self.bn_train_base * 4
        L.check(L.lib.dsl_bn_fold(L.ptr(self.tview('bn_train.weight')), L.ptr(self.tview('bn_train.bias')),
                                  L.ptr(self.fview('bn_train.running_mean')), L.ptr(self.fview('bn_train.running_var')), 1e-5,
                                  self.bn_scale.data_ptr() + o, self.bn_bias.data_ptr() + o, self.bn_train_ch,
                                  sp or L.stream_ptr()), 'dsl_bn_fold')

    def refresh_train_packs(self, stream_ptr=None):
        """bf16 forward pack (= cast of the flat buffer) and dgrad packs.  Called after load_state_dict;
        the fused SGD kernel keeps train16 current afterwards, the dgrad packs are re-made per step."""
        if self.train16 is None or self.train16.device != self.device:
            self.train16 = torch.empty(self.n_train, dtype=torch.bfloat16, device=self.device)
        sp = stream_ptr or L.stream_ptr()
        L.check(L.lib.dsl_cast_bf16(L.ptr(self.train), L.ptr(self.train16), self.n_train, sp), 'dsl_cast_bf16')
        self.repack_dgrad(sp)

    def repack_dgrad(self, sp=None, side=False):
        """All dgrad packs in ONE launch (dsl_pack_dgrad_batched) from a device-resident item table.
        side=True: the launch goes to the library's side stream behind everything queued on `sp` so far and marks named
        event SLOT_PACKS; only the backward pass reads these packs, it waits for that event (Engine: first backward op),
        and the next forward pass starts without them."""
        import ctypes as C
        lay, total = self.wT_layout()
        if self.wT16 is None or self.wT16.device != self.device:
            self.wT16 = torch.zeros(total, dtype=torch.bfloat16, device=self.device)
            self._pack_tab = self._pack_ops = None
        sp = sp or L.stream_ptr()
        self.refold_bn(sp)          # the packs fold the BatchNorm scale: keep it current first
        if getattr(self, '_pack_tab', None) is None:
            items, start = [], 0
            for name, (off, n) in lay.items():
                if name == 'head.cls':
                    w, co, cop, taps, cin, sc = self.tview('head.cls_w'), 80, 128, 9, 256, None
                elif name == 'head.regctr':
                    w, co, cop, taps, cin, sc = self.tview('head.regctr_w'), 5, 64, 9, 256, None
                else:
                    s = self.convs[name.split('#')[0]]
                    w, co, cop, taps, cin = self.tview(s.name + '.weight'), s.cout, s.cout_pad, s.k * s.k, s.cin_store
                    sc = self.bn_scale[self.bn_off[s.bn]:] if s.bn else None
                tapmap = 0
                if '#s2' in name:         # a parity-class pack: k_c * k_c out taps selected from the 9 source taps
                    from .ops import s2_class
                    k_c, _, srcs = s2_class(int(name[-2]), int(name[-1]))
                    tapmap = 9 << 16
                    for t, src in enumerate(srcs):
                        tapmap |= (src if src >= 0 else 0xF) << (4 * t)
                    taps = k_c * k_c
                it = L.PackItem()
                it.w, it.scale, it.out = w.data_ptr(), (sc.data_ptr() if sc is not None else 0), self.wT16.data_ptr() + off * 2
                it.cout, it.cout_pad, it.taps, it.cin = co, cop, taps, cin
                it.tapmap = tapmap
                it.tiles_ci, it.tiles_co, it.block_start = (cin + 63) // 64, (cop + 63) // 64, start     # 64x64 tiles (optim.hip)
                start += it.tiles_ci * it.tiles_co * taps
                items.append(it)
            arr = (L.PackItem * len(items))(*items)
            host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).clone()
            self._pack_tab = (host.to(self.device), len(items), start)
        tab, n, blocks = self._pack_tab
        if side:
            ops_ = getattr(self, '_pack_ops', None)
            if ops_ is None:
                ops_ = (L.Op * 3)()
                ops_[0].kind, ops_[0].i[0], ops_[0].i[1] = L.OP_FORK, 1, 0
                ops_[1].kind, ops_[1].p[0], ops_[1].i[0], ops_[1].i[1], ops_[1].i[6] = L.OP_PACK_DGRAD, tab.data_ptr(), n, blocks, 1
                ops_[2].kind, ops_[2].i[0], ops_[2].i[1] = L.OP_RECORD, 1, L.SLOT_PACKS
                self._pack_ops = ops_
            L.check(L.lib.dsl_run_ops(ops_, 3, sp), 'dsl_run_ops(pack_dgrad)')
        else:
            L.check(L.lib.dsl_pack_dgrad_batched(L.ptr(tab), n, blocks, sp), 'dsl_pack_dgrad_batched')

    def refresh(self):
        assert self.device.type == 'cuda', 'the HIP packs live on the GPU'
        self.wait_pending()
        self.refresh_frozen()
        self.refresh_train_packs()
        self.dirty = False

    def grad_buckets(self):
        """Contiguous gradient ranges in the order the backward completes them: head+FPN, last backbone stage, ..., first
        trainable stage (with the trainable BatchNorm blocks in front of it).  Together they cover the whole flat gradient
        buffer exactly once."""
        r = self.train_regions
        if self.backbone == 'resnet':
            first = [r[f'backbone.layer{i}.0.conv1.weight'][0] for i in (2, 3, 4)]
        else:
            first = [0] + [r[f'backbone.stages.{i}.0.conv1.weight'][0] for i in (2, 3)]
        b = first + [r['neck.lateral_convs.0.conv.weight'][0], self.n_train]
        return [(b[3], b[4]), (b[2], b[3]), (b[1], b[2]), (b[0], b[1])]

    # -- pointers used by the plans -----------------------------------------------------------------
    def w16_ptr(self, s):
        if s.trainable:
            return self.train16.data_ptr() + self.toff(s.name + '.weight') * 2
        if s.name == 'backbone.conv1':
            return self.stem16.data_ptr()
        return self.frozen16.data_ptr() + self.frozen_regions[s.name + '.weight'][0] * 2

    def t16_ptr(self, region):
        return self.train16.data_ptr() + self.toff(region) * 2

    def t32_ptr(self, region, buf=None):
        return (self.train if buf is None else buf).data_ptr() + self.toff(region) * 4

    def wT_ptr(self, name):
        lay, _ = self.wT_layout()
        return self.wT16.data_ptr() + lay[name][0] * 2

    def bn_ptrs(self, bn):
        o = self.bn_off[bn] * 4
        return self.bn_scale.data_ptr() + o, self.bn_bias.data_ptr() + o

    # -- reference-style initialisation (random-init weights for the benchmark) ---------------------
    def init_reference_style(self, seed=0):
        """Kaiming-normal backbone convs, Xavier-uniform FPN, Normal(0, 0.01) head with the focal prior
        bias on conv_cls (fcos_head.py:83-91), BN gamma 1 / beta 0 / running stats (0, 1), scales 1."""
        g = torch.Generator().manual_seed(seed)
        sd = {}
        for k, v in self.named_views().items():
            leaf = k.rsplit('.', 1)[-1]
            shape = tuple(v.shape)
            if leaf == 'num_batches_tracked':
                t = torch.zeros((), dtype=torch.long)
            elif leaf == 'scale':
                t = torch.tensor(1.0)
            elif leaf == 'running_mean':
                t = torch.zeros(shape)
            elif leaf == 'running_var':
                t = torch.ones(shape)
            elif len(shape) == 4:
                fan_in = shape[1] * shape[2] * shape[3]
                fan_out = shape[0] * shape[2] * shape[3]
                if k.startswith('backbone.'):
                    t = torch.randn(shape, generator=g) * math.sqrt(2.0 / fan_out)
                elif k.startswith('neck.'):
                    a = math.sqrt(6.0 / (fan_in + fan_out))
                    t = (torch.rand(shape, generator=g) * 2 - 1) * a
                else:
                    t = torch.randn(shape, generator=g) * 0.01
            elif leaf == 'weight':
                t = torch.ones(shape)
                if self.backbone == 'rla' and k.endswith('.bn3.weight'):
                    t = torch.zeros(shape)          # zero_init_last_bn (resnet_rla.py:251-256)
            else:
                t = torch.zeros(shape)
                if k == 'bbox_head.conv_cls.bias':
                    t = torch.full(shape, -math.log((1 - 0.01) / 0.01))
            sd[k] = t
        self.load_named(sd)
        return self
