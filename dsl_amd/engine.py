"""Execution plans for the FCOS R50-FPN step: for one input shape (N, H, W) the engine allocates every
activation / gradient buffer once and pre-builds flat lists of kernel descriptors (dsl_op[]) that
libdsl_hip.so replays with ONE C call per segment (dsl_run_ops) — no per-launch Python.

What the lists compute is the reference's
  SingleStageDetector.forward_train (mmdet/models/detectors/single_stage.py:56-84)
    = ResNet.forward (backbones/resnet.py:630-645) -> FPN.forward (necks/fpn.py:150-202)
      -> FCOSHead.forward (dense_heads/fcos_head.py:118-168) -> FCOSHead.loss (:170-338)
and its autograd backward, written out by hand (no autograd graph exists here).
"""
import ctypes as C
import os

import torch

from . import _lib as L
from . import ops
from .head_loss import FcosLossPlan
from .params import STAGE_BLOCKS, STAGE_PLANES
from .tuning import skip_items, tune, tune_int

DEFER_SLOTS = 144      # workgroup budget of the towers' weight-gradient group when the head update is deferred (FlatSGD(defer_head_update=True))

BF = torch.bfloat16


class OpList:
    def __init__(self):
        self.items, self.keep, self.arr = [], [], None

    def _add(self, kind, desc=None, i=(), p=(), l=()):
        o = L.Op()
        o.kind = kind
        for k, v in enumerate(i):
            o.i[k] = int(v)
        for k, v in enumerate(p):
            o.p[k] = v if isinstance(v, int) else (0 if v is None else v.data_ptr())
        for k, v in enumerate(l):
            o.l[k] = int(v)
        if desc is not None:
            o.desc = C.addressof(desc)
            self.keep.append(desc)
        self.items.append(o)
        self.arr = None
        # has the caller's stream queued work since side stream 1 last waited for it?  (a FORK costs the recording stream
        # ~6 us, tools/microbench/sync_cost.hip: consecutive side launches with nothing new on the caller's stream share one)
        if kind == L.OP_FORK and (o.i[0] or 1) == 1 and o.i[1] == 0:
            self._fork1_fresh = True
        elif o.i[6] == 0 and kind not in (L.OP_FORK, L.OP_JOIN, L.OP_RECORD, L.OP_WAIT, L.OP_PROF):
            self._fork1_fresh = False

    def _fork1(self):
        if not getattr(self, '_fork1_fresh', False):
            self._add(L.OP_FORK)

    # Step-level ablation by graph region (tools/step_ablation_regions.sh; results are WRONG, only the clock is read): the tuning key
    # `skip` lists region tags (fwd.l2 fwd.l3 fwd.l4 fwd.fpn fwd.head bwd.head bwd.fpn bwd.l4 bwd.l3 bwd.l2) whose convolution /
    # data-gradient launches are left out of the op lists
    tag = ''

    def conv(self, d, side=False):
        """side: False/0 = the caller's stream, True/1..3 = that side stream of the library."""
        if self.tag and self.tag in skip_items():
            self.keep.append(d)
            return
        self._add(L.OP_CONV, d, i=(0, 0, 0, 0, 0, 0, int(side)))

    def bneck(self, d):
        """A whole bottleneck's forward pass as one launch on the caller's stream (dsl_bottleneck_fwd)."""
        if self.tag and self.tag in skip_items():
            self.keep.append(d)
            return
        self._add(L.OP_BNECK, d)

    def fork(self, side=1, other=0):
        """Side stream `side` waits for everything queued so far on stream `other` (0 = the caller's)."""
        self._add(L.OP_FORK, i=(side, other))

    def wgrad(self, d, side=False):
        """side=True: on the library's side stream, after a FORK (the weight gradient only depends on tensors
        that are complete at this point and are never overwritten within the step)."""
        if side:
            self._fork1()
            self._add(L.OP_WGRAD, d, i=(0, 0, 0, 0, 0, 0, 1))
        else:
            self._add(L.OP_WGRAD, d)

    def wgrad_group(self, arr, side=False):
        """arr: ops.wgrad_group(...) array (same-geometry convolutions, one launch)."""
        if side:
            self._fork1()
        self._add(L.OP_WGRAD_GROUP, arr, i=(len(arr), 0, 0, 0, 0, 0, 1 if side else 0))

    def wgrad_multi(self, plan, side=False):
        """plan: ops.WgradMulti (sub-launches of one tile configuration as one grid)."""
        if side:
            self._fork1()
        self._add(L.OP_WGRAD_MULTI, i=(0, 0, 0, 0, 0, 0, 1 if side else 0), p=(plan.host.data_ptr(), plan.dev.data_ptr()))
        self.keep.append(plan)

    @staticmethod
    def _fbits(x):
        import struct
        return struct.unpack('<i', struct.pack('<f', float(x)))[0]

    def quant_fp8(self, x, y, rows, c, ld_x, scale=1.0, partials=None, side=False):
        """The fp8 copy a fp8 convolution reads: y = e4m3(x * scale) with the tensor's own scale 448 / max|x| when `partials` (a
        float workspace for the block maxima) is given, the static `scale` otherwise."""
        n_p = 0 if partials is None else partials.numel()
        self._add(L.OP_QUANT_FP8, i=(c, ld_x, n_p, 0, 0, 0, int(side)), p=(x, y, partials), l=(rows, self._fbits(scale)))

    def fp8_prep(self, items_dev, n_items, cout_pad, k, margin):
        """Delayed scaling: weights, epilogue scales and input scales of n_items fp8 convolutions in one launch (dsl_fp8_prep)."""
        self._add(L.OP_FP8_PREP, i=(n_items, cout_pad, k), p=(items_dev,), l=(0, self._fbits(margin)))

    def quant_fp8_delayed(self, x, y, rows, c, ld_x, scale_dev, partials, side=False):
        self._add(L.OP_QUANT_FP8_DELAYED, i=(c, ld_x, partials.numel(), 0, 0, 0, int(side)), p=(x, y, partials, scale_dev), l=(rows, 0))

    def fp8_comb(self, winv, comb, n, partials, side=False):
        self._add(L.OP_FP8_COMB, i=(n, 0, partials.numel(), 0, 0, 0, int(side)), p=(winv, comb, partials))

    def quant_fp8_w(self, w, w8, comb, cout, cout_pad, k, inv_act_scale, side=False):
        self._add(L.OP_QUANT_FP8_W, i=(cout, cout_pad, k, 0, 0, 0, int(side)), p=(w, w8, comb, None), l=(0, self._fbits(inv_act_scale)))

    def prof(self, cls, end, flops=0.0, nbytes=0.0):
        """Phase mark on the caller's stream (a no-op unless dsl_prof_enable(3))."""
        import struct
        bits = lambda x: struct.unpack('<q', struct.pack('<d', float(x)))[0]
        self._add(L.OP_PROF, i=(cls, int(end)), l=(bits(flops), bits(nbytes)))

    def join(self, side=1):
        self._add(L.OP_JOIN, i=(side,))

    def record(self, slot, stream=1):
        """Named event: everything queued so far on `stream` (1 = the side stream)."""
        self._add(L.OP_RECORD, i=(stream, slot))

    def wait(self, slot, stream=0):
        self._add(L.OP_WAIT, i=(stream, slot))

    def gn_fwd(self, d, side=False):
        self._add(L.OP_GN_FWD, d, i=(0, 0, 0, 0, 0, 0, int(side)))

    def gn_bwd(self, d, side=False):
        self._add(L.OP_GN_BWD, d, i=(0, 0, 0, 0, 0, 0, int(side)))

    def assign(self, d):
        self._add(L.OP_ASSIGN, d)

    def loss(self, d):
        self._add(L.OP_LOSS, d)

    def maxpool(self, x, y, n, h, w, c):
        self._add(L.OP_MAXPOOL, i=(n, h, w, c), p=(x, y))

    def sum2x2(self, g, out, n, h, w, ch, cw, c, side=False):
        self._add(L.OP_SUM2X2, i=(n, h, w, ch, cw, c, int(side)), p=(g, out))

    def memset(self, ptr, nbytes, value=0):
        self._add(L.OP_MEMSET, i=(value,), p=(ptr,), l=(nbytes,))

    def pack_image(self, img, out, n, h, w):
        self._add(L.OP_PACK_IMAGE, i=(n, h, w), p=(img, out))

    def stem_pool(self, img, w_groups, scale, bias, out, ld_out, n, h, w):
        """Image layout + conv1 + BN + ReLU + max pool as one kernel (dsl_stem_pool); scale / bias: raw device pointers."""
        self._add(L.OP_STEM_POOL, i=(ld_out, n, h, w), p=(img, w_groups, out), l=(int(scale), int(bias)))

    def run(self):
        if not self.items:
            return
        if self.arr is None:
            self.arr = (L.Op * len(self.items))(*self.items)
        L.check(L.lib.dsl_run_ops(self.arr, len(self.items), L.stream_ptr()), 'dsl_run_ops')


def conv_out(h, k, s, p):
    return (h + 2 * p - k) // s + 1


class Plan:
    """All buffers + op lists for one (N, H, W) and one ParamStore."""

    def __init__(self, store, N, H, W, training=True, max_gt=1024, single_stream=False):
        """single_stream: every op on the caller's stream (a teacher sweep that runs beside the student's step must not queue
        work on the library's side streams, which the student's op lists use in order)."""
        assert store.device.type == 'cuda' and not (training and single_stream)
        self.single_stream = single_stream
        if store.dirty:
            store.refresh()
        self.store, self.N, self.H, self.W, self.training = store, N, H, W, training
        # Deferred head update (set on the store by an optimizer that updates bucket by bucket, FlatSGD): the towers'
        # weight-gradient group - 423 GFLOP, a third of the backward pass's weight-gradient work - leaves the backward pass and runs
        # LAST on the weight-gradient stream, under the NEXT step's backbone forward (half of the chip idles there: two chains of
        # 33-66-workgroup launches), followed by the head + FPN bucket's optimizer step; the forward list waits for that update where
        # it first reads the bucket (SLOT_HEADW, in front of the FPN).  Same kernels, same arithmetic, same bits.
        self.defer = bool(training and getattr(store, 'defer_head', False) and store.backbone != 'rla'
                          and tune('side') != '0' and not single_stream)
        dev = store.device
        self.dev = dev
        self.bufs = {}
        self._multi_on, self._wg_pending = False, []
        self.fwd = OpList()
        self.loss_ops = OpList()
        self.assign_ops = OpList()
        self.bwd_segments = []          # [(OpList, dict(bucket=(lo, hi), slot=event slot, main=bool))] in execution order
        self.img = torch.zeros(N, 3, H, W, dtype=torch.float32, device=dev)
        # split-K scratch shared by all convs of the plan (they run back to back on one stream); the library
        # lowers its split factor if a conv would need more than this
        self.conv_ws = torch.empty(128 << 20, dtype=torch.uint8, device=dev)
        self._gn_ws = {}          # GroupNorm block-record workspaces, one per stream that runs GroupNorm launches
        self._build_forward()
        self.prefix = None
        if training and store.backbone != 'rla' and tune('pipe_prefix') != '0' and tune('side') != '0':
            self._split_prefix()
        if training:
            self.lossplan = FcosLossPlan(N, self.level_sizes, dev, max_gt=max_gt)
            self.lossplan.bind_outputs(self.bufs['cls_logits'], self.bufs['regctr'], store.t32_ptr('head.scales'))
            # scale gradients go straight into the flat gradient buffer
            self.lossplan.desc.g_scales = store.t32_ptr('head.scales', store.grad)
            self.assign_ops.assign(self.lossplan.desc)
            if self._fside:             # the regression tower's side stream: joined here, not at the end of the forward list,
                self.loss_ops.join(self._fside)       # so that target upload + assignment run while it finishes
            self.loss_ops.prof(4, 1)
            self.loss_ops.loss(self.lossplan.desc)
            self._build_backward()

    # ---------------------------------------------------------------------------------------------
    def buf(self, name, *shape, dtype=BF, zero=False):
        t = (torch.zeros if zero else torch.empty)(*shape, dtype=dtype, device=self.dev)
        self.bufs[name] = t
        return t

    def _conv(self, spec, src, dst, n, in_hw, out_hw, *, relu=False, addend=None, add_hw=None, flags=0, dst_ld=None,
              small_c=False, cs=None, lds=0, lda=None, affine=True):
        """cs: source channels read per tap (default: the stored weight rows' cin_store); lds: source pixel stride when the
        source is a channel slice of wider rows; lda: addend row stride (default cout); affine=False: no BN fold / bias."""
        st = self.store
        scale = bias = None
        if affine and spec.bn:
            scale, bias = st.bn_ptrs(spec.bn)
        elif affine and spec.bias:
            bias = st.t32_ptr(spec.name + '.bias')
        f = flags | (L.CONV_RELU_OUT if relu else 0) | (L.CONV_SMALL_C if small_c else 0)
        if add_hw is not None:
            f |= L.CONV_ADD_UPSAMPLE
        return ops.conv_desc(src, st.w16_ptr(spec), dst, n=n, grid=out_hw, src_hw=in_hw, dst_hw=out_hw,
                             cs=8 if small_c else (cs or spec.cin_store), cd=spec.cout, cd_pad=spec.cout_pad,
                             ldd=dst_ld or spec.cout, kh=spec.k, kw=spec.k, stride=spec.stride, pad=spec.pad,
                             flags=f, scale=scale, bias=bias, addend=addend, lda=lda or spec.cout, add_hw=add_hw,
                             workspace=None if small_c else self.conv_ws, lds=lds)

    def _build_forward(self):
        st, N, H, W, f = self.store, self.N, self.H, self.W, self.fwd
        cv = st.convs
        h1, w1 = conv_out(H, 7, 2, 3), conv_out(W, 7, 2, 3)
        h, w = conv_out(h1, 3, 2, 1), conv_out(w1, 3, 2, 1)
        # The frozen stem: image layout + conv1 + BN + ReLU + max pool as ONE kernel (dsl_stem_pool, csrc/stem.hip)
        self._stem_fused = True

        def emit_pool(out, ld):
            """The pooled stem output into rows of `ld` elements."""
            sc, bi = st.bn_ptrs(cv['backbone.conv1'].bn)
            self._img_op = len(f.items)         # bind_image() points this op at the caller's tensor
            f.stem_pool(self.img, st.stem_groups16, sc, bi, out, ld, N, H, W)
        self.stage_out, self.stage_ld = [], []      # per stage: (tensor / pointer of the output's first channel, (h, w)), row stride
        # independent branches of the forward graph (a stage's downsample conv next to conv1 -> conv2; P5 -> P6 -> P7 next to
        # the P4 / P3 path) run on side stream 3: their kernels are too small to fill the chip alone
        self.BR = 3 if (tune('side') != '0' and not self.single_stream) else 0
        self.conv_ws_br = torch.empty(32 << 20, dtype=torch.uint8, device=self.dev)
        if st.backbone == 'rla':
            from . import engine_rla
            engine_rla.build_forward(self, emit_pool, h1, w1)
        else:
            x = self.buf('pool', N, h, w, 64)
            emit_pool(x, 64)
            self._fwd_resnet(x, h, w)
        # ---- FPN (start_level=1) ----
        f.tag = 'fwd.fpn'
        (c3, hw3), (c4, hw4), (c5, hw5) = self.stage_out[1], self.stage_out[2], self.stage_out[3]
        hw6 = (conv_out(hw5[0], 3, 2, 1), conv_out(hw5[1], 3, 2, 1))
        hw7 = (conv_out(hw6[0], 3, 2, 1), conv_out(hw6[1], 3, 2, 1))
        self.level_sizes = [hw3, hw4, hw5, hw6, hw7]
        self.P = sum(a * b for a, b in self.level_sizes)
        self.M = N * self.P
        self.seg_off = [0]
        for a, b in self.level_sizes:
            self.seg_off.append(self.seg_off[-1] + N * a * b)
        lat = [self.buf(f'lat{i}', N, hw[0], hw[1], 256) for i, hw in enumerate((hw3, hw4, hw5))]
        lc = [cv[f'neck.lateral_convs.{i}.conv'] for i in range(3)]
        ld3, ld4, ld5 = self.stage_ld[1:4]       # RLA keeps a stage output as the x part of the next [C + 128]-wide block input
        BR = self.BR if st.backbone != 'rla' else 0

        def br(cd_):        # a conv of the side branch: its own split-K scratch
            if BR:
                cd_.workspace, cd_.workspace_bytes = L.ptr(self.conv_ws_br), self.conv_ws_br.numel()
            return cd_
        feats = self.buf('feats', self.M, 256)
        self.feat_seg = [feats.data_ptr() + self.seg_off[i] * 256 * 2 for i in range(5)]
        # late exchange (data parallel): bucket 0 (head + FPN) of the last optimizer step may still be on its way - its
        # all-reduce and update run beside this step's backbone.  Bucket 0 is the LAST one updated, so everything behind this wait
        # (the loss kernel and every gradient write of this step's backward pass) also follows every update.  No-op until recorded.
        f.wait(L.SLOT_UPD + 0, stream=0)
        if self.defer:
            # the FPN's and the head's parameters (one optimizer bucket) may still be on their way: the previous step's tower weight
            # gradients and that bucket's update run beside this step's backbone (no-op until the slot is first recorded).  The wait
            # also orders this step's head, which overwrites the towers' activations, behind the weight gradients that read them
            f.wait(L.SLOT_HEADW, stream=0)
        fc = [cv[f'neck.fpn_convs.{i}.conv'] for i in range(5)]
        p6r = self.buf('p6r', N, hw6[0], hw6[1], 256)
        f.conv(self._conv(lc[2], c5, lat[2], N, [hw5], [hw5], lds=ld5))
        if BR:
            f.fork(BR)
        # P5 -> P6 -> P7: five tiny launches, beside the P4 / P3 path
        f.conv(br(self._conv(fc[2], lat[2], self.feat_seg[2], N, [hw5], [hw5])), side=BR)
        f.conv(br(self._conv(fc[3], self.feat_seg[2], self.feat_seg[3], N, [hw5], [hw6])), side=BR)      # P6 (no relu)
        f.conv(br(self._conv(fc[3], self.feat_seg[2], p6r, N, [hw5], [hw6], relu=True)), side=BR)        # relu(P6)
        f.conv(br(self._conv(fc[4], p6r, self.feat_seg[4], N, [hw6], [hw7])), side=BR)                   # P7
        f.conv(self._conv(lc[1], c4, lat[1], N, [hw4], [hw4], addend=lat[2], add_hw=[hw5], lds=ld4))
        if BR:
            f.fork(BR)          # P4's output conv joins the side chain once its lateral exists: the caller's stream keeps C3 -> P3
        f.conv(br(self._conv(fc[1], lat[1], self.feat_seg[1], N, [hw4], [hw4])), side=BR)
        f.conv(self._conv(lc[0], c3, lat[0], N, [hw3], [hw3], addend=lat[1], add_hw=[hw4], lds=ld3))
        f.conv(self._conv(fc[0], lat[0], self.feat_seg[0], N, [hw3], [hw3]))
        if BR:
            f.join(BR)
        # ---- head: shared weights, all 5 levels per launch ----
        f.tag = 'fwd.head'
        ls = self.level_sizes
        self.tower = {}
        # the two towers are independent chains: the regression tower (+ its predictor) runs on the side stream
        self.conv_ws_side = torch.empty(32 << 20, dtype=torch.uint8, device=self.dev)
        FSIDE = 2 if (tune('side') != '0' and not self.single_stream) else 0      # side stream 1 carries the weight gradients (and may be CU-masked)
        # phase marks for bench.py: 8 tower convs + 2 predictors over all M locations, 8 GroupNorm+ReLU passes
        self._head_flops = 2.0 * self.M * (8 * 256 * 2304 + (80 + 5) * 2304)
        self._head_bytes = self.M * 256 * 2.0 * (8 * 2 + 8 * 3 + 2) + self.M * (80 + 8) * 4.0
        f.prof(4, 0, self._head_flops, self._head_bytes)
        # fp8 forward of the tower convolutions (FCOS(fp8=dict(...)), BASELINE.json configs[4], off by default), DELAYED scaling
        # (round 6): every fp8 tensor is written by its producer's own pass - GroupNorm's apply pass writes the e4m3 copy of its output
        # next to the bf16 one, with the scale of the PREVIOUS step's maximum, and leaves this step's block maxima; one dsl_fp8_prep
        # launch per step quantises the eight weight tensors (per-output-channel scales, from the fp32 master weights) and turns the
        # recorded maxima into this step's input scales and epilogue scales.  The bf16 tensors stay what the backward pass reads
        # (straight-through).  A plan's first forward pass runs twice more in front (Plan.fp8_warm): nothing is recorded yet.
        fp8 = getattr(st, 'fp8', None)
        f8 = bool(fp8) and 'towers' in str(fp8.get('layers', 'towers'))
        self.fp8_cold = f8
        feats8 = None
        if f8:
            margin = float(fp8.get('margin', 1.25))         # headroom over the previous step's maximum (e4m3 saturates at 448)
            nblk = (max(h_ * w_ for h_, w_ in ls) + 127) // 128
            feats8 = self.buf('feats.f8', self.M, 256, dtype=torch.uint8)
            am_feats = self.buf('feats.f8.amax', 512, dtype=torch.float32, zero=True)
            scales = self.buf('fp8.scales', 8, dtype=torch.float32, zero=True)
            items = (L.Fp8PrepItem * 8)()
            f8lay = {}
            for t_, tower in enumerate(('cls_convs', 'reg_convs')):
                for i in range(4):
                    spec = cv[f'bbox_head.{tower}.{i}.conv']
                    assert spec.cout_pad == 256 and spec.cin == 256 and spec.k == 3, 'the fp8 slice is the 3x3 256 -> 256 tower layers'
                    w8 = self.buf(f'{tower}.{i}.w8', spec.cout_pad, 9 * 256, dtype=torch.uint8)
                    comb = self.buf(f'{tower}.{i}.comb', spec.cout_pad, dtype=torch.float32)
                    am_out = self.buf(f'{tower}.{i}.act8.amax', 5 * N * nblk, dtype=torch.float32, zero=True) if i < 3 else None
                    am_in = am_feats if i == 0 else f8lay[tower, i - 1]['am_out']
                    it = items[t_ * 4 + i]
                    it.w, it.w8, it.comb = st.t32_ptr(spec.name + '.weight'), w8.data_ptr(), comb.data_ptr()
                    it.amax, it.n_amax, it.cout = am_in.data_ptr(), am_in.numel(), spec.cout
                    it.scale = scales.data_ptr() + 4 * (t_ * 4 + i)
                    f8lay[tower, i] = dict(w8=w8, comb=comb, am_out=am_out, scale_in=it.scale)
            items_dev = torch.frombuffer(bytearray(bytes(items)), dtype=torch.uint8).to(self.dev)
            self.bufs['fp8.items'] = items_dev
            f.fp8_prep(items_dev, 8, 256, 9 * 256, margin)
            # the FPN outputs feed both towers: one pass writes their e4m3 copy (both first layers' scales are the same number)
            f.quant_fp8_delayed(feats, feats8, self.M, 256, 256, f8lay['cls_convs', 0]['scale_in'], am_feats)
        if FSIDE:        # (starting the regression tower one convolution late, so that each tower's GroupNorm runs beside the other's
            f.fork(FSIDE)    # convolution instead of beside its GroupNorm: measured, 407.6 vs 408.9 img/s - no effect)
        for tower in ('cls_convs', 'reg_convs'):
            side = FSIDE if tower == 'reg_convs' else 0
            xin, xin8 = feats, feats8
            lays = []
            for i in range(4):
                spec = cv[f'bbox_head.{tower}.{i}.conv']
                pre = self.buf(f'{tower}.{i}.pre', self.M, 256)
                act = self.buf(f'{tower}.{i}.act', self.M, 256)
                stats = self.buf(f'{tower}.{i}.stats', 5 * N * 32, 2, dtype=torch.float32)
                if f8:
                    cd_ = ops.conv_desc(xin8, f8lay[tower, i]['w8'], pre, n=N, grid=ls, src_hw=ls, dst_hw=ls, cs=256, cd=spec.cout,
                                        cd_pad=spec.cout_pad, ldd=spec.cout, kh=3, kw=3, stride=1, pad=1, flags=L.CONV_FP8,
                                        scale=f8lay[tower, i]['comb'], bias=st.t32_ptr(spec.name + '.bias'))
                else:
                    cd_ = self._conv(spec, xin, pre, N, ls, ls)
                if side:
                    cd_.workspace, cd_.workspace_bytes = L.ptr(self.conv_ws_side), self.conv_ws_side.numel()
                base = f'bbox_head.{tower}.{i}.gn'
                act8 = self.buf(f'{tower}.{i}.act8', self.M, 256, dtype=torch.uint8) if f8 and i < 3 else None
                gd = ops.gn_desc(pre, act, st.t32_ptr(base + '.weight'), st.t32_ptr(base + '.bias'), stats,
                                 self._gn_workspace('side' if side else 'main'), n=N, hw=ls, y8=act8,
                                 y8_scale=f8lay[tower, i + 1]['scale_in'] if act8 is not None else None,
                                 y8_amax=f8lay[tower, i]['am_out'] if act8 is not None else None)
                # conv -> GN -> ReLU (ConvModule, anchor_free_head.py:104-133): the convolution's epilogue leaves the statistics
                # records, GroupNorm is then ONE pass over the tensor (a launch that cannot leave them: its own statistics pass)
                if L.lib.dsl_conv2d_gn_fusable(C.byref(cd_)):
                    cd_.gn_ws = gd.workspace
                    gd.conv_stats = 1
                f.conv(cd_, side=side)
                f.gn_fwd(gd, side=side)
                lays.append(dict(spec=spec, xin=xin, pre=pre, act=act, stats=stats, gn=base))
                xin, xin8 = act, act8
            self.tower[tower] = lays
        cls_logits = self.buf('cls_logits', self.M, 80, dtype=torch.float32)
        regctr = self.buf('regctr', self.M, 8, dtype=torch.float32, zero=True)
        f.conv(ops.conv_desc(self.tower['cls_convs'][3]['act'], st.t16_ptr('head.cls_w'), cls_logits, n=N, grid=ls,
                             src_hw=ls, dst_hw=ls, cs=256, cd=80, cd_pad=128, ldd=80, kh=3, kw=3, stride=1, pad=1,
                             flags=L.CONV_OUT_F32, bias=st.t32_ptr('head.cls_b'), workspace=self.conv_ws))
        f.conv(ops.conv_desc(self.tower['reg_convs'][3]['act'], st.t16_ptr('head.regctr_w'), regctr, n=N, grid=ls,
                             src_hw=ls, dst_hw=ls, cs=256, cd=5, cd_pad=64, ldd=8, kh=3, kw=3, stride=1, pad=1,
                             flags=L.CONV_OUT_F32, bias=st.t32_ptr('head.regctr_b'), workspace=self.conv_ws_side),
               side=FSIDE)
        self._fside = FSIDE
        if FSIDE and not self.training:
            f.join(FSIDE)               # (training plans: the loss op list starts with this join, see __init__)
        if not self.training:
            f.prof(4, 1)

    def _fwd_resnet(self, x, h, w):
        st, N, f = self.store, self.N, self.fwd
        cv = st.convs
        self.blocks = []        # per block: dict(xin, a1, a2, out, idt, in_hw, out_hw, prefix, stride)
        self._pp = {}           # descriptors that touch layer1's output (pipelined prefix: two buffers, patched per step)
        # Image-split stages (tuning key img_split; measured in round 3's first half: none 372.6, 3: 384.0, 34: 383.4, 4: 372.1, 23: -0.5 % vs 3; re-measured on
        # the final kernels, two boxes, both orders (profiles/r03_step_boundary.txt): 3: 416.8 / 432.1, 34: 420.3 / 435.1, 23: 420.0, 24: 420.9 (second
        # box), 234: 420.9 / 436.2 img/s -> default layer2 + layer3 + layer4, + 1 %;
        # profiles/r03b_* are the profiles of this setting, r03_* those of the layer3 split): their launches are 66-132 workgroups of 15-40 us - mostly
        # fill, epilogue and kernel boundary on half a chip.  The images of a batch are independent through the backbone, so the
        # batch goes through these stages as TWO chains (images [0, ceil(N/2)) on the caller's stream, the rest on stream 3) of
        # half-size launches: one chain's fixed per-launch costs hide under the other chain's kernels.
        SPLIT = tune('img_split') if (self.BR and N >= 2 and self.training) else ''
        split_open = False

        def br_ws(d_):
            d_.workspace, d_.workspace_bytes = L.ptr(self.conv_ws_br), self.conv_ws_br.numel()
            return d_
        for li, (planes, nb) in enumerate(zip(STAGE_PLANES, STAGE_BLOCKS)):
            # Fused stages (tuning key bneck_fwd, default layer2; '23' adds layer3): every bottleneck's forward pass is ONE launch over the whole
            # batch (dsl_bottleneck_fwd, csrc/bneck.hip) - the image-split chains of three launches per block and half are its predecessor
            fuse = str(li + 1) in tune('bneck_fwd') and planes in (128, 256)
            split = str(li + 1) in SPLIT and not fuse
            f.tag = f'fwd.l{li + 1}'
            if li >= 1 and self.training:
                # late exchange: this stage's gradient bucket (layer2: 3, layer3: 2, layer4: 1) of the last optimizer step may still be
                # exchanged / updated beside the stages in front of it; both image chains wait (no-ops until the slot is recorded)
                f.wait(L.SLOT_UPD + 4 - li, stream=0)
                if split_open:
                    f.wait(L.SLOT_UPD + 4 - li, stream=self.BR)
            if split and not split_open:
                f.fork(self.BR)
                split_open = True
            elif split_open and not split:
                f.join(self.BR)
                split_open = False
            groups = [(0, (N + 1) // 2, 0), ((N + 1) // 2, N, self.BR)] if split else [(0, N, 0)]
            for b in range(nb):
                p = f'backbone.layer{li + 1}.{b}'
                c1, c2, c3 = cv[p + '.conv1'], cv[p + '.conv2'], cv[p + '.conv3']
                s = c1.stride
                oh, ow = conv_out(h, 1, s, 0), conv_out(w, 1, s, 0)
                a1 = self.buf(p + '.a1', N, oh, ow, planes)
                a2 = self.buf(p + '.a2', N, oh, ow, planes)
                out = self.buf(p + '.out', N, oh, ow, planes * 4)
                idt = x
                if b == 0:      # the downsample conv only meets the main branch at conv3's residual add
                    idt = self.buf(p + '.idt', N, oh, ow, planes * 4)
                if split:
                    for g0, g1, sd in groups:
                        wsf = br_ws if sd else (lambda d_: d_)
                        n_ = g1 - g0
                        off = g0 * h * w * x.shape[-1] * 2          # this group's byte offset into the stage input
                        if b == 0:
                            dd = wsf(self._conv(cv[p + '.downsample.0'], x[g0:g1], idt[g0:g1], n_, [(h, w)], [(oh, ow)]))
                            f.conv(dd, side=sd)
                            if li == 1:
                                self._pp.setdefault('ds', []).append((dd, off))
                        d1 = wsf(self._conv(c1, x[g0:g1], a1[g0:g1], n_, [(h, w)], [(oh, ow)], relu=True))
                        f.conv(d1, side=sd)
                        if li == 1 and b == 0:
                            self._pp.setdefault('c1', []).append((d1, off))
                        f.conv(wsf(self._conv(c2, a1[g0:g1], a2[g0:g1], n_, [(oh, ow)], [(oh, ow)], relu=True)), side=sd)
                        f.conv(wsf(self._conv(c3, a2[g0:g1], out[g0:g1], n_, [(oh, ow)], [(oh, ow)], relu=True, addend=idt[g0:g1])), side=sd)
                    self.blocks.append(dict(prefix=p, xin=x, a1=a1, a2=a2, out=out, in_hw=(h, w), out_hw=(oh, ow), stride=s,
                                            stage=li, b=b, planes=planes))
                    x, h, w = out, oh, ow
                    continue
                use_br = self.BR
                bd = None
                if fuse:
                    bd = ops.bneck_desc(x, st.w16_ptr(c1), st.w16_ptr(c2), st.w16_ptr(c3), idt, st.bn_ptrs(c1.bn), st.bn_ptrs(c2.bn),
                                        st.bn_ptrs(c3.bn), a1, a2, out, n=N, hin=h, win=w, h=oh, w=ow, planes=planes, cin=c1.cin_store,
                                        ldx=int(x.shape[-1]), stride=s)
                    if not L.lib.dsl_bottleneck_fwd_supported(C.byref(bd)):
                        bd = None
                if bd is not None:
                    if b == 0:          # the downsample branch beside nothing: it IS the block's identity, joined in front of the launch
                        dd = self._conv(cv[p + '.downsample.0'], x, idt, N, [(h, w)], [(oh, ow)])
                        f.conv(dd)
                        if li == 1:
                            self._pp['ds'] = [(dd, 0)]
                            self._pp['c1'] = [(bd, 0)]
                    f.bneck(bd)
                    self.blocks.append(dict(prefix=p, xin=x, a1=a1, a2=a2, out=out, in_hw=(h, w), out_hw=(oh, ow), stride=s,
                                            stage=li, b=b, planes=planes))
                    x, h, w = out, oh, ow
                    continue
                if b == 0:
                    dd = self._conv(cv[p + '.downsample.0'], x, idt, N, [(h, w)], [(oh, ow)])
                    if use_br:
                        dd.workspace, dd.workspace_bytes = L.ptr(self.conv_ws_br), self.conv_ws_br.numel()
                        f.fork(self.BR)
                    f.conv(dd, side=use_br)
                if li == 1 and b == 0:
                    self._pp['ds'] = [(dd, 0)]
                d1 = self._conv(c1, x, a1, N, [(h, w)], [(oh, ow)], relu=True)
                if li == 1 and b == 0:
                    self._pp['c1'] = [(d1, 0)]
                f.conv(d1)
                f.conv(self._conv(c2, a1, a2, N, [(oh, ow)], [(oh, ow)], relu=True))
                if b == 0 and use_br:
                    f.join(self.BR)
                d3 = self._conv(c3, a2, out, N, [(oh, ow)], [(oh, ow)], relu=True, addend=idt)
                if li == 0 and b == nb - 1:
                    self._pp['c3'], self._pp['out'] = d3, out
                f.conv(d3)
                self.blocks.append(dict(prefix=p, xin=x, a1=a1, a2=a2, out=out, in_hw=(h, w), out_hw=(oh, ow), stride=s,
                                        stage=li, b=b, planes=planes))
                x, h, w = out, oh, ow
            self.stage_out.append((x, (h, w)))
            self.stage_ld.append(planes * 4)
            if li == 0:
                self._prefix_end = len(f.items)        # pack + stem + pool + layer1: frozen weights, a function of the image alone
        if split_open:
            f.join(self.BR)

    def _gn_workspace(self, key):
        """The block-record scratch of one chain of (record-writing launch, GroupNorm pass) pairs that run one after the other."""
        ws = self._gn_ws.get(key)
        if ws is None:
            d = L.GnDesc()
            d.nseg, d.n, d.c, d.groups = len(self.level_sizes), self.N, 256, 32
            d.h, d.w = ops._segs(self.level_sizes)
            ws = torch.empty((L.lib.dsl_groupnorm_workspace_bytes(C.byref(d)) + 3) // 4, dtype=torch.float32, device=self.dev)
            self._gn_ws[key] = ws
        return ws

    # ---------------------------------------------------------------------------------------------
    def _wgrad(self, ol, spec, dy, x, n, out_hw, in_hw, cy=None, cd=None, wregion=None, bregion=None, side=False,
               emit=True, no_db=False, ldx=0, shared=0, raw=False, db_ptr=None, slots=0):
        """emit=False: only build the descriptor (for a later grouped launch).  raw=True: no BatchNorm scale on the rows (the
        BN post-pass derives dgamma from the unscaled gradient and scales it afterwards); db_ptr: where the column sum of dy
        goes (that BatchNorm's dbeta)."""
        st = self.store
        scale = st.bn_ptrs(spec.bn)[0] if (spec is not None and spec.bn and not raw) else None
        name = wregion or (spec.name + '.weight')
        db = db_ptr
        if bregion is not None:
            db = st.t32_ptr(bregion, st.grad)
        elif spec is not None and spec.bias and not no_db:
            db = st.t32_ptr(spec.name + '.bias', st.grad)
        k = 3 if spec is None else spec.k
        cs = 256 if spec is None else spec.cin_store
        d = ops.wgrad_desc(dy, x, st.t32_ptr(name, st.grad), n=n, grid=out_hw, src_hw=in_hw,
                           cs=cs, cy=cy or spec.cout_pad, cd=cd or spec.cout,
                           kh=k, kw=k, stride=1 if spec is None else spec.stride, pad=1 if spec is None else spec.pad,
                           scale=scale, db=db, workspace=self._wg_ws(n, out_hw, in_hw, spec, cy), ldx=ldx, shared=shared, slots=slots)
        if emit:
            if side and self._multi_on:
                self._wg_pending.append([d])        # goes out with the next _flush_wgrads
            else:
                ol.wgrad(d, side=side)
        return d

    def _wgrad_group(self, ol, descs, side=True, ws_name='wg_ws'):
        """One launch for same-geometry convolutions (the blocks of a ResNet stage, the layers of a head tower): the
        split-K partial traffic of the group is what ONE of its members would need alone."""
        if not descs:
            return
        if side and self._multi_on:
            self._wg_pending.append(list(descs))
            return
        if len(descs) == 1:
            if ws_name != 'wg_ws':
                # the descriptor was built with the shared 'wg_ws' scratch: a launch that runs on ANOTHER stream beside the
                # weight-gradient stream's needs its own split-K scratch as the multi-member groups get below (a one-member group on
                # the caller's stream raced the side stream's partial tiles: round 4, test_training_step_is_bit_reproducible)
                need = L.lib.dsl_wgrad_workspace_bytes(C.byref(descs[0]))
                ws = self._wg_buf(need, ws_name)
                descs[0].workspace, descs[0].workspace_bytes = L.ptr(ws), ws.numel()
            ol.wgrad(descs[0], side=side)
            return
        arr0 = (L.WgradDesc * len(descs))()
        for i, d in enumerate(descs):
            C.memmove(C.addressof(arr0[i]), C.addressof(d), C.sizeof(L.WgradDesc))
        need = L.lib.dsl_wgrad_group_workspace_bytes(arr0, len(descs))
        ol.wgrad_group(ops.wgrad_group(descs, workspace=self._wg_buf(need, ws_name)), side=side)

    def _flush_wgrads(self, ol, side=True, ws_name='wg_ws'):
        """Emits the deferred weight gradients: all sub-launches of one tile configuration as ONE multi launch
        (dsl_conv2d_wgrad_multi: one grid + one reduce grid, split factors chosen for the launch as a whole)."""
        pending, self._wg_pending = self._wg_pending, []
        if not pending:
            return
        by_cfg = {}
        for sub in pending:
            by_cfg.setdefault(L.lib.dsl_wgrad_multi_config(C.byref(sub[0])), []).append(sub)
        on, self._multi_on = self._multi_on, False
        for cfg, subs in by_cfg.items():
            for lo in range(0, len(subs), L.MAX_MULTI):
                chunk = subs[lo:lo + L.MAX_MULTI]
                if cfg == 0 or len(chunk) == 1:
                    for sub in chunk:
                        self._wgrad_group(ol, sub, side=side, ws_name=ws_name)
                else:
                    ol.wgrad_multi(ops.WgradMulti(chunk, workspace=lambda need: self._wg_buf(need, ws_name)), side=side)
        self._multi_on = on

    def _wg_ws(self, n, out_hw, in_hw, spec, cy):
        need = ops.wgrad_workspace_bytes(n=n, grid=out_hw, src_hw=in_hw, cs=256 if spec is None else spec.cin_store,
                                         cy=cy or spec.cout_pad, cd=1, kh=3 if spec is None else spec.k,
                                         kw=3 if spec is None else spec.k, stride=1 if spec is None else spec.stride,
                                         pad=1 if spec is None else spec.pad)
        return self._wg_buf(need)

    def _wg_buf(self, need, name='wg_ws'):
        """The split-K partial-tile scratch of the weight gradients (they run one after the other on one stream)."""
        ws = self.bufs.get(name)
        if ws is None or ws.numel() < need:
            # grow: descriptors built earlier keep pointing at the old (smaller, still alive) buffer
            ws = torch.empty(max(need, 96 << 20), dtype=torch.uint8, device=self.dev)
            self.bufs.setdefault(name + '_old', []).append(self.bufs.get(name))
            self.bufs[name] = ws
        return ws

    def _dgrad(self, name, dy, dst, n, dy_hw, dst_hw, *, cs, cd, k, stride, pad, os=1, addend=None, mask=None,
               mask_first=False, mask_last=False, cs_real=0, wptr=None, cd_pad=None, ldd=None, lda=None, ldm=None):
        """wptr: explicit pointer into a dgrad pack (a row range = an input-channel range of the forward conv)."""
        st = self.store
        f = (L.CONV_MASK_FIRST if mask_first else 0) | (L.CONV_MASK_LAST if mask_last else 0)
        grid = dy_hw if os > 1 else dst_hw
        return ops.conv_desc(dy, wptr if wptr is not None else st.wT_ptr(name), dst, n=n, grid=grid, src_hw=dy_hw, dst_hw=dst_hw,
                             cs=cs, cd=cd, cd_pad=cd_pad or cd, ldd=ldd or cd, kh=k, kw=k, stride=stride, pad=pad, mode=1, os=os,
                             flags=f, addend=addend, lda=lda or cd, mask=mask, ldm=ldm or cd, workspace=self.conv_ws,
                             cs_real=cs_real)

    def _build_backward(self):
        """Backward op lists.  The data-gradient chain runs on the caller's stream; every weight gradient is
        forked onto the library's side stream (its inputs - the forward activation and a gradient buffer that is
        written exactly once per step - are complete at the fork point), so the many small, under-filled
        backbone launches of the chain overlap with weight-gradient work.  Each segment ends with a JOIN."""
        st, N, ls = self.store, self.N, self.level_sizes
        lp = self.lossplan
        reg = st.train_regions
        M = self.M
        SIDE = tune('side') != '0'
        # tuning knobs (both measured: on is better): group the last segment's weight gradients too, although nothing
        # is left on the caller's stream to overlap their tail with ...
        GROUP_LAST = True
        TAIL_SLOTS = tune_int('tail_slots')      # (round 3, tools/exp_env.sh: 160 / 192 / 224 / 256 = 421.3 / 421.9 / 419.8 / 418.2 img/s)
        GROUP = True       # same-geometry weight gradients (tower layers, the blocks of a stage) as one launch
        # ... and all weight gradients of a segment that share a tile configuration as one multi launch
        self._multi_on = SIDE
        self._wg_pending = []
        g_feats = self.buf('g_feats', M, 256)
        # ================= segment 0: head + FPN =================
        # where in this pass the next step's frozen prefix (FCOS.pipeline_prefix) may start: 0 = with the pass, 3 / 2 / 1 = behind
        # layer4's / layer3's / layer2's data gradients (1 = the whole chain, the round-2 setting)
        PREFIX_LI = 2       # measured (tools/exp_env.sh): 1: 355.5, 3: 360, 2 / 0: +0.2 % over 3
        ol = OpList()
        ol.tag = 'bwd.head'
        ol.wait(L.SLOT_PACKS, stream=0)      # the data-gradient weight packs of the last optimizer step (ParamStore.repack_dgrad)
        if PREFIX_LI == 0:
            ol.record(L.SLOT_TAIL, stream=0)
        ol.prof(5, 0, self._head_flops, self.M * 256 * 2.0 * (8 * 2 + 8 * 5 + 2))       # same FLOPs as the forward phase: data gradients
        tower_group = []
        # the two towers' backward chains are independent until both have added into g_feats: the regression tower's runs on
        # side stream 2 (as in the forward pass); its last data gradient - the one that adds into g_feats - waits for the
        # classification tower's
        BT = 2 if SIDE else 0
        if BT:
            ol.fork(BT)

        def side_ws(cd_):
            cd_.workspace, cd_.workspace_bytes = L.ptr(self.conv_ws_side), self.conv_ws_side.numel()
            return cd_
        towers = (('reg_convs', BT), ('cls_convs', 0)) if BT else (('cls_convs', 0), ('reg_convs', 0))
        # the predictors' weight gradients need nothing from this pass but the loss gradients: first thing on the side stream
        self._wgrad(ol, None, lp.g_cls, self.tower['cls_convs'][3]['act'], N, ls, ls, cy=128, cd=80, wregion='head.cls_w',
                    bregion='head.cls_b', side=SIDE)
        self._wgrad(ol, None, lp.g_rc, self.tower['reg_convs'][3]['act'], N, ls, ls, cy=64, cd=5, wregion='head.regctr_w',
                    bregion='head.regctr_b', side=SIDE)
        # (measured, tools/experiments_r2.txt (exp_r2t) / exp_r2u.sh: weight gradients that start while the head's large data-gradient launches
        # still run cost more than the idle side stream saves - predictors first: 5.97 vs 5.94 ms, tower halves: 6.10 vs 6.05)
        g_act = {}
        # The data gradient that produces dY of a tower layer's GroupNorm + ReLU also leaves that norm's backward block records
        # (dsl_conv_desc.gn_x: x is read once more in the epilogue, dY never again): GroupNorm backward is then ONE pass
        # (a data gradient that cannot leave them: GroupNorm's own reduction pass first)
        gn_from_conv = set()

        def gn_records(cd_, tower, j, sd):
            lay = self.tower[tower][j]
            cd_.gn_x = L.ptr(lay['pre'])                 # (set first: the query answers for the backward records' tiles)
            if L.lib.dsl_conv2d_gn_fusable(C.byref(cd_)):
                cd_.gn_ws = L.ptr(self._gn_workspace('bwd.' + tower))      # per tower: with one stream for both (tuning side=0) the second
                #                                                             tower's data gradient would overwrite the first one's records
                cd_.gn_gamma, cd_.gn_beta = L.ptr(st.t32_ptr(lay['gn'] + '.weight')), L.ptr(st.t32_ptr(lay['gn'] + '.bias'))
                cd_.gn_stats = L.ptr(lay['stats'])
                gn_from_conv.add((tower, j))
            else:
                cd_.gn_x = None
            return cd_
        for tower, sd in towers:
            wsf = side_ws if sd else (lambda c: c)
            g_act[tower] = self.buf(f'g_{tower}_act3', M, 256)
            if tower == 'cls_convs':
                ol.conv(gn_records(wsf(self._dgrad('head.cls', lp.g_cls, g_act[tower], N, ls, ls, cs=128, cd=256, k=3, stride=1, pad=1, cs_real=80)),
                                   tower, 3, sd), side=sd)
            else:
                ol.conv(gn_records(wsf(self._dgrad('head.regctr', lp.g_rc, g_act[tower], N, ls, ls, cs=64, cd=256, k=3, stride=1, pad=1, cs_real=5)),
                                   tower, 3, sd), side=sd)
        # layer by layer, both towers: their weight gradients go out in two groups of four (layers 3, 2 and layers 1, 0 of
        # both towers) as soon as the GroupNorm backward passes that produce their dY are queued - the side stream works from
        # the first quarter of this segment on instead of waiting for its end
        for i in (3, 2, 1, 0):
            g_pre = {}
            for tower, sd in towers:
                lay = self.tower[tower][i]
                base = lay['gn']
                g_pre[tower] = self.buf(f'g_{tower}_pre{i}', M, 256)
                # the GroupNorm backward also yields the conv bias gradient (sum over pixels of g_pre) from its block
                # records: the weight gradient below runs without its column-sum pass
                gd = ops.gn_desc(lay['pre'], lay['act'], st.t32_ptr(base + '.weight'), st.t32_ptr(base + '.bias'),
                                 lay['stats'], self._gn_workspace('bwd.' + tower), n=N, hw=ls, dy=g_act[tower], dx=g_pre[tower],
                                 dgamma=st.t32_ptr(base + '.weight', st.grad), dbeta=st.t32_ptr(base + '.bias', st.grad),
                                 dbias=st.t32_ptr(lay['spec'].name + '.bias', st.grad))
                gd.conv_stats = 1 if (tower, i) in gn_from_conv else 0
                ol.gn_bwd(gd, side=sd)
                tower_group.append(self._wgrad(ol, lay['spec'], g_pre[tower], lay['xin'], N, ls, ls, side=SIDE, emit=False, no_db=True,
                                               slots=DEFER_SLOTS if self.defer else tune_int('tower_slots')))     # measured: tools/experiments_r2.txt (exp_r2z) (48-128: 5.84 ms, 160-192: 5.89); round 3: 72 / 96 / 128: 5.435 / 5.461 / 5.456
            if i == 0:
                if BT and SIDE:
                    ol.fork(1, other=BT)       # the weight-gradient stream also waits for the regression tower's stream
                if self.defer:
                    self._deferred_towers = list(tower_group)      # emitted behind the last segment, see below
                else:
                    self._wgrad_group(ol, tower_group, side=SIDE)
                self._flush_wgrads(ol, side=SIDE)
                tower_group = []
            if i > 0:
                for tower, sd in towers:
                    wsf = side_ws if sd else (lambda c: c)
                    lay = self.tower[tower][i]
                    g_act[tower] = self.buf(f'g_{tower}_act{i - 1}', M, 256)
                    ol.conv(gn_records(wsf(self._dgrad(lay['spec'].name, g_pre[tower], g_act[tower], N, ls, ls, cs=256, cd=256, k=3, stride=1, pad=1)),
                                       tower, i - 1, sd), side=sd)
            else:
                cl, rl = self.tower['cls_convs'][0], self.tower['reg_convs'][0]
                ol.conv(self._dgrad(cl['spec'].name, g_pre['cls_convs'], g_feats, N, ls, ls, cs=256, cd=256, k=3, stride=1, pad=1))
                # the regression tower's last data gradient adds into g_feats, after the classification tower's: on the caller's
                # stream behind a JOIN of the tower stream (a FORK -> side launch -> JOIN round trip with nothing else to do on the
                # caller's stream costs ~27 us, tools/microbench/sync_cost.hip)
                if BT:
                    ol.join(BT)
                ol.conv(self._dgrad(rl['spec'].name, g_pre['reg_convs'], g_feats, N, ls, ls, cs=256, cd=256, k=3, stride=1, pad=1, addend=g_feats))
                ol.prof(5, 1)
        self._flush_wgrads(ol, side=SIDE)          # towers + predictors: ready now, the FPN's follow below
        # ---- FPN backward ----
        ol.tag = 'bwd.fpn'
        cv = st.convs
        fc = [cv[f'neck.fpn_convs.{i}.conv'] for i in range(5)]
        lc = [cv[f'neck.lateral_convs.{i}.conv'] for i in range(3)]
        hw3, hw4, hw5, hw6, hw7 = ls
        gseg = [g_feats.data_ptr() + self.seg_off[i] * 256 * 2 for i in range(5)]
        lat = [self.bufs[f'lat{i}'] for i in range(3)]
        p6r = self.bufs['p6r']
        g_p6 = self.buf('g_p6', N, hw6[0], hw6[1], 256)
        g_p5 = self.buf('g_p5', N, hw5[0], hw5[1], 256)
        g_lat = [self.buf(f'g_lat{i}', N, hw[0], hw[1], 256) for i, hw in enumerate((hw3, hw4, hw5))]
        s0 = self.buf('s0', N, hw4[0], hw4[1], 256)
        s1 = self.buf('s1', N, hw5[0], hw5[1], 256)
        # independent chains of this part of the graph run on side stream 3 beside the caller's (small launches, as in the
        # forward pass): the P3 / P4 output convs next to P7 -> P6 -> P5; the C3 / C4 laterals next to layer4's backward
        rla = st.backbone == 'rla'
        BB = 3 if (SIDE and not rla) else 0

        def br_ws(cd_):
            if BB:
                cd_.workspace, cd_.workspace_bytes = L.ptr(self.conv_ws_br), self.conv_ws_br.numel()
            return cd_
        if BB:
            ol.fork(BB)
        # P3, P4 = fpn_i(lat_i); the top-down path's nearest upsampling is a 2x2 sum in the backward direction
        ol.conv(br_ws(self._dgrad(fc[0].name, gseg[0], g_lat[0], N, [hw3], [hw3], cs=256, cd=256, k=3, stride=1, pad=1)), side=BB)
        ol.sum2x2(g_lat[0], s0, N, hw4[0], hw4[1], hw3[0], hw3[1], 256, side=BB)
        ol.conv(br_ws(self._dgrad(fc[1].name, gseg[1], g_lat[1], N, [hw4], [hw4], cs=256, cd=256, k=3, stride=1, pad=1,
                                  addend=s0)), side=BB)
        ol.sum2x2(g_lat[1], s1, N, hw5[0], hw5[1], hw4[0], hw4[1], 256, side=BB)
        # P7 = fpn4(relu(P6))
        ol.conv(self._dgrad(fc[4].name, gseg[4], g_p6, N, [hw7], [hw6], cs=256, cd=256, k=3, stride=2, pad=1,
                            addend=gseg[3], mask=p6r, mask_first=True))
        # P6 = fpn3(P5)
        ol.conv(self._dgrad(fc[3].name, g_p6, g_p5, N, [hw6], [hw5], cs=256, cd=256, k=3, stride=2, pad=1,
                            addend=gseg[2]))
        if BB:
            ol.join(BB)
        ol.conv(self._dgrad(fc[2].name, g_p5, g_lat[2], N, [hw5], [hw5], cs=256, cd=256, k=3, stride=1, pad=1,
                            addend=s1))
        # the output convolutions' weight gradients: behind the data gradients that produce g_p6 / g_p5 (with side streams they
        # are collected for the segment's multi launch anyway; on one stream - tuning side=0 - they run right here, and in front of
        # those data gradients they read buffers nobody had written: found by the side=0 leg of test_train_step_vs_reference_and_oracle)
        self._wgrad(ol, fc[4], gseg[4], p6r, N, [hw7], [hw6], side=SIDE)
        self._wgrad(ol, fc[3], g_p6, self.feat_seg[2], N, [hw6], [hw5], side=SIDE)
        self._wgrad(ol, fc[2], g_p5, lat[2], N, [hw5], [hw5], side=SIDE)
        self._wgrad(ol, fc[1], gseg[1], lat[1], N, [hw4], [hw4], side=SIDE)
        self._wgrad(ol, fc[0], gseg[0], lat[0], N, [hw3], [hw3], side=SIDE)
        # laterals -> gradients w.r.t. C3, C4, C5 (masked by the ReLU that produced them)
        self.g_stage = {}
        self._br_pending = False        # side-stream-3 work the next segment must join before it touches g_stage[1], g_stage[2]
        for i, (li, hw) in ((2, (3, hw5)), (1, (2, hw4)), (0, (1, hw3))):
            cfeat = self.stage_out[li][0]
            cch = STAGE_PLANES[li] * 4
            ldc = self.stage_ld[li]
            self._wgrad(ol, lc[i], g_lat[i], cfeat, N, [hw], [hw], side=SIDE, ldx=ldc if ldc != cch else 0)
            g0b = self.buf(f'g_stage{li}_last', N, hw[0], hw[1], cch)
            self.g_stage[li] = g0b
            # ResNet: the gradient w.r.t. the stage output is masked by its ReLU here.  RLA: the outputs of stages 1, 2 also
            # feed the recurrent path; their mask is applied once every contribution has arrived (engine_rla)
            masked = not rla or li == 3
            sd = BB if li < 3 else 0        # layer4's backward starts from the C5 lateral alone; C4, C3 are needed at its end
            if sd and not self._br_pending:
                ol.fork(BB)
                self._br_pending = True
            ol.conv(br_ws(self._dgrad(lc[i].name, g_lat[i], g0b, N, [hw], [hw], cs=256, cd=cch, k=1, stride=1, pad=0,
                                      mask=cfeat if masked else None, mask_first=masked, ldm=ldc)) if sd else
                    self._dgrad(lc[i].name, g_lat[i], g0b, N, [hw], [hw], cs=256, cd=cch, k=1, stride=1, pad=0,
                                mask=cfeat if masked else None, mask_first=masked, ldm=ldc), side=sd)
        buckets = st.grad_buckets()
        # no JOIN here: this segment's weight gradients keep running on the side stream under the next segment's
        # data-gradient chain.  Named event slot s marks "side-stream work of segment s queued": once it has fired, gradient
        # bucket s is complete - the data-parallel wrapper's communication stream waits for exactly that
        # (dsl_stream_wait_slot) and starts the bucket's all-reduce, independent of the caller's stream.
        self._flush_wgrads(ol, side=SIDE)
        if self.defer:
            # bucket 0 is complete only behind the deferred tower group: its entry follows the last segment (bucket None = no bucket
            # completes with this list)
            self.bwd_segments.append((ol, dict(bucket=None, slot=None, main=False)))
        else:
            ol.record(0)
            self.bwd_segments.append((ol, dict(bucket=buckets[0], slot=0, main=False)))
        # ================= backbone: layer4, layer3, layer2 =================
        if rla:
            from . import engine_rla
            # (the recurrent path's BatchNorm post-pass runs on the weight-gradient stream behind the stage's flush)
            self._multi_on = SIDE
            engine_rla.build_backward(self, buckets, SIDE)
            return
        blocks_by_stage = {li: [b for b in self.blocks if b['stage'] == li] for li in (1, 2, 3)}
        BSPLIT = tune('img_split_bwd')      # measured (tools/exp_env.sh): '' 372-377, 3: 378.5, 23: 382.5, 34: 381.5, 234: 379.7 img/s
        for li in (3, 2, 1):
            ol = OpList()
            ol.tag = f'bwd.l{li + 1}'
            if li == 1:
                self._multi_on = False         # layer2: four tile configurations, and the caller's stream takes the tail group itself
            blks = blocks_by_stage[li]
            hw = blks[0]['out_hw']
            planes = blks[0]['planes']
            g_pre = self.g_stage[li]          # gradient w.r.t. the (pre-ReLU-masked) output of the stage's last block
            g3, g2, g1 = [], [], []           # same-geometry weight gradients of this stage's blocks
            # image-split data-gradient chains (as in the forward pass; tuning key img_split_bwd): images [0, ceil(N/2)) on the caller's
            # stream, the rest on stream 3; the stage's weight gradients (whole batch) go out behind the JOIN at its end
            grp_all = GROUP and (li > 1 or GROUP_LAST)
            bsplit = bool(BB) and N >= 2 and grp_all and str(li + 1) in BSPLIT
            groups = [(0, (N + 1) // 2, 0), ((N + 1) // 2, N, BB)] if bsplit else [(0, N, 0)]
            if bsplit:
                ol.join(BB)          # whatever stream 3 still runs (laterals, an earlier scatter) is visible to the caller's stream ...
                ol.fork(BB)          # ... and stream 3 starts behind everything the caller's stream has queued
                self._br_pending = False
            for blk in reversed(blks):
                p = blk['prefix']
                c1, c2, c3 = cv[p + '.conv1'], cv[p + '.conv2'], cv[p + '.conv3']
                gA2 = self.buf(p + '.g_a2', N, hw[0], hw[1], planes)
                gA1 = self.buf(p + '.g_a1', N, hw[0], hw[1], planes)
                grp = GROUP and (li > 1 or GROUP_LAST)
                tsl = TAIL_SLOTS if li == 1 else 0      # last segment: nothing else is left to run beside these launches
                if bsplit:
                    g3.append(self._wgrad(ol, c3, g_pre, blk['a2'], N, [hw], [hw], side=SIDE, emit=False, slots=tsl))
                    g2.append(self._wgrad(ol, c2, gA2, blk['a1'], N, [hw], [hw], side=SIDE, emit=False, slots=tsl))
                    g_prev = self.buf(p + '.g_in', N, hw[0], hw[1], planes * 4) if blk['b'] > 0 else None
                    for g0, g1_, sd in groups:
                        wsf = br_ws if sd else (lambda d_: d_)
                        n_ = g1_ - g0
                        ol.conv(wsf(self._dgrad(c3.name, g_pre[g0:g1_], gA2[g0:g1_], n_, [hw], [hw], cs=c3.cout, cd=c3.cin, k=1, stride=1, pad=0,
                                                mask=blk['a2'][g0:g1_], mask_last=True)), side=sd)
                        ol.conv(wsf(self._dgrad(c2.name, gA2[g0:g1_], gA1[g0:g1_], n_, [hw], [hw], cs=c2.cout, cd=c2.cin, k=3, stride=1, pad=1,
                                                mask=blk['a1'][g0:g1_], mask_last=True)), side=sd)
                        if blk['b'] > 0:
                            ol.conv(wsf(self._dgrad(c1.name, gA1[g0:g1_], g_prev[g0:g1_], n_, [hw], [hw], cs=c1.cout, cd=c1.cin, k=1, stride=1,
                                                    pad=0, addend=g_pre[g0:g1_], mask=blk['xin'][g0:g1_], mask_last=True)), side=sd)
                        elif li > 1:      # stride-2 scatters into the previous stage's gradient: downsample path, then conv1
                            ds = cv[p + '.downsample.0']
                            tgt, ihw = self.g_stage[li - 1], blk['in_hw']
                            for spec, dy in ((ds, g_pre), (c1, gA1)):
                                ol.conv(wsf(self._dgrad(spec.name, dy[g0:g1_], tgt[g0:g1_], n_, [hw], [ihw], cs=spec.cout, cd=spec.cin, k=1,
                                                        stride=1, pad=0, os=2, addend=tgt[g0:g1_], mask=blk['xin'][g0:g1_], mask_first=True)), side=sd)
                    if blk['b'] > 0:
                        g1.append(self._wgrad(ol, c1, gA1, blk['xin'], N, [hw], [blk['in_hw']], side=SIDE, emit=False, slots=tsl))
                        g_pre = g_prev
                    else:
                        ol.join(BB)       # both chains are done: the weight gradients below (and the groups) read whole-batch tensors
                        d1 = self._wgrad(ol, c1, gA1, blk['xin'], N, [hw], [blk['in_hw']], side=SIDE, emit=True, slots=tsl)
                        dwd = self._wgrad(ol, cv[p + '.downsample.0'], g_pre, blk['xin'], N, [hw], [blk['in_hw']], side=SIDE, slots=tsl)
                        if li == 1:
                            self._pp['wg_c1'], self._pp['wg_ds'] = d1, dwd
                    continue
                ds_early = BB and blk['b'] == 0 and li > 1
                if ds_early:        # the downsample path's scatter into the previous stage's gradient, beside conv3 -> conv2 -> conv1
                    ds = cv[p + '.downsample.0']
                    tgt, ihw = self.g_stage[li - 1], blk['in_hw']
                    ol.fork(BB)          # (stream 3 is in order: the lateral that initialised tgt is already queued there)
                    ol.conv(br_ws(self._dgrad(ds.name, g_pre, tgt, N, [hw], [ihw], cs=ds.cout, cd=ds.cin, k=1, stride=1, pad=0,
                                              os=2, addend=tgt, mask=blk['xin'], mask_first=True)), side=BB)
                g3.append(self._wgrad(ol, c3, g_pre, blk['a2'], N, [hw], [hw], side=SIDE, emit=not grp, slots=tsl))
                ol.conv(self._dgrad(c3.name, g_pre, gA2, N, [hw], [hw], cs=c3.cout, cd=c3.cin, k=1, stride=1, pad=0,
                                    mask=blk['a2'], mask_last=True))
                g2.append(self._wgrad(ol, c2, gA2, blk['a1'], N, [hw], [hw], side=SIDE, emit=not grp, slots=tsl))
                ol.conv(self._dgrad(c2.name, gA2, gA1, N, [hw], [hw], cs=c2.cout, cd=c2.cin, k=3, stride=1, pad=1,
                                    mask=blk['a1'], mask_last=True))
                d1 = self._wgrad(ol, c1, gA1, blk['xin'], N, [hw], [blk['in_hw']], side=SIDE,
                                 emit=not (grp and blk['b'] > 0), slots=tsl)
                if li == 1 and blk['b'] == 0:
                    self._pp['wg_c1'] = d1
                if blk['b'] > 0:
                    g1.append(d1)
                if blk['b'] > 0:
                    g_prev = self.buf(p + '.g_in', N, hw[0], hw[1], planes * 4)
                    ol.conv(self._dgrad(c1.name, gA1, g_prev, N, [hw], [hw], cs=c1.cout, cd=c1.cin, k=1, stride=1,
                                        pad=0, addend=g_pre, mask=blk['xin'], mask_last=True))
                    g_pre = g_prev
                else:
                    ds = cv[p + '.downsample.0']
                    dwd = self._wgrad(ol, ds, g_pre, blk['xin'], N, [hw], [blk['in_hw']], side=SIDE, slots=tsl)
                    if li == 1:
                        self._pp['wg_ds'] = dwd
                    if li > 1:      # data gradient into the previous stage's output (stride-2 scatter)
                        tgt = self.g_stage[li - 1]
                        ihw = blk['in_hw']
                        if BB:          # the laterals / the early downsample scatter on stream 3 wrote tgt first
                            ol.join(BB)
                        for spec, dy in ((c1, gA1),) if ds_early else ((ds, g_pre), (c1, gA1)):
                            ol.conv(self._dgrad(spec.name, dy, tgt, N, [hw], [ihw], cs=spec.cout, cd=spec.cin, k=1,
                                                stride=1, pad=0, os=2, addend=tgt, mask=blk['xin'], mask_first=True))
            if li == PREFIX_LI:
                ol.record(L.SLOT_TAIL, stream=0)       # from here on the next step's frozen prefix may run beside this pass
            if GROUP and (li > 1 or GROUP_LAST):
                # last segment: the caller's stream has nothing left to do, it takes part of the groups itself
                on_main = '2' if li == 1 else '0'     # measured: tools/experiments_r2.txt (exp_r2w)
                order = sorted(((3, g3), (2, g2), (1, g1)), key=lambda t: str(t[0]) in on_main)     # side-stream groups first: one FORK
                for gi, grp_descs in order:
                    mine = str(gi) in on_main
                    self._wgrad_group(ol, grp_descs, side=SIDE and not mine, ws_name='wg_ws_main' if mine else 'wg_ws')
            self._flush_wgrads(ol, side=SIDE)
            seg = 4 - li                      # 1, 2, 3
            ol.record(seg)
            if li == 1:                       # last segment: everything must be complete when the list returns
                ol.join()
            # main=True: part of this bucket's gradients was computed on the caller's stream (the last group above)
            self.bwd_segments.append((ol, dict(bucket=buckets[seg], slot=seg, main=(li == 1))))
            if li == 1 and self.defer:
                # the towers' weight gradients, last on the weight-gradient stream and NOT joined: they run under whatever the caller
                # queues next (the next step's backbone forward); event slot 0 = "bucket 0 (head + FPN) complete"
                dl = OpList()
                on, self._multi_on = self._multi_on, False
                self._wgrad_group(dl, self._deferred_towers, side=True)
                self._multi_on = on
                dl.record(0)
                self.bwd_segments.append((dl, dict(bucket=buckets[0], slot=0, main=False, deferred=True)))

    # ---------------------------------------------------------------------------------------------
    def _split_prefix(self):
        """Pipelined frozen prefix (FCOS.pipeline_prefix): image layout + stem + pool + layer1 have frozen weights, so the next
        step's can run on its own stream from the moment the previous backward's data-gradient chain is done - beside the tail
        of the weight gradients, the log-variable ops and SGD, when the caller's stream has little to do.  The prefix is its
        own op list (run with that stream as the list's 'caller'); layer1's output has two buffers, alternating per step,
        because the previous step's last weight gradients (layer2.0's conv1 / downsample) still read theirs."""
        f, end = self.fwd, self._prefix_end
        pre, rest = OpList(), OpList()
        pre.items, rest.items = f.items[:end], f.items[end:]
        if 'prefix' in skip_items():      # step-level ablation (tools/step_ablation.sh): layer1 costs nothing - timing only
            pre.items = [o for o in pre.items if o.kind != L.OP_CONV]
        pre.keep = rest.keep = f.keep
        ws = torch.empty(32 << 20, dtype=torch.uint8, device=self.dev)      # the prefix runs beside the caller's convs: own split-K scratch
        for o in pre.items:
            if o.kind == L.OP_CONV and o.i[6] == 0:
                d = C.cast(o.desc, C.POINTER(L.ConvDesc)).contents
                if d.workspace:
                    d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel()
        pre.keep.append(ws)
        w0 = L.Op()
        w0.kind, w0.i[0], w0.i[1] = L.OP_WAIT, 0, L.SLOT_TAIL
        r0 = L.Op()
        r0.kind, r0.i[0], r0.i[1] = L.OP_RECORD, 0, L.SLOT_PREFIX
        pre.items = [w0] + pre.items + [r0]
        self._img_op += 1
        w1 = L.Op()
        w1.kind, w1.i[0], w1.i[1] = L.OP_WAIT, 0, L.SLOT_PREFIX
        self.fwd_inline = f                     # the whole forward pass as one list on the caller's stream (pipeline_prefix off)
        self.fwd_rest = rest
        rest.items = [w1] + rest.items
        self.prefix = pre
        out = self._pp['out']
        self._l1out = [out, torch.empty_like(out)]
        self._parity = 0

    def set_parity(self, p):
        """Points the five descriptors that touch layer1's output at buffer p."""
        ptr = self._l1out[p].data_ptr()
        self._pp['c3'].dst = ptr
        for d_, off in self._pp['c1'] + self._pp['ds']:
            if isinstance(d_, L.BneckDesc):
                d_.x = ptr + off
            else:
                d_.src = ptr + off
        self._pp['wg_c1'].x = ptr
        self._pp['wg_ds'].x = ptr
        self._parity = p

    def bind_image(self, img, half_last=False):
        """The batch for the next fwd.run(): a dense fp32 tensor on this device is read in place by the layout kernel (it
        must stay alive until that kernel has run - stream order on the caller's stream); anything else is staged in self.img.
        half_last: `img` holds N - 1 images; the plan's last image is the stem kernel's half-scale view of img[-1]
        (dsl_stem_pool_half; SemiEpochBasedRunner's scale-invariant copy, never materialised)."""
        assert img.shape[0] == self.N - (1 if half_last else 0), (tuple(img.shape), self.N, half_last)
        if half_last and self._stem_fused is False:
            raise RuntimeError('half_last needs the fused stem kernel (DSL_STEM_FUSED=0 is set)')
        direct = img.is_cuda and img.dtype == torch.float32 and img.is_contiguous() and img.device == self.img.device
        if not direct:
            self.img[:img.shape[0]].copy_(img, non_blocking=True)
        ptr = img.data_ptr() if direct else self.img.data_ptr()
        self._img_ref = img if direct else None
        lists = [(self.fwd, self._img_op - (1 if self.prefix is not None else 0))]
        if self.prefix is not None:
            lists.append((self.prefix, self._img_op))
        for f, idx in lists:
            if f.arr is None:
                f.arr = (L.Op * len(f.items))(*f.items)
            f.arr[idx].p[0] = ptr
            if self._stem_fused:
                f.arr[idx].i[4] = 1 if half_last else 0

    def fp8_warm(self, fwd):
        """Delayed fp8 scaling needs a recorded maximum per tensor: in front of a plan's FIRST forward pass the list runs twice (pass 1
        records the FPN outputs' maximum with scale 1, pass 2 the tower activations' under it); afterwards every step uses the maxima
        the step before it left."""
        if getattr(self, 'fp8_cold', False):
            self.fp8_cold = False
            fwd.run()
            fwd.run()

    def forward(self, img=None):
        if img is not None:
            self.bind_image(img)
        self.fp8_warm(self.fwd)
        self.fwd.run()


class Engine:
    """Caches plans per (store, N, H, W, training) and runs the student step / the teacher forward."""

    MAX_PLANS = 12      # multi-scale training (img_scale ranges) visits a few dozen padded shapes; a plan holds every
                        # activation of its shape (~1.5 GB per 800x1344 image), so keep the most recently used ones

    def __init__(self):
        self.plans = {}         # insertion-ordered: least recently used first

    def plan(self, store, N, H, W, training=True, single_stream=False):
        if store.dirty:
            store.refresh()           # in place where the packs exist; a re-allocation bumps store.generation
        key = (id(store), getattr(store, 'generation', 0), bool(getattr(store, 'defer_head', False)), single_stream, N, H, W, training)
        p = self.plans.pop(key, None)
        if p is None:
            stale = [k for k in self.plans if k[0] == id(store) and k[1] != key[1]]
            while stale or len(self.plans) >= self.MAX_PLANS:
                torch.cuda.synchronize()                    # the evicted plan's buffers may still be in use on a stream
                self.plans.pop(stale.pop() if stale else next(iter(self.plans)))
            p = Plan(store, N, H, W, training, single_stream=single_stream)
        self.plans[key] = p
        return p
