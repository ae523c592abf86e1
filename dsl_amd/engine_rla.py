"""Kernel lists of the RLA_ResNet backbone (reference: mmdet/models/backbones/resnet_rla.py:71-137 RLA_Bottleneck,
:289-327 RLA_ResNet._forward_impl, :343-388 freezing) for dsl_amd.engine.Plan.

Data layout.  A block's input cat(x, h) (:106) is ONE NHWC buffer "XH" of C + 128 channels = [x (C) | h (32) | zeros]:
the previous block's conv3 epilogue writes x = relu(bn3(.) + identity) into the first C channels (row stride C + 128),
its recurrent conv writes h behind them, and conv1 (weights stored with C + 128 input channels, zero columns behind the
real C + 32) reads the rows as an ordinary source - no concat copy.  Consumers of x alone (downsample, conv_out, the FPN
laterals, the ReLU masks and residual addends of the backward) address the same rows with that row stride.

What the reference computes, restated (pinned by tests/golden/rla_tiny.npz against the reference's own module):
  * style 'pytorch': the 3x3 strides (:84-86); conv1 is stride 1 on the block's input resolution;
  * `y = out` (:123) aliases the tensor that `out += identity` and the in-place ReLU then modify (:132-133), so the y that
    feeds conv_out is the block OUTPUT; the recurrent update is h = recurrent_conv(tanh(bn_b(h' + conv_out(out)))) with
    h' = avgpool2x2(h) in the first block of stages 1-3 (:129-130,314-321);
  * every BatchNorm runs in eval mode; those of stages 1-3 keep TRAINABLE affine parameters: forward folds them per step
    (ParamStore.refold_bn), backward gets (dgamma, dbeta) of conv -> BN pairs from the unscaled weight gradient
    (dsl_bn_wgrad_post) and of the recurrent path's BN from dsl_bn_tanh_bwd;
  * the h produced by the last block of the last stage is never used (:322-327): not computed.

Backward.  The gradient w.r.t. a block output `out` collects, UNMASKED, the next block's conv1 x-part data gradient, its
identity path (or downsample data gradient), the FPN lateral's (stage outputs) - and is masked by ReLU(out) in the epilogue
of conv_out's data gradient, which adds the recurrent path's contribution last.  conv_out / recurrent_conv are shared by
the blocks of a stage: their weight gradients are one grouped launch whose members are summed (dsl_wgrad_desc.shared).
"""
import ctypes as C

import torch

from . import _lib as L
from . import ops
from .params import RLA_C, RLA_PAD, STAGE_BLOCKS, STAGE_PLANES
from .tuning import tune, tune_int

BF = torch.bfloat16


def _rla_op(ol, kind, p=(), i=(), f=(), rows=0, side=False):
    d = L.RlaDesc()
    d.kind = kind
    for k, v in enumerate(p):
        d.p[k] = v if isinstance(v, int) else (0 if v is None else v.data_ptr())
    for k, v in enumerate(i):
        d.i[k] = int(v)
    for k, v in enumerate(f):
        d.f[k] = float(v)
    d.rows = int(rows)
    d._keep = [v for v in p if not isinstance(v, int)]
    ol._add(L.OP_RLA, d, i=(0, 0, 0, 0, 0, 0, int(side)))
    return d


def conv_out_hw(h, k, s, p):
    return (h + 2 * p - k) // s + 1


def build_forward(plan, emit_pool, h1, w1):
    """Stem (N, h1, w1, 64) -> max pool into the first XH (emit_pool(out, ld): the plan's stem / pool ops) -> 16 RLA blocks;
    fills plan.stage_out / stage_ld and plan.rla_blocks (what the backward needs)."""
    st, N, f = plan.store, plan.N, plan.fwd
    cv = st.convs
    h, w = conv_out_hw(h1, 3, 2, 1), conv_out_hw(w1, 3, 2, 1)
    cx = 64
    xh = plan.buf('rla.xh.0.0', N * h * w, cx + RLA_PAD, zero=True)
    emit_pool(xh, cx + RLA_PAD)
    plan.rla_blocks = []
    # image-split stages (as in Plan._fwd_resnet): the images of a batch are independent through the backbone - eval-mode
    # BatchNorms, per-image recurrent state - so the batch runs through these stages as two chains of half-batch launches on two
    # streams (tuning key rla_split: stage indices, default 1, 2, 3; the DSL iteration has N = 3: images [0, 2) and [2, 3))
    SPLIT = tune('rla_split') if (plan.BR and N >= 2 and plan.training) else ''
    TAIL = tune('rla_tail') != '0'
    split_open = False

    def br_ws(d_):
        d_.workspace, d_.workspace_bytes = L.ptr(plan.conv_ws_br), plan.conv_ws_br.numel()
        return d_

    def rows(t, g0, hw, ld):
        """Pointer to image g0's first row of a [N * hw][ld] bf16 tensor (or of a raw pointer into one)."""
        return (t if isinstance(t, int) else t.data_ptr()) + g0 * hw * ld * 2
    for s_, (planes, nb) in enumerate(zip(STAGE_PLANES, STAGE_BLOCKS)):
        c4 = planes * 4
        co, rc = cv[f'backbone.conv_outs.{s_}'], cv[f'backbone.recurrent_convs.{s_}']
        split = str(s_) in SPLIT
        if split and not split_open:
            f.fork(plan.BR)
            split_open = True
        elif split_open and not split:
            f.join(plan.BR)
            split_open = False
        groups = [(0, (N + 1) // 2, 0), ((N + 1) // 2, N, plan.BR)] if split else [(0, N, 0)]
        for b in range(nb):
            p = f'backbone.stages.{s_}.{b}'
            c1, c2, c3 = cv[p + '.conv1'], cv[p + '.conv2'], cv[p + '.conv3']
            stride = c2.stride
            oh, ow = conv_out_hw(h, 3, stride, 1), conv_out_hw(w, 3, stride, 1)
            ldx = cx + RLA_PAD
            a1 = plan.buf(p + '.a1', N * h * w, planes)
            a2 = plan.buf(p + '.a2', N * oh * ow, planes)
            last = s_ == 3 and b == nb - 1
            nxt = plan.buf(f'rla.xh.{s_}.{b + 1}', N * oh * ow, c4 + RLA_PAD, zero=True)      # the next block's XH
            idt = plan.buf(p + '.idt', N * oh * ow, c4) if b == 0 else None
            blk = dict(prefix=p, stage=s_, b=b, planes=planes, cx=cx, c4=c4, xh=xh, ldx=ldx, a1=a1, a2=a2, out=nxt,
                       ld_out=c4 + RLA_PAD, in_hw=(h, w), out_hw=(oh, ow), stride=stride, last=last, pooled=False)
            if not last:
                pooled = b == 0 and stride != 1
                hp = plan.buf(p + '.hpool', N * oh * ow, RLA_C) if pooled else None
                u = plan.buf(p + '.u', N * oh * ow, RLA_C)
                tw = rc.cin_store                              # 64 (frozen stage 0) / 128: T rows are as wide as the stored K
                t = plan.buf(p + '.t', N * oh * ow, tw, zero=True)
                bnn = f'backbone.stage_bns.{s_}.{b}'
                sc, bi = st.bn_ptrs(bnn)
                blk.update(u=u, t=t, tw=tw, bn=bnn, pooled=pooled)
            ihw, ohw = h * w, oh * ow
            for g0, g1, sd in groups:
                n_ = g1 - g0
                wsf = br_ws if sd else (lambda d_: d_)
                xh_g, a1_g, a2_g, nxt_g = rows(xh, g0, ihw, ldx), rows(a1, g0, ihw, planes), rows(a2, g0, ohw, planes), rows(nxt, g0, ohw, c4 + RLA_PAD)
                f.conv(wsf(plan._conv(c1, xh_g, a1_g, n_, [(h, w)], [(h, w)], relu=True, cs=ldx, lds=ldx)), side=sd)
                f.conv(wsf(plan._conv(c2, a1_g, a2_g, n_, [(h, w)], [(oh, ow)], relu=True)), side=sd)
                if b == 0:
                    idt_g, idt_ld = rows(idt, g0, ohw, c4), c4
                    f.conv(wsf(plan._conv(cv[p + '.downsample.0'], xh_g, idt_g, n_, [(h, w)], [(oh, ow)], cs=cx, lds=ldx)), side=sd)
                else:
                    idt_g, idt_ld = xh_g, ldx                  # the block input itself (x part of its rows)
                f.conv(wsf(plan._conv(c3, a2_g, nxt_g, n_, [(oh, ow)], [(oh, ow)], relu=True, addend=idt_g, lda=idt_ld,
                                      dst_ld=c4 + RLA_PAD)), side=sd)
                if not last:
                    h_ptr, h_ld = xh_g + cx * 2, ldx           # h part of this block's input rows
                    if blk['pooled']:
                        hp_g = rows(hp, g0, ohw, RLA_C)
                        _rla_op(f, L.RLA_AVGPOOL, p=(h_ptr, hp_g), i=(ldx, RLA_C, n_, h, w, RLA_C), side=sd)
                        h_ptr, h_ld = hp_g, RLA_C
                    u_g, t_g = rows(u, g0, ohw, RLA_C), rows(t, g0, ohw, tw)
                    if TAIL:
                        # conv_out -> BN + tanh -> recurrent 3x3 as ONE launch (round 6, csrc/rla.hip rla_tail_fwd_kernel): the chain the
                        # next block's conv1 waits for is one kernel boundary long instead of three
                        _rla_op(f, L.RLA_TAIL_FWD, p=(nxt_g, h_ptr, st.w16_ptr(co), sc, bi, st.w16_ptr(rc), u_g, t_g, nxt_g + c4 * 2),
                                i=(c4 + RLA_PAD, h_ld, c4, tw, c4 + RLA_PAD, n_, oh, ow), side=sd)
                        continue
                    f.conv(wsf(plan._conv(co, nxt_g, u_g, n_, [(oh, ow)], [(oh, ow)], cs=c4, lds=c4 + RLA_PAD, addend=h_ptr, lda=h_ld,
                                          dst_ld=RLA_C, affine=False)), side=sd)
                    _rla_op(f, L.RLA_BN_TANH, p=(u_g, sc, bi, t_g), i=(RLA_C, tw, RLA_C), rows=n_ * ohw, side=sd)
                    f.conv(wsf(plan._conv(rc, t_g, nxt_g + c4 * 2, n_, [(oh, ow)], [(oh, ow)], cs=tw, dst_ld=c4 + RLA_PAD,
                                          affine=False)), side=sd)
            plan.rla_blocks.append(blk)
            xh, cx, h, w = nxt, c4, oh, ow
        plan.stage_out.append((xh, (h, w)))
        plan.stage_ld.append(c4 + RLA_PAD)
    if split_open:
        f.join(plan.BR)
    plan._keep_rla = getattr(plan, '_keep_rla', [])


def build_backward(plan, buckets, SIDE):
    """Backward segments of stages 3, 2, 1 (plan.bwd_segments already holds head + FPN)."""
    st, N = plan.store, plan.N
    cv = st.convs
    g32 = lambda name: st.t32_ptr(name, st.grad)
    bn_g = lambda bn, leaf: st.t32_ptr('bn_train.' + leaf, st.grad) + st.bn_train_off[bn][0] * 4
    bn_p = lambda bn, leaf: st.t32_ptr('bn_train.' + leaf) + st.bn_train_off[bn][0] * 4
    bn_f = lambda bn, leaf: st.frozen.data_ptr() + (st.frozen_regions['bn_train.' + leaf][0] + st.bn_train_off[bn][0]) * 4
    from .engine import OpList
    # Image-split data-gradient chains (as in the forward pass and in Plan's ResNet backward): the images of a batch are
    # independent through the backbone's backward too - per-image recurrent state, eval-mode BatchNorms - so the chain of a stage
    # runs as two chains of part-batch launches, images [0, ceil(N/2)) on the caller's stream and the rest on stream 3
    # (BSPLIT: stage indices).  What sums over the batch waits behind the JOIN at the stage's end: the weight gradients
    # (deferred into the stage's grouped / multi launches anyway) and the (dgamma, dbeta) of the recurrent path's BatchNorms,
    # whose block records the two chains write side by side (dsl_rec_sum_multi, one launch per stage).
    # OFF by default - measured on the DSL iteration (N = 3: a 2 + 1 split; tools/exp_env.sh, three alternations in one box):
    # '' 11.99 / 12.00 / 12.12 ms, '123' 12.17 / 12.16 / 12.15, '23' 12.07 / 12.13 / 12.01: unlike the forward chains and the
    # ResNet engine's backward, the caller's stream is full of kernel time here (10.7 of 12.1 ms busy, profiles/r03_rla_timeline.txt),
    # not of launch gaps, and the weight-gradient grids already hold the CUs a second chain would use.
    TAIL = tune('rla_tail') != '0'
    BSPLIT = tune('rla_split_bwd')        # default '': built, same gradients (test_rla_image_split_backward_chains_give_the_same_gradients), slower - LAB_NOTES.md
    S2_CLASSES = True         # stride-2 3x3 data gradients as four parity-class launches
    BB = plan.BR

    def br_ws(d_):
        d_.workspace, d_.workspace_bytes = L.ptr(plan.conv_ws_br), plan.conv_ws_br.numel()
        return d_
    gh_next = None            # gradient w.r.t. the h a block PRODUCES ([px_out][64], 32 real), written by its consumer
    for s_ in (3, 2, 1):
        ol = OpList()
        blks = [b for b in plan.rla_blocks if b['stage'] == s_]
        planes, c4 = blks[0]['planes'], blks[0]['c4']
        co, rc = cv[f'backbone.conv_outs.{s_}'], cv[f'backbone.recurrent_convs.{s_}']
        gx = plan.g_stage[s_]                 # gradient w.r.t. the stage's last output (FPN lateral + next stage), unmasked for s < 3
        g3, g2, g1, g_co, g_rc, post = [], [], [], [], [], []
        bsplit = str(s_) in BSPLIT
        groups = [(0, (N + 1) // 2, 0), ((N + 1) // 2, N, BB)] if bsplit else [(0, N, 0)]
        rec_items = []
        # the last stage's weight gradients go out behind the whole data-gradient chain, with nothing left to run beside them (0.8 ms
        # of the iteration, profiles/r03_rla_sequence.txt): they take the chip instead of the weight-gradient stream's usual budget
        tsl = tune_int('tail_slots') if s_ == 1 else 0
        if bsplit:
            ol.fork(BB)

        def emit(d):            # a single weight gradient: with the stage's multi launch (Plan._flush_wgrads) or on its own
            if plan._multi_on and SIDE:
                plan._wg_pending.append([d])
            else:
                ol.wgrad(d, side=SIDE)
        for blk in reversed(blks):
            p, b = blk['prefix'], blk['b']
            c1, c2, c3 = cv[p + '.conv1'], cv[p + '.conv2'], cv[p + '.conv3']
            (h, w), (oh, ow) = blk['in_hw'], blk['out_hw']
            cx, ldx, stride = blk['cx'], blk['ldx'], blk['stride']
            pin, pout = N * h * w, N * oh * ow
            ihw, ohw = h * w, oh * ow
            first = s_ == 1 and b == 0                       # input comes from the frozen stage 0: no data gradient
            rows2d = lambda t_: t_.view(-1, t_.shape[-1])
            I = lambda t_, b_, e_: None if t_ is None else rows2d(t_)[b_ * ihw:e_ * ihw]      # rows of images [b_, e_) at the input resolution
            O = lambda t_, b_, e_: None if t_ is None else rows2d(t_)[b_ * ohw:e_ * ohw]      # ... at the output resolution

            def dg(sd, d_):
                ol.conv(br_ws(d_) if sd else d_, side=sd)
            if blk['last']:
                g_pre = gx                                   # masked by the lateral's data gradient already
                g_u = None
            else:
                # ---- recurrent path: h_out = recurrent_conv(t), t = tanh(bn(u)), u = h' + conv_out(out)
                gh = gh_next
                g_t = plan.buf(p + '.g_t', pout, RLA_C)
                g_u = plan.buf(p + '.g_u', pout, 64, zero=True)
                g_pre = plan.buf(p + '.g_pre', pout, c4)
                # block records per group: the kernel's own rows-per-block constant decides (rla.hip BT_ROWS), through the ABI
                nrec = [int(L.lib.dsl_bn_tanh_bwd_workspace_bytes((e_ - b_) * ohw, RLA_C)) // (2 * RLA_C * 4) for b_, e_, _ in groups]
                ws = plan.buf(p + '.bnws', sum(nrec) * 2 * RLA_C + 8, dtype=torch.float32)
                sc, _ = st.bn_ptrs(blk['bn'])
                rec0 = 0
                for gi, (g0, ge, sd) in enumerate(groups):
                    n_ = ge - g0
                    whole = len(groups) == 1
                    if TAIL and whole:
                        # recurrent-conv data gradient + BN / tanh backward as ONE launch (round 6, csrc/rla.hip rla_tail_bwd_kernel):
                        # g_t never leaves the chip, the (dgamma, dbeta) records are per 14 x 14 tile
                        ws2 = plan.buf(p + '.bnws2', int(L.lib.dsl_rla_tail_bwd_workspace_bytes(n_, oh, ow)) // 4, dtype=torch.float32)
                        _rla_op(ol, L.RLA_TAIL_BWD,
                                p=(O(gh, g0, ge), st.wT_ptr(rc.name), O(blk['t'], g0, ge), O(blk['u'], g0, ge), sc, bn_f(blk['bn'], 'running_mean'),
                                   bn_f(blk['bn'], 'running_var'), O(g_u, g0, ge), bn_g(blk['bn'], 'weight'), bn_g(blk['bn'], 'bias'), ws2),
                                i=(64, rc.cout_pad, blk['tw'], 64, n_, oh, ow), f=(1e-5,), side=sd)
                        dg(sd, plan._dgrad(co.name, O(g_u, g0, ge), O(g_pre, g0, ge), n_, [(oh, ow)], [(oh, ow)], cs=64, cd=c4, k=1,
                                           stride=1, pad=0, addend=O(gx, g0, ge), mask=O(blk['out'], g0, ge), ldm=blk['ld_out'],
                                           mask_last=True, cs_real=RLA_C))
                        continue
                    dg(sd, plan._dgrad(rc.name, O(gh, g0, ge), O(g_t, g0, ge), n_, [(oh, ow)], [(oh, ow)], cs=64, cd=RLA_C,
                                       cd_pad=rc.cin_store, k=3, stride=1, pad=1, ldd=RLA_C))
                    _rla_op(ol, L.RLA_BN_TANH_BWD,
                            p=(O(g_t, g0, ge), O(blk['t'], g0, ge), O(blk['u'], g0, ge), sc, bn_f(blk['bn'], 'running_mean'),
                               bn_f(blk['bn'], 'running_var'), O(g_u, g0, ge), bn_g(blk['bn'], 'weight') if whole else None,
                               bn_g(blk['bn'], 'bias') if whole else None, ws.data_ptr() + rec0 * 2 * RLA_C * 4),
                            i=(RLA_C, blk['tw'], RLA_C, 64, RLA_C), f=(1e-5,), rows=n_ * ohw, side=sd)
                    rec0 += nrec[gi]
                    # g_pre = (gx + conv_out^T g_u) * [out > 0]: every contribution to d/d(out) has arrived, mask once
                    dg(sd, plan._dgrad(co.name, O(g_u, g0, ge), O(g_pre, g0, ge), n_, [(oh, ow)], [(oh, ow)], cs=64, cd=c4, k=1,
                                       stride=1, pad=0, addend=O(gx, g0, ge), mask=O(blk['out'], g0, ge), ldm=blk['ld_out'],
                                       mask_last=True, cs_real=RLA_C))
                if len(groups) > 1:
                    it = L.RecSumItem()
                    it.rec, it.out_a, it.out_b, it.nrec = ws.data_ptr(), bn_g(blk['bn'], 'weight'), bn_g(blk['bn'], 'bias'), sum(nrec)
                    rec_items.append(it)
                g_rc.append(plan._wgrad(ol, rc, gh, blk['t'], N, [(oh, ow)], [(oh, ow)], cy=64, cd=RLA_C, emit=False, shared=1, slots=tsl))
                g_co.append(plan._wgrad(ol, co, g_u, blk['out'], N, [(oh, ow)], [(oh, ow)], cy=64, cd=RLA_C, emit=False,
                                        ldx=blk['ld_out'], shared=1, slots=tsl))
            # ---- bottleneck: out = relu(bn3(conv3(a2)) + identity)
            gA2 = plan.buf(p + '.g_a2', pout, planes)
            gA1 = plan.buf(p + '.g_a1', pin, planes)
            if not first:
                wT = st.wT_ptr(c1.name)                  # CRSK rows = input channels of conv1: [x (cx) | h (32) | zeros]
                gx_prev = plan.buf(p + '.g_in', pin, cx) if b > 0 else plan.g_stage[s_ - 1]      # (holds the FPN lateral's contribution)
                gh_prev = plan.buf(p + '.g_hin', pin, 64, zero=True)
                add_buf = plan.buf(p + '.g_hpool', pin, 64, zero=True) if (g_u is not None and blk['pooled']) else None
            for g0, ge, sd in groups:
                n_ = ge - g0
                dg(sd, plan._dgrad(c3.name, O(g_pre, g0, ge), O(gA2, g0, ge), n_, [(oh, ow)], [(oh, ow)], cs=c4, cd=planes, k=1,
                                   stride=1, pad=0, mask=O(blk['a2'], g0, ge), mask_last=True))
                if stride == 2 and S2_CLASSES:
                    # the stride-2 3x3 data gradient as four stride-1 launches of the pipelined kernel (one per output-pixel parity
                    # class, ops.dgrad_s2_descs) instead of the general strided gather: 13 instead of 36 tap products per 2 x 2 block
                    packs = {(py, px): st.wT_ptr(f'{c2.name}#s2{py}{px}') for py in (0, 1) for px in (0, 1)}
                    for d_ in ops.dgrad_s2_descs(O(gA2, g0, ge), packs, I(gA1, g0, ge), n=n_, dy_hw=(oh, ow), dst_hw=(h, w), cs=planes,
                                                 cd=planes, mask=I(blk['a1'], g0, ge), ldm=planes, flags=L.CONV_MASK_LAST,
                                                 workspace=plan.conv_ws):
                        dg(sd, d_)
                else:
                    dg(sd, plan._dgrad(c2.name, O(gA2, g0, ge), I(gA1, g0, ge), n_, [(oh, ow)], [(h, w)], cs=planes, cd=planes, k=3,
                                       stride=stride, pad=1, mask=I(blk['a1'], g0, ge), mask_last=True))
                if first:
                    continue
                # ---- gradient w.r.t. the block input (x part) and w.r.t. the incoming h
                if b > 0:
                    dg(sd, plan._dgrad(c1.name, I(gA1, g0, ge), I(gx_prev, g0, ge), n_, [(h, w)], [(h, w)], cs=planes, cd=cx, k=1,
                                       stride=1, pad=0, addend=I(g_pre, g0, ge)))      # + the identity path; masked by the consumer of gx_prev
                else:
                    ds = cv[p + '.downsample.0']
                    dg(sd, plan._dgrad(ds.name, O(g_pre, g0, ge), I(gx_prev, g0, ge), n_, [(oh, ow)], [(h, w)], cs=c4, cd=cx, k=1,
                                       stride=1, pad=0, os=stride, addend=I(gx_prev, g0, ge)))
                    dg(sd, plan._dgrad(c1.name, I(gA1, g0, ge), I(gx_prev, g0, ge), n_, [(h, w)], [(h, w)], cs=planes, cd=cx, k=1,
                                       stride=1, pad=0, addend=I(gx_prev, g0, ge)))
                # h part: rows [cx, cx + 64) of the pack; + the recurrent path's own h' term
                add, lda = None, None
                if g_u is not None:
                    if blk['pooled']:
                        add = I(add_buf, g0, ge)
                        _rla_op(ol, L.RLA_AVGPOOL_BWD, p=(O(g_u, g0, ge), add), i=(64, 64, n_, h, w, RLA_C), side=sd)
                    else:
                        add = I(g_u, g0, ge)                 # (no pooling: input and output resolutions agree)
                    lda = 64
                dg(sd, plan._dgrad(c1.name, I(gA1, g0, ge), I(gh_prev, g0, ge), n_, [(h, w)], [(h, w)], cs=planes, cd=RLA_C, cd_pad=64,
                                   k=1, stride=1, pad=0, ldd=64, addend=add, lda=lda, wptr=wT + cx * c1.cout_pad * 2))
            g3.append(plan._wgrad(ol, c3, g_pre, blk['a2'], N, [(oh, ow)], [(oh, ow)], emit=False, raw=True,
                                  db_ptr=bn_g(c3.bn, 'bias'), slots=tsl))
            d2 = plan._wgrad(ol, c2, gA2, blk['a1'], N, [(oh, ow)], [(h, w)], emit=False, raw=True, db_ptr=bn_g(c2.bn, 'bias'), slots=tsl)
            if stride == 1:
                g2.append(d2)                # the stage's stride-1 3x3 convolutions share a geometry
            else:
                emit(d2)
            d1 = plan._wgrad(ol, c1, gA1, blk['xh'], N, [(h, w)], [(h, w)], emit=False, raw=True, db_ptr=bn_g(c1.bn, 'bias'), slots=tsl)
            if b > 0:
                g1.append(d1)
            else:
                emit(d1)
                ds = cv[p + '.downsample.0']
                emit(plan._wgrad(ol, ds, g_pre, blk['xh'], N, [(oh, ow)], [(h, w)], emit=False, raw=True,
                                 db_ptr=bn_g(ds.bn, 'bias'), ldx=ldx, slots=tsl))
            if not first:
                gh_next = gh_prev
                gx = gx_prev
            post += [c1, c2, c3] + ([cv[p + '.downsample.0']] if b == 0 else [])
        if bsplit:
            ol.join(BB)           # both chains are done: everything below reads whole-batch tensors
            if rec_items:
                arr = (L.RecSumItem * len(rec_items))(*rec_items)
                tab = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).clone().to(plan.dev)
                plan.bufs[f'rla.recsum.{s_}'] = tab
                _rla_op(ol, L.RLA_REC_SUM, p=(tab,), i=(len(rec_items), RLA_C))
        # ---- the stage's weight gradients: grouped where the geometry is shared, then the BatchNorm post-pass
        post_specs = post
        for grp in (g3, g2, g1):
            plan._wgrad_group(ol, grp, side=SIDE)
        for grp in (g_co, g_rc):
            if grp:
                plan._wgrad_group(ol, grp, side=SIDE, ws_name='wg_ws_shared')
        plan._flush_wgrads(ol, side=SIDE)         # everything deferred above: one multi launch per tile configuration
        items, rows = [], 0
        for spec in post_specs:
            it = L.BnPostItem()
            it.w, it.dw = st.t32_ptr(spec.name + '.weight'), g32(spec.name + '.weight')
            it.dgamma, it.dbeta, it.gamma = bn_g(spec.bn, 'weight'), bn_g(spec.bn, 'bias'), bn_p(spec.bn, 'weight')
            it.mean, it.var = bn_f(spec.bn, 'running_mean'), bn_f(spec.bn, 'running_var')
            it.rows, it.k, it.row_start = spec.cout, spec.k * spec.k * spec.cin_store, rows
            rows += spec.cout
            items.append(it)
        arr = (L.BnPostItem * len(items))(*items)
        tab = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).clone().to(plan.dev)
        plan.bufs[f'rla.bnpost.{s_}'] = tab
        # behind the weight gradients on their stream: dgamma from <W, dWu>, then the rows are scaled in place
        _rla_op(ol, L.RLA_BN_POST, p=(tab,), i=(len(items), rows), f=(1e-5,), side=SIDE)
        seg = 4 - s_
        ol.record(seg)
        if s_ == 1:
            ol.join()
        plan.bwd_segments.append((ol, dict(bucket=buckets[seg], slot=seg, main=False)))
