"""Fused SGD over the flat parameter buffer (dsl_sgd_step): momentum, weight decay, the reference's
paramwise rules (bias_lr_mult / bias_decay_mult), gradient-norm clipping and the bf16 re-pack in one
pass.  Replaces torch.optim.SGD + mmcv OptimizerHook's clip_grad_norm_
(mmdet/apis/train.py:111,157-166; configs/fcos_semi/*.py `optimizer`, `optimizer_config`)."""
import ctypes as C
import os

import torch

from . import _lib as L
from .registry import OPTIMIZERS


from .tuning import skip_items

_PACK_SIDE = True                          # data-gradient weight packs off the caller's stream (measured: tools/experiments_r2.txt (exp_r2l))


@OPTIMIZERS.register_module(name='SGD')
class FlatSGD:
    def __init__(self, model, lr=0.01, momentum=0.9, weight_decay=0.0, paramwise_cfg=None, grad_clip=None,
                 defer_head_update=False, late_exchange=True, **kw):
        assert not kw.get('nesterov', False) and kw.get('dampening', 0) == 0
        self.model = model
        self.store = model.store
        pw = paramwise_cfg or {}
        self.bias_lr_mult = float(pw.get('bias_lr_mult', 1.0))
        self.bias_decay_mult = float(pw.get('bias_decay_mult', 1.0))
        self.momentum, self.weight_decay = float(momentum), float(weight_decay)
        self.max_norm = None
        if grad_clip:
            assert grad_clip.get('norm_type', 2) == 2
            self.max_norm = float(grad_clip['max_norm'])
        # two groups so that LR schedulers written against torch optimizers keep working
        self.param_groups = [dict(lr=lr, initial_lr=lr, name='weights'),
                             dict(lr=lr * self.bias_lr_mult, initial_lr=lr * self.bias_lr_mult, name='conv_bias')]
        self.momentum_buf = None
        self.gnorm_sq = None
        self.steps = 0
        self.late_exchange = bool(late_exchange)      # data parallel without clipping: collectives behind the backward pass (see _sync_defer)
        self.defer_head_update = bool(defer_head_update)      # opt-in (measured slower on one GPU, LAB_NOTES: deferred head update)
        self._sync_defer()

    def _sync_defer(self):
        """Deferred head update (engine.Plan.defer): possible when the update is element-wise per bucket - no gradient
        clipping (a global norm needs every gradient first), per-bucket steps on, packs on the side stream.  The flag lives on
        the parameter store because it shapes the op lists; opt-in (constructor argument defer_head_update)."""
        want = (self.max_norm is None and _PACK_SIDE and self.defer_head_update
                and self.store.backbone != 'rla')
        # Late exchange (data parallel): with an element-wise per-bucket update nothing at the end of a step has to
        # wait for the collectives - the detector queues them behind the backward pass and the next forward pass waits stage by stage
        self.model.late_exchange = bool(self.max_norm is None and _PACK_SIDE and self.late_exchange and self.store.backbone != 'rla')
        if bool(getattr(self.store, 'defer_head', False)) != want:
            self.store.wait_pending() if self.store.train.is_cuda else None
            self.store.defer_head = want        # plans are keyed by it (Engine.plan): lists built the other way are not reused

    def zero_grad(self, set_to_none=False):
        # every gradient element is overwritten by the backward kernels; nothing to clear
        return

    def _adopt_loaded_state(self):
        """A momentum tensor restored by load_state_dict (CPU tensor, possibly laid out by another build) -> a buffer laid out like this
        model's parameter buffer.  Called by step(); separate so that it can be checked without a device."""
        st = self.store
        # state restored from a checkpoint (CPU tensor, possibly laid out by another build): move it into a buffer laid out like
        # the parameter buffer, REGION BY REGION (name -> offset, size, saved with the state), and keep the step count (resume
        # must not re-zero momentum).  A prefix copy is only right when the two layouts differ by trailing padding; a state
        # without a region table whose size is off by more than that is refused instead of being attached to the wrong parameters.
        src = self.momentum_buf.reshape(-1)
        buf = torch.zeros_like(st.train)
        regs = getattr(self, '_loaded_regions', None)
        cur = {k: (int(v[0]), int(v[1])) for k, v in st.train_regions.items()}
        if regs:
            regs = {k: (int(v[0]), int(v[1])) for k, v in regs.items()}
            missing = [k for k in cur if k not in regs]
            if missing:
                raise RuntimeError(f'optimizer state: no momentum for {len(missing)} parameter regions of this model (first: {missing[:3]})')
            for k, (o, n) in cur.items():
                so, sn = regs[k]
                m = min(n, sn)          # (a region's tail is tile padding: zeros in both)
                buf.view(-1)[o:o + m].copy_(src[so:so + m].to(device=st.device, dtype=buf.dtype))
        else:
            if abs(buf.numel() - src.numel()) > 4096:
                raise RuntimeError(f'optimizer state: momentum has {src.numel()} elements, the parameter buffer {buf.numel()}, and the '
                                   'state carries no region table to remap it by (saved by an older build with another layout?)')
            n = min(buf.numel(), src.numel())
            buf.view(-1)[:n].copy_(src[:n].to(device=st.device, dtype=buf.dtype))
        self.momentum_buf = buf
        self._loaded_regions = None

    def step(self):
        st = self.store
        fresh = False
        if self.momentum_buf is None:
            # (zeros, queued on the CALLER's stream: the per-bucket path below runs on other streams that are not ordered behind the
            # caller's - `fresh` makes them wait for this fill.  Round 4: without that wait the fill could land AFTER the first
            # bucket's update had written its momentum - that bucket then started its second step from zero momentum; which buckets,
            # if any, depended on how far the GPU lagged behind the host: found as a flaky bit-exactness test once the optimizer's
            # stream was shared between model instances)
            self.momentum_buf = torch.zeros_like(st.train)
            self.steps = 0
            fresh = True
        elif (self.momentum_buf.device != st.device or self.momentum_buf.shape != st.train.shape
              or getattr(self, '_loaded_regions', None) is not None):
            self._adopt_loaded_state()
            fresh = True          # (copied into place on the caller's stream just now: the same ordering as for the zero fill)
        if self.gnorm_sq is None or self.gnorm_sq.device != st.device:
            self.gnorm_sq = torch.zeros(1, device=st.device)
        sp = L.stream_ptr()
        lr = float(self.param_groups[0]['lr'])
        blr = float(self.param_groups[1]['lr']) / lr if lr != 0 else self.bias_lr_mult
        self._sync_defer()          # (OptimizerHook may set max_norm after construction)
        infos = getattr(self.model, '_last_bwd_infos', None)
        infos = [i for i in infos if i['bucket'] is not None] if infos else infos      # completion order; a deferred bucket comes last
        if self.max_norm is None and infos and st.grad.is_cuda and all(i['bucket'][0] % 4 == 0 for i in infos):
            # No gradient clipping (the supervised config): the update is element-wise, so each gradient bucket - head + FPN,
            # layer4, layer3, layer2, in the order the backward pass completes them - is updated on the optimizer's own stream
            # as soon as its weight gradients (data parallel: its all-reduce) are done, beside the rest of the backward pass,
            # instead of one pass over all 32 M parameters behind the last weight gradient.  Same arithmetic, same bits.
            cur = torch.cuda.current_stream()
            deferred = any(i.get('deferred') for i in infos)
            if deferred:
                # Deferred head update: the buckets the next forward pass needs first are updated on the CALLER's stream (idle once
                # the data-gradient chain is through), the deferred bucket on the weight-gradient stream itself, in order behind the
                # towers' group.  A stream of the optimizer's own would do logically, but streams share four hardware queues: it
                # landed on the weight-gradient stream's queue, and the three early updates - hence the whole next forward pass -
                # then sat behind the deferred 0.5 ms group (measured: 401 instead of 430 img/s, profiles/r04_defer_first.txt).
                if getattr(self, '_side1', None) is None:
                    h = C.c_void_p()
                    L.check(L.lib.dsl_side_stream(1, C.byref(h)), 'dsl_side_stream')
                    self._side1 = torch.cuda.ExternalStream(h.value)
                os_ = None
            else:
                if getattr(self, '_opt_stream', None) is None:
                    from .detectors import role_stream          # one optimizer stream per device and process (hardware queues)
                    self._opt_stream = role_stream('optimizer', st.device)
                os_ = self._opt_stream
            pend = list(getattr(self.model, '_pending', []) or [])
            # late exchange: the detector queued no collective - each is asked for here, in front of its bucket's update
            late = bool(getattr(self.model, '_late_todo', None)) and not deferred and not pend
            seq = list(enumerate(infos[::-1] if late else infos))
            if fresh:
                for s_ in ([self._side1] if deferred else [os_]):
                    s_.wait_stream(cur)          # the momentum buffer's zero fill (see above)
            for k, info in seq:
                lo, hi = info['bucket']
                tgt = (self._side1 if info.get('deferred') else cur) if deferred else os_
                tp = C.c_void_p(tgt.cuda_stream)
                if late:
                    got = self.model.exchange_late()
                    assert got is not None and got[0] is info, 'late exchange: bucket order'
                    pend.append(got[1])
                if pend:
                    with torch.cuda.stream(tgt):
                        pend[k].wait()
                        tr = getattr(self.model, 'comm_trace', None)
                        if tr and k < len(tr[-1]['buckets']):
                            ed = torch.cuda.Event(enable_timing=True)
                            ed.record()
                            tr[-1]['buckets'][k]['done'] = ed
                elif not (deferred and info.get('deferred')):          # (the deferred bucket's stream IS the one its gradients ran on)
                    never = L.lib.dsl_stream_wait_slot(int(info['slot']), tp)
                    # (data parallel with the exchanges already waited for on the caller's stream - FCOS.wait_grads: the update must
                    #  follow them, not only its weight gradients)
                    if (info['main'] or never != 0 or getattr(self.model, 'world_size', 1) > 1) and tgt is not cur:
                        tgt.wait_stream(cur)
                o4, o2, o1 = lo * 4, lo * 2, lo
                if 'sgd' in skip_items():          # step-level ablation (tools/step_ablation.sh): timing only
                    continue
                L.check(L.lib.dsl_sgd_step(C.c_void_p(st.train.data_ptr() + o4), C.c_void_p(st.grad.data_ptr() + o4),
                                           C.c_void_p(self.momentum_buf.data_ptr() + o4), C.c_void_p(st.train16.data_ptr() + o2),
                                           C.c_void_p(st.group.data_ptr() + o1), hi - lo, lr, self.momentum, self.weight_decay, blr,
                                           self.bias_decay_mult, None, 0.0, int(self.steps == 0), tp), 'dsl_sgd_step')
                if late:          # "gradient bucket <slot> is updated": what the next forward pass waits for in front of that stage
                    L.check(L.lib.dsl_stream_record_slot(L.SLOT_UPD + int(info['slot']), tp), 'dsl_stream_record_slot')
            if hasattr(self.model, '_pending'):
                self.model._pending = []
            if deferred:
                s1p = C.c_void_p(self._side1.cuda_stream)
                L.check(L.lib.dsl_stream_record_slot(L.SLOT_HEADW, s1p), 'dsl_stream_record_slot')
                st._pending_ev = torch.cuda.Event()
                st._pending_ev.record(self._side1)
                # the data-gradient packs read every bucket: behind the caller's stream (its three updates) AND the deferred one,
                # i.e. forked from the caller's stream onto the weight-gradient stream, in order behind the deferred update
                st.repack_dgrad(sp, side=True)
            elif late:
                # nothing on the caller's stream waits: the next forward pass waits per stage (SLOT_UPD), every other reader of the
                # parameters calls ParamStore.wait_pending; the data-gradient packs follow the last update on the optimizer's stream
                # (= side stream 1, where repack_dgrad(side=True) queues them) and mark SLOT_PACKS for the next backward pass
                st._pending_ev = torch.cuda.Event()
                st._pending_ev.record(os_)
                st.repack_dgrad(sp, side=True)
            else:
                cur.wait_stream(os_)
                st.repack_dgrad(sp, side=_PACK_SIDE)
            self.steps += 1
            return
        if hasattr(self.model, 'wait_grads'):
            self.model.wait_grads()
        for i in infos or []:
            if i.get('deferred'):        # a list built for the deferred head update left its last weight gradients unjoined
                L.lib.dsl_stream_wait_slot(int(i['slot']), sp)
        gptr = None
        if self.max_norm is not None:
            m = self.model
            if getattr(m, 'world_size', 1) > 1 and st.grad.is_cuda and getattr(m, 'clip_partials', None) is None and hasattr(m, 'clip_partials'):
                # data parallel + clipping: from the next backward pass on, the norm arrives in pieces - one partial sum per bucket,
                # computed on the communication stream behind that bucket's all-reduce - and only their fold precedes the update
                m.clip_partials = torch.zeros(8 * L.SUMSQ_PARTS, device=st.device)
            if getattr(m, '_partials_valid', False):
                L.check(L.lib.dsl_sumsq_fold(L.ptr(m.clip_partials), int(m._n_partials), L.ptr(self.gnorm_sq), sp), 'dsl_sumsq_fold')
                m._partials_valid = False
            else:
                if getattr(self, '_sumsq_ws', None) is None or self._sumsq_ws.device != st.device:
                    self._sumsq_ws = torch.zeros(1024, device=st.device)
                L.check(L.lib.dsl_sumsq_det(L.ptr(st.grad), st.n_train, L.ptr(self.gnorm_sq), L.ptr(self._sumsq_ws), sp), 'dsl_sumsq_det')
            gptr = self.gnorm_sq
        L.check(L.lib.dsl_sgd_step(L.ptr(st.train), L.ptr(st.grad), L.ptr(self.momentum_buf), L.ptr(st.train16),
                                   L.ptr(st.group), st.n_train, lr, self.momentum, self.weight_decay, blr,
                                   self.bias_decay_mult, L.ptr(gptr), self.max_norm or 0.0, int(self.steps == 0), sp),
                'dsl_sgd_step')
        st.repack_dgrad(sp, side=_PACK_SIDE)
        self.steps += 1

    def state_dict(self):
        # regions: where every named parameter lives in the flat momentum buffer - what load_state_dict / step remap by
        return dict(momentum=self.momentum_buf, steps=self.steps, param_groups=self.param_groups,
                    regions={k: (int(v[0]), int(v[1])) for k, v in self.store.train_regions.items()})

    def load_state_dict(self, sd):
        """Restores momentum, the step count (first-step rule of torch.optim.SGD: buf = grad) and the groups' learning
        rates.  The momentum tensor may live on the CPU (runner.resume loads with map_location='cpu'); step() moves it."""
        m = sd['momentum']
        self.momentum_buf = m.detach().clone() if isinstance(m, torch.Tensor) else None
        self.steps = int(sd['steps']) if self.momentum_buf is not None else 0
        self.param_groups = [dict(g) for g in sd['param_groups']]
        self._loaded_regions = dict(sd['regions']) if sd.get('regions') else None
        if self._loaded_regions is None and self.momentum_buf is not None:
            self._loaded_regions = {}          # (falsy, but makes step() run the size check once)
        self.gnorm_sq = None
        self._sumsq_ws = None


def build_optimizer(model, cfg, grad_clip=None):
    cfg = dict(cfg)
    t = cfg.pop('type')
    cls = OPTIMIZERS.get(t)
    if cls is None:
        raise KeyError(f'optimizer {t} is not available on the HIP path')
    m = model.module if hasattr(model, 'module') else model
    return cls(m, grad_clip=grad_clip, **cfg)
