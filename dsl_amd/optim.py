"""Fused SGD over the flat parameter buffer (dsl_sgd_step): momentum, weight decay, the reference's
paramwise rules (bias_lr_mult / bias_decay_mult), gradient-norm clipping and the bf16 re-pack in one
pass.  Replaces torch.optim.SGD + mmcv OptimizerHook's clip_grad_norm_
(mmdet/apis/train.py:111,157-166; configs/fcos_semi/*.py `optimizer`, `optimizer_config`)."""
import ctypes as C
import os

import torch

from . import _lib as L
from .registry import OPTIMIZERS


_BUCKET_SGD = os.environ.get('DSL_BUCKET_SGD', '1') != '0'    # per-bucket optimizer steps beside the backward pass (no clipping only)
_PACK_SIDE = os.environ.get('DSL_PACK_SIDE', '1') != '0'     # data-gradient weight packs off the caller's stream (measured: tools/experiments_r2.txt (exp_r2l))


@OPTIMIZERS.register_module(name='SGD')
class FlatSGD:
    def __init__(self, model, lr=0.01, momentum=0.9, weight_decay=0.0, paramwise_cfg=None, grad_clip=None, **kw):
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

    def zero_grad(self, set_to_none=False):
        # every gradient element is overwritten by the backward kernels; nothing to clear
        return

    def step(self):
        st = self.store
        if self.momentum_buf is None:
            self.momentum_buf = torch.zeros_like(st.train)
            self.steps = 0
        elif self.momentum_buf.device != st.device or self.momentum_buf.shape != st.train.shape:
            # state restored from a checkpoint (CPU tensor, possibly without the tile padding of this build): move it
            # into a buffer laid out like the parameter buffer and keep the step count (resume must not re-zero momentum)
            buf = torch.zeros_like(st.train)
            n = min(buf.numel(), self.momentum_buf.numel())
            buf.view(-1)[:n].copy_(self.momentum_buf.reshape(-1)[:n].to(device=st.device, dtype=buf.dtype))
            self.momentum_buf = buf
        if self.gnorm_sq is None or self.gnorm_sq.device != st.device:
            self.gnorm_sq = torch.zeros(1, device=st.device)
        sp = L.stream_ptr()
        lr = float(self.param_groups[0]['lr'])
        blr = float(self.param_groups[1]['lr']) / lr if lr != 0 else self.bias_lr_mult
        infos = getattr(self.model, '_last_bwd_infos', None)
        if self.max_norm is None and _BUCKET_SGD and infos and st.grad.is_cuda and all(i['bucket'][0] % 4 == 0 for i in infos):
            # No gradient clipping (the supervised config): the update is element-wise, so each gradient bucket - head + FPN,
            # layer4, layer3, layer2, in the order the backward pass completes them - is updated on the optimizer's own stream
            # as soon as its weight gradients (data parallel: its all-reduce) are done, beside the rest of the backward pass,
            # instead of one pass over all 32 M parameters behind the last weight gradient.  Same arithmetic, same bits.
            cur = torch.cuda.current_stream()
            if getattr(self, '_opt_stream', None) is None:
                self._opt_stream = torch.cuda.Stream()
            os_ = self._opt_stream
            osp = C.c_void_p(os_.cuda_stream)
            pend = list(getattr(self.model, '_pending', []) or [])
            for k, info in enumerate(infos):
                lo, hi = info['bucket']
                if pend:
                    with torch.cuda.stream(os_):
                        pend[k].wait()
                        tr = getattr(self.model, 'comm_trace', None)
                        if tr and k < len(tr[-1]['buckets']):
                            ed = torch.cuda.Event(enable_timing=True)
                            ed.record()
                            tr[-1]['buckets'][k]['done'] = ed
                else:
                    never = L.lib.dsl_stream_wait_slot(int(info['slot']), osp)
                    if info['main'] or never != 0:
                        os_.wait_stream(cur)
                o4, o2, o1 = lo * 4, lo * 2, lo
                L.check(L.lib.dsl_sgd_step(C.c_void_p(st.train.data_ptr() + o4), C.c_void_p(st.grad.data_ptr() + o4),
                                           C.c_void_p(self.momentum_buf.data_ptr() + o4), C.c_void_p(st.train16.data_ptr() + o2),
                                           C.c_void_p(st.group.data_ptr() + o1), hi - lo, lr, self.momentum, self.weight_decay, blr,
                                           self.bias_decay_mult, None, 0.0, int(self.steps == 0), osp), 'dsl_sgd_step')
            if hasattr(self.model, '_pending'):
                self.model._pending = []
            cur.wait_stream(os_)
            st.repack_dgrad(sp, side=_PACK_SIDE)
            self.steps += 1
            return
        if hasattr(self.model, 'wait_grads'):
            self.model.wait_grads()
        gptr = None
        if self.max_norm is not None:
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
        return dict(momentum=self.momentum_buf, steps=self.steps, param_groups=self.param_groups)

    def load_state_dict(self, sd):
        """Restores momentum, the step count (first-step rule of torch.optim.SGD: buf = grad) and the groups' learning
        rates.  The momentum tensor may live on the CPU (runner.resume loads with map_location='cpu'); step() moves it."""
        m = sd['momentum']
        self.momentum_buf = m.detach().clone() if isinstance(m, torch.Tensor) else None
        self.steps = int(sd['steps']) if self.momentum_buf is not None else 0
        self.param_groups = [dict(g) for g in sd['param_groups']]
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
