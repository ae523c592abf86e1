"""Fused SGD over the flat parameter buffer (dsl_sgd_step): momentum, weight decay, the reference's
paramwise rules (bias_lr_mult / bias_decay_mult), gradient-norm clipping and the bf16 re-pack in one
pass.  Replaces torch.optim.SGD + mmcv OptimizerHook's clip_grad_norm_
(mmdet/apis/train.py:111,157-166; configs/fcos_semi/*.py `optimizer`, `optimizer_config`)."""
import os

import torch

from . import _lib as L
from .registry import OPTIMIZERS


_PACK_SIDE = os.environ.get('DSL_PACK_SIDE', '1') != '0'     # data-gradient weight packs off the caller's stream (measured: tools/exp_r2l.sh)


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
        if hasattr(self.model, 'wait_grads'):
            self.model.wait_grads()
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
        gptr = None
        if self.max_norm is not None:
            if getattr(self, '_sumsq_ws', None) is None or self._sumsq_ws.device != st.device:
                self._sumsq_ws = torch.zeros(1024, device=st.device)
            L.check(L.lib.dsl_sumsq_det(L.ptr(st.grad), st.n_train, L.ptr(self.gnorm_sq), L.ptr(self._sumsq_ws), sp), 'dsl_sumsq_det')
            gptr = self.gnorm_sq
        lr = float(self.param_groups[0]['lr'])
        blr = float(self.param_groups[1]['lr']) / lr if lr != 0 else self.bias_lr_mult
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
