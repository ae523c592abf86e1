"""Where does the difference between the bf16-storage gradients and the fp32 gradients of one training step come from?
CPU, oracle only (no GPU, no reference): one step at 128x192 on the synthetic weights / batch of tests/golden/
make_trajectory.py, relative L2 distance of the gradient to the fp32 gradient per parameter group, with the bf16 rounding
switched on at selected storage points only.  Result: rounding the GRADIENT tensors alone gives
0.3-0.7 %; the 12-27 % of the full emulation is already there with the forward roundings alone - bf16 WEIGHTS alone
give 9-20 %, bf16 activations alone 10-23 %.  It is the sensitivity of this network's gradient to 2^-9 perturbations of
its forward pass, not an artefact of a backward kernel.
Usage: python tools/grad_noise_probe.py"""
import os
import sys

import numpy as np  # noqa: F401
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
from oracle import fcos_oracle as O
import make_trajectory as M
torch.set_num_threads(32)
sd=O.synth_state_dict(0); img,gtb,gtl=M.batches()[0]
def grads(q):
    l,g,_=O.train_step(sd,img,gtb,gtl,None,quant=q); return g
g32=grads(O.Quant(False))
keys=sorted(g32); gr=M.groups(keys)
def rep(name,q):
    g=grads(q)
    print(name, {n: round(M.dist(g,g32,gr[n]),4) for n in sorted(gr)}, flush=True)
rep('emu all        ', O.Quant(True))
rep('g_act fp32     ', O.Quant(True, fp32_grad_tags=('tower_act',)))
rep('g_act+g_pre f32', O.Quant(True, fp32_grad_tags=('tower_act','tower_pre')))
class FwdOnly(O.Quant):
    def act(self,x,tag=None): return O._RoundFwd.apply(x)
rep('no grad round  ', FwdOnly(True))
class GradOnly(O.Quant):
    def act(self,x,tag=None):
        class F_(torch.autograd.Function):
            @staticmethod
            def forward(ctx,x): return x
            @staticmethod
            def backward(ctx,g): return g.bfloat16().float()
        return F_.apply(x)
    def wt(self,w): return w
rep('grad round only', GradOnly(True))
class WtOnly(O.Quant):
    def act(self,x,tag=None): return x
rep('weights only   ', WtOnly(True))
class ActFwdOnly(O.Quant):
    def act(self,x,tag=None): return O._RoundFwd.apply(x)
    def wt(self,w): return w
rep('act fwd only   ', ActFwdOnly(True))
class TowerActOnly(O.Quant):
    def act(self,x,tag=None): return O._RoundFwd.apply(x) if tag in ('tower_pre','tower_act') else x
    def wt(self,w): return w
rep('tower acts only', TowerActOnly(True))
class NonTowerActOnly(O.Quant):
    def act(self,x,tag=None): return x if tag in ('tower_pre','tower_act') else O._RoundFwd.apply(x)
    def wt(self,w): return w
rep('backbone/fpn acts only', NonTowerActOnly(True))
