import os, sys, time
sys.path.insert(0, '/root/repo' if os.path.exists('/root/repo/bench.py') else os.getcwd())
import torch, bench, collections
from dsl_amd import detectors, head_loss, engine
from dsl_amd.optim import FlatSGD
from dsl_amd.registry import build_detector
model = build_detector(bench.model_cfg()).cuda()
model.lazy_log = True; model.eager_backward = True
opt = FlatSGD(model, lr=0.01, momentum=0.9, weight_decay=1e-4, paramwise_cfg=dict(bias_lr_mult=2., bias_decay_mult=0.))
batch = bench.synth_batch(0, 2)
from dsl_amd.data import mark_ready
_ev = torch.cuda.Event(); _ev.record()
acc = collections.Counter()
def wrap(obj, name, key):
    f = getattr(obj, name)
    def g(*a, **k):
        t = time.perf_counter(); r = f(*a, **k); acc[key] += time.perf_counter() - t; return r
    setattr(obj, name, g)
wrap(engine.OpList, 'run', 'OpList.run')
wrap(head_loss.FcosLossPlan, 'set_targets', 'set_targets')
wrap(model, '_run_backward', '_run_backward')
wrap(model, '_parse_losses', '_parse_losses')
wrap(opt, 'step', 'opt.step')
def step():
    mark_ready(batch['img'], event=_ev); out = model.train_step(batch, opt); out['loss'].backward(); opt.step()
for _ in range(6): step()
torch.cuda.synchronize(); acc.clear()
K = 20; t0 = time.perf_counter()
for _ in range(K): step()
th = time.perf_counter() - t0
torch.cuda.synchronize()
print('host per step %.3f ms' % (th / K * 1e3), {k: round(v / K * 1e3, 3) for k, v in acc.items()})
