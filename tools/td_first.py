"""Is the gap between bench.py's loop and train_detector (extra.train_detector) the hooks or the stream / hardware-queue layout of a
process that built other models first?  train_detector as the FIRST thing a fresh process does, then the bench loop in the same process."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
batch = bench.synth_batch(0, 2)
print('train_detector first:', {k: v for k, v in bench.train_detector_timing(batch, steps=40, warm=8).items() if k in ('imgs_per_s', 'ms_per_step')})
from dsl_amd.data import mark_ready
from dsl_amd.optim import FlatSGD
from dsl_amd.registry import build_detector
model = build_detector(bench.model_cfg()).cuda()
opt = FlatSGD(model, lr=0.01, momentum=0.9, weight_decay=1e-4, paramwise_cfg=dict(bias_lr_mult=2., bias_decay_mult=0.))
ev = torch.cuda.Event(); ev.record()
def step():
    mark_ready(batch['img'], event=ev); out = model.train_step(batch, opt); out['loss'].backward(); opt.step()
for _ in range(8): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(40): step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 40
print('bench loop second: %.1f img/s %.3f ms' % (2 / dt, dt * 1e3))
