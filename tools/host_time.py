"""How long the HOST needs to queue one training step (no device sync) vs the GPU time of the step."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from dsl_amd import detectors  # noqa: F401
from dsl_amd.optim import FlatSGD
from dsl_amd.registry import build_detector
model = build_detector(bench.model_cfg()).cuda()
model.lazy_log = True
model.eager_backward = '--eager' in sys.argv
opt = FlatSGD(model, lr=0.01, momentum=0.9, weight_decay=1e-4, paramwise_cfg=dict(bias_lr_mult=2., bias_decay_mult=0.))
batch = bench.synth_batch(0, 2)


def step():
    out = model.train_step(batch, opt)
    out['loss'].backward()
    opt.step()


for _ in range(5):
    step()
torch.cuda.synchronize()
K = 20
t0 = time.perf_counter()
marks = []
for _ in range(K):
    a = time.perf_counter()
    step()
    marks.append(time.perf_counter() - a)
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f'host queues a step in {t_host / K * 1e3:.3f} ms (min {min(marks)*1e3:.3f}, max {max(marks)*1e3:.3f}); '
      f'GPU completes a step every {t_all / K * 1e3:.3f} ms')
# split: forward_train / parse / backward / optimizer (host time only)
import collections
acc = collections.Counter()
for _ in range(K):
    a = time.perf_counter(); losses = model(**batch); b = time.perf_counter()
    loss, log = model._parse_losses(losses); c = time.perf_counter()
    loss.backward(); d = time.perf_counter()
    opt.step(); e = time.perf_counter()
    acc['forward_train'] += b - a; acc['parse_losses'] += c - b; acc['backward'] += d - c; acc['optimizer'] += e - d
torch.cuda.synchronize()
print({k: round(v / K * 1e3, 3) for k, v in acc.items()}, 'ms host time per step')
