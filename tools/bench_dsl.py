"""Timing of the semi-supervised (DSL) iteration, BASELINE.json configs[2] (not the bench.py line, which is
configs[1]): labeled + unlabeled image per GPU (+ the half-scale copy, N = 3), ignore boxes, sisoft term, SGD, EMA
teacher update every iteration and - optionally - the teacher's pseudo-label refresh sweep of the unlabeled image.
Usage (GPU box): python tools/bench_dsl.py [--refresh] [--steps K]"""
import argparse
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from dsl_amd import detectors  # noqa: F401
from dsl_amd.optim import FlatSGD
from dsl_amd.registry import build_detector
from dsl_amd.runner import EMAOWNHook, OptimizerHook, SemiEpochBasedRunner, UnlabelPredHook

ap = argparse.ArgumentParser()
ap.add_argument('--steps', type=int, default=20)
ap.add_argument('--refresh', action='store_true', help='teacher sweep of the unlabeled image every iteration')
args = ap.parse_args()

student, teacher = build_detector(bench.model_cfg(dsl=True)).cuda(), build_detector(bench.model_cfg(dsl=True)).cuda()
student.lazy_log = True
student.eager_backward = True
opt = FlatSGD(student, lr=0.01, momentum=0.9, weight_decay=1e-4, paramwise_cfg=dict(bias_lr_mult=2., bias_decay_mult=0.),
              grad_clip=dict(max_norm=35, norm_type=2))
runner = SemiEpochBasedRunner(student, optimizer=opt, max_epochs=1, ema_model=teacher, scale_invariant=True)
runner.register_hook(OptimizerHook(grad_clip=dict(max_norm=35, norm_type=2)), priority=30)
runner.register_hook(EMAOWNHook(interval=1, mode='iteration', ratio=0.99, start_point=0), priority=40)
refresh = UnlabelPredHook(infer_score_thre=0.1, use_ema=True)
b = bench.synth_batch(0, 2)
rng = np.random.RandomState(7)
b['gt_bboxes_ignore'] = [torch.zeros(0, 4), torch.from_numpy(bench.synth_boxes(rng, max(1, rng.poisson(3))))]
for m in b['img_metas']:
    m['filename'] = 'unl.jpg'


class Refresh:
    priority = 45

    def __getattr__(self, name):
        return lambda r: None

    def after_train_iter(self, r):
        if args.refresh:
            refresh.refresh(r, b['img'][1:2], b['img_metas'][1:2], ['unl.jpg'])


runner.register_hook(Refresh(), priority=45)
warm = [b] * 5
runner.train(warm)
torch.cuda.synchronize()
t0 = time.perf_counter()
runner.train([b] * args.steps)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f'DSL iteration (N=3 student step + EMA{" + teacher refresh sweep" if args.refresh else ""}): {dt / args.steps * 1e3:.3f} ms/iter, '
      f'{2 * args.steps / dt:.1f} real imgs/s per GPU, losses {dict((k, round(float(v), 4)) for k, v in runner.outputs["log_vars"].items())}')
