"""The semi-supervised iteration end to end on the GPU (SURVEY.md §8 rows a13-a15 together):
SemiEpochBasedRunner.train = dual-stream batch (+ half-scale copy) -> train_step -> OptimizerHook -> EMAOWNHook ->
teacher sweep / pseudo-label refresh, against the CPU oracle of the same pieces."""
import copy
import json
import os

import numpy as np
import pytest
import torch

from util import fcos_model_cfg

pytestmark = pytest.mark.gpu
T = torch.from_numpy


def build(**head):
    from dsl_amd import detectors  # noqa: F401
    from dsl_amd.registry import build_detector
    from oracle import fcos_oracle as O
    model = build_detector(fcos_model_cfg(**head))
    model.load_state_dict(O.synth_state_dict(0))
    return model.cuda()


def make_batches(n_iter, H=128, W=192):
    from oracle import fcos_oracle as O
    rng = np.random.RandomState(5)
    g = torch.Generator().manual_seed(9)
    out = []
    for _ in range(n_iter):
        img = (torch.randn(2, 3, H, W, generator=g) * 30).bfloat16().float()
        gtb = [T(O.synth_boxes(rng, 3, H=H, W=W, lo=8, hi=100)) for _ in range(2)]
        gtl = [T(rng.randint(0, 80, len(b)).astype('int64')) for b in gtb]
        ig = [torch.zeros(0, 4), T(O.synth_boxes(rng, 2, H=H, W=W, lo=8, hi=100))]      # labeled image has none
        metas = [dict(img_shape=(H, W - 2, 3), pad_shape=(H, W, 3), scale_factor=1.0, filename=f'im{i}.jpg') for i in range(2)]
        out.append(dict(img=img.cuda(), img_metas=metas, gt_bboxes=gtb, gt_labels=gtl, gt_bboxes_ignore=ig))
    return out


def test_semi_supervised_iterations(tmp_path):
    from dsl_amd.optim import FlatSGD
    from dsl_amd.runner import EMAOWNHook, OptimizerHook, SemiEpochBasedRunner, UnlabelPredHook
    from oracle import fcos_oracle as O
    head = dict(loss_weight=3.0, soft_weight=1.0, soft_warm_up=0)
    student, teacher = build(**head), build(**head)
    student.bbox_head.cur_iter = 1                 # past the warm-up window: full sisoft weight
    opt = FlatSGD(student, lr=0.01, momentum=0.9, weight_decay=1e-4, paramwise_cfg=dict(bias_lr_mult=2., bias_decay_mult=0.),
                  grad_clip=dict(max_norm=35, norm_type=2))
    runner = SemiEpochBasedRunner(student, optimizer=opt, max_epochs=1, ema_model=teacher, scale_invariant=True)
    runner.register_hook(OptimizerHook(grad_clip=dict(max_norm=35, norm_type=2)), priority=30)
    runner.register_hook(EMAOWNHook(interval=1, mode='iteration', ratio=0.9, start_point=0), priority=40)
    batches = make_batches(3)

    class Spy:          # records what the runner hands to / gets from each iteration
        priority = 45
        log = []

        def __getattr__(self, name):
            return lambda runner: None

        def before_train_iter(self, r):
            self.s0 = r._det(r.model).store.train.clone()
            self.t0 = r._det(r.ema_model).store.train.clone()

        def after_train_iter(self, r):
            s1, t1 = r._det(r.model).store.train, r._det(r.ema_model).store.train
            Spy.log.append(dict(loss={k: float(v) for k, v in r.outputs['log_vars'].items()}, n=r.outputs['num_samples'],
                                ema_err=float((t1 - (0.9 * self.t0 + 0.1 * s1)).abs().max()),
                                moved=float((s1 - self.s0).abs().max())))
    runner.register_hook(Spy(), priority=45)
    sd0 = {k: v.clone().cpu() for k, v in student.state_dict().items()}
    runner.run([batches], max_epochs=1)
    torch.cuda.synchronize()
    assert runner.iter == 3 and runner.epoch == 1 and len(Spy.log) == 3
    for e in Spy.log:
        assert e['n'] == 3                                   # labeled + unlabeled + half-scale copy
        assert all(np.isfinite(v) for v in e['loss'].values()) and 'loss_sisoft' in e['loss']
        assert e['moved'] > 0                                 # the optimizer stepped
        assert e['ema_err'] < 1e-6                            # teacher = 0.9 teacher + 0.1 student (a14)
    # iteration 0 against the oracle: same dual-stream + half-scale batch, same DSL loss
    b = batches[0]
    img3, gb3, gl3, ig3 = O.append_half_scale(b['img'].cpu(), b['gt_bboxes'], b['gt_labels'], b['gt_bboxes_ignore'])
    ref, _, _ = O.train_step(sd0, img3, gb3, gl3, ig3, emulate_bf16=False, want_grads=False, loss_weight=3.0, soft_weight=1.0)
    for k, v in ref.items():
        assert Spy.log[0]['loss'][k] == pytest.approx(float(v), rel=2e-3), (k, Spy.log[0]['loss'][k], float(v))
    # bf16 forward pack of the teacher follows its master weights after EMA
    ts = teacher.store
    assert torch.equal(ts.train16.float(), ts.train.bfloat16().float())
    # a15: teacher sweep -> pseudo-label bank -> thresholds -> (gt, ignore) split, JSON export in the reference's layout
    hook = UnlabelPredHook(infer_score_thre=0.0, use_ema=True, export_dir=str(tmp_path))
    names = ['im_a.jpg', 'im_b.jpg']
    bank = hook.refresh(runner, batches[1]['img'], batches[1]['img_metas'], names)
    assert set(bank) == set(names)
    for n in names:
        e = bank[n]
        assert e['rects'].dtype == np.int64 and e['rects'].shape[1] == 4 and len(e['tags']) == len(e['scores']) == len(e['rects'])
        j = json.load(open(os.path.join(str(tmp_path), n + '.json')))
        assert set(j) == {'imageName', 'targetNum', 'rects', 'tags', 'masks', 'scores'} and j['targetNum'] == len(e['tags'])
    # the same detections as the oracle's get_bboxes on the teacher's weights
    tsd = {k: v.clone().cpu() for k, v in teacher.state_dict().items()}
    with torch.no_grad():
        cls, reg, ctr = O.extract_and_head(tsd, batches[1]['img'].cpu(), O.Quant(True), training=False)[:3]
    m = batches[1]['img_metas'][0]
    dets = O.get_bboxes(cls, reg, ctr, [m['img_shape']] * 2, [[1.0, 1.0, 1.0, 1.0]] * 2)
    for i, n in enumerate(names):
        n_ref = len(dets[i][0])
        assert abs(n_ref - len(bank[n]['tags'])) <= max(2, n_ref // 10), (n_ref, len(bank[n]['tags']))
    hook.update_thresholds()
    gt, gl, ig = hook.targets_for(names[0], img_wh=(190, 128))
    assert gt.shape[1] == 4 and ig.shape[1] == 4 and len(gl) == len(gt)
