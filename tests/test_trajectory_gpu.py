"""A multi-step training TRAJECTORY on the GPU against the CPU oracle (tests/golden/trajectory.npz, made by
tests/golden/make_trajectory.py): 60 SGD steps (momentum, weight decay, clip, bias rules) at 128x192 from the same
initial weights and the same four batches.

What is asserted, and why in this form.  The SGD trajectory of this network is sensitive to perturbations: an fp32 run
whose activations are perturbed by a relative 1e-6 (no bf16 anywhere) has weight updates 0.3 % away from the unperturbed
run after 1 step, 5 % after 5, 17 % after 10 and 90 % after 40 steps - while all runs' losses stay within 1 % of each
other.  So (a) the LOSS CURVE of the HIP run is held to the fp32 oracle over all 60 steps, within the band the
bf16-storage-emulating oracle runs themselves keep; (b) the WEIGHT UPDATES are compared over the first steps only
(1, 2, 5, 10), per parameter group, and must be no farther from the fp32 run than the emulated runs are - i.e. the
kernels add nothing beyond the noise of the bf16 activation storage that `north_star` asks for."""
import os

import numpy as np
import pytest
import torch

from util import fcos_model_cfg

pytestmark = pytest.mark.gpu
T = torch.from_numpy


def test_training_trajectory_vs_fp32_oracle(golden):
    from dsl_amd import detectors  # noqa: F401
    from dsl_amd.optim import FlatSGD
    from dsl_amd.registry import build_detector
    from oracle import fcos_oracle as O
    d = golden('trajectory.npz')
    steps, H, W, B, NB, SUB = (int(d[k]) for k in ('steps', 'H', 'W', 'B', 'NB', 'SUB'))
    model = build_detector(fcos_model_cfg())
    sd0 = O.synth_state_dict(0)
    model.load_state_dict(sd0)
    model = model.cuda()
    opt = FlatSGD(model, lr=float(d['lr']), momentum=float(d['momentum']), weight_decay=float(d['wd']),
                  paramwise_cfg=dict(bias_lr_mult=2., bias_decay_mult=0.), grad_clip=dict(max_norm=float(d['clip']), norm_type=2))
    rng = np.random.RandomState(11)                   # the batches of make_trajectory.batches()
    g = torch.Generator().manual_seed(12)
    bs = []
    for _ in range(NB):
        img = (torch.randn(B, 3, H, W, generator=g) * 40).bfloat16().float()
        gtb = [T(O.synth_boxes(rng, 3, H=H, W=W, lo=8, hi=100)) for _ in range(B)]
        gtl = [T(rng.randint(0, 80, len(b)).astype('int64')) for b in gtb]
        bs.append((img.cuda(), gtb, gtl))
    metas = [dict(img_shape=(H, W, 3), pad_shape=(H, W, 3), scale_factor=1.0)] * B
    losses, snaps = [], {}
    snap_at = [int(t) for t in d['snaps']]
    for it in range(steps):
        img, gtb, gtl = bs[it % NB]
        out = model.forward_train(img, metas, gtb, gtl)
        sum(out.values()).backward()
        opt.step()
        losses.append([float(out[k]) for k in ('loss_cls', 'loss_bbox', 'loss_centerness')])
        if it + 1 in snap_at:
            snaps[it + 1] = {k: v.detach().clone() for k, v in model.state_dict().items()}
    torch.cuda.synchronize()
    hip = np.array(losses).sum(1)
    l32, la, lb = d['loss32'].sum(1), d['lossA'].sum(1), d['lossB'].sum(1)
    # (a) loss curve.  Per step over the first 10 steps (before the trajectories decorrelate): within 2 % of the fp32
    # oracle or twice the emulated runs' own deviation.  Afterwards per 8-step window (two passes over the four batches):
    # within 3 % or 2.5 x the largest deviation any oracle run (bf16-emulating A / B, fp32 + 1e-6 perturbation) shows in
    # that window - those deviate by up to 13 % mid-run themselves - and never more than 25 %.
    lj = d['lossJ'].sum(1)
    dev = np.abs(hip - l32) / l32
    band = np.maximum(0.02, 2.0 * np.maximum(np.abs(la - l32), np.abs(lb - l32)) / l32)
    print('per-step deviation vs fp32, first 10 steps', np.round(dev[:10], 4), 'final losses', hip[-1], l32[-1])
    assert (dev[:10] <= band[:10]).all(), (dev[:10], band[:10])

    def wm(x, w=8):
        return np.array([x[i:i + w].mean() for i in range(0, len(x) - w + 1, w)])
    wdev = np.abs(wm(hip) - wm(l32)) / wm(l32)
    wref = np.max([np.abs(wm(x) - wm(l32)) / wm(l32) for x in (la, lb, lj)], axis=0)
    print('8-step window deviation: hip', np.round(wdev, 3), ' oracle runs (max)', np.round(wref, 3))
    assert (wdev <= np.minimum(0.25, np.maximum(0.03, 2.5 * wref))).all(), (wdev, wref)
    assert hip[-8:].mean() < 0.85 * hip[:4].mean()                     # it trains
    # (b) weight updates after 1, 2, 5, 10 steps, 1-in-SUB subsample, per parameter group
    keys = [str(k) for k in d['keys']]
    sizes = np.array([sd0[k].numel() for k in keys])
    starts = np.concatenate([[0], np.cumsum(sizes)])
    owner = np.searchsorted(starts, np.arange(0, starts[-1], SUB), side='right') - 1

    def grp(k):
        if k.startswith('backbone.layer'):
            return k.split('.')[1]
        if k.startswith('neck.'):
            return 'fpn'
        return 'towers' if ('cls_convs' in k or 'reg_convs' in k) else 'predictors'
    gname = np.array([grp(k) for k in keys])[owner]
    for t in snap_at:
        upd = torch.cat([(snaps[t][k].cpu().float() - sd0[k]).flatten() for k in keys])[::SUB].numpy()
        ref = d[f'update32_sub_{t}']
        assert upd.shape == ref.shape
        for i, n in enumerate(str(x) for x in d['group_names']):
            m = gname == n
            dist = float(np.linalg.norm(upd[m] - ref[m]) / np.linalg.norm(ref[m]))
            floor = max(float(d[f'dA_{t}'][i]), float(d[f'dB_{t}'][i]))
            print(f'step {t:2d} {n:10s} update distance to fp32: hip {dist:.3f}   emulated-bf16 {float(d[f"dA_{t}"][i]):.3f} '
                  f'{float(d[f"dB_{t}"][i]):.3f}   fp32 + 1e-6 perturbation {float(d[f"dJ_{t}"][i]):.4f}')
            assert dist <= 1.3 * floor + 0.02, (t, n, dist, floor)
