"""Multi-step training trajectories of the CPU oracle (no reference import needed: the oracle is pinned to the
reference by the other fixtures): the fp32 run, two bf16-storage-emulating runs that differ only in the realisation of
the storage rounding, and an fp32 run perturbed at the 1e-6 level (no bf16 at all).  Output: trajectory.npz - per-step
losses, 1-in-SUB subsamples of the fp32 run's weight updates after SNAPS steps, and the distances of the other runs to
them, which tests/test_trajectory_gpu.py uses as the noise floor for the HIP run of the same steps.

Finding that shapes the test: the SGD trajectory of this network is sensitive to perturbations - the 1e-6 fp32
perturbation alone moves the weight update by 0.3 % after 1 step, 5 % after 5, 17 % after 10, 40 % after 20 and 90 % after
40 steps, while the losses of all runs stay within 1 % of each other.  Weight distances are therefore only
meaningful over the first few steps; beyond that the loss curve is the criterion.

Run here (CPU, ~10 min):  python tests/golden/make_trajectory.py
"""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import fcos_oracle as O  # noqa: E402

STEPS, H, W, B, NB, SUB = 60, 128, 192, 2, 4, 128
SNAPS = (1, 2, 5, 10)
LR, MOM, WD, CLIP = 0.01, 0.9, 1e-4, 35.0


def batches():
    rng = np.random.RandomState(11)
    g = torch.Generator().manual_seed(12)
    out = []
    for _ in range(NB):
        img = (torch.randn(B, 3, H, W, generator=g) * 40).bfloat16().float()
        gtb = [torch.from_numpy(O.synth_boxes(rng, 3, H=H, W=W, lo=8, hi=100)) for _ in range(B)]
        gtl = [torch.from_numpy(rng.randint(0, 80, len(b)).astype('int64')) for b in gtb]
        out.append((img, gtb, gtl))
    return out


def groups(keys):
    def grp(k):
        if k.startswith('backbone.layer'):
            return k.split('.')[1]
        if k.startswith('neck.'):
            return 'fpn'
        if 'cls_convs' in k or 'reg_convs' in k:
            return 'towers'
        return 'predictors'
    out = {}
    for k in keys:
        out.setdefault(grp(k), []).append(k)
    return out


def run(quant_fn):
    sd = O.synth_state_dict(0)
    tk = O.trainable_keys(sd)
    params = {k: sd[k].clone() for k in tk}
    bufs, losses, snaps = {}, [], {}
    bs = batches()
    for it in range(STEPS):
        img, gtb, gtl = bs[it % NB]
        cur = dict(sd)
        cur.update(params)
        l, grads, _ = O.train_step(cur, img, gtb, gtl, None, quant=quant_fn(it))
        params, bufs = O.sgd_step(params, grads, bufs, base_lr=LR, momentum=MOM, base_wd=WD, max_norm=CLIP, first_step=(it == 0))
        losses.append([l['loss_cls'], l['loss_bbox'], l['loss_centerness']])
        if it + 1 in SNAPS:
            snaps[it + 1] = {k: v.clone() for k, v in params.items()}
    return np.array(losses, np.float64), snaps


def dist(a, b, keys):
    num = sum(float(((a[k].double() - b[k].double()) ** 2).sum()) for k in keys)
    den = sum(float((b[k].double() ** 2).sum()) for k in keys)
    return (num / den) ** 0.5


if __name__ == '__main__':
    torch.set_num_threads(min(32, os.cpu_count() or 8))
    t0 = time.time()
    l32, p32 = run(lambda it: O.Quant(False))
    print('fp32 done', time.time() - t0, l32[0].sum(), l32[-1].sum(), flush=True)
    la, pa = run(lambda it: O.Quant(True, jitter=1e-6, seed=1000 + it))
    print('emu A done', time.time() - t0, la[-1].sum(), flush=True)
    lb, pb = run(lambda it: O.Quant(True, jitter=1e-6, seed=5000 + it))
    print('emu B done', time.time() - t0, lb[-1].sum(), flush=True)
    lj, pj = run(lambda it: O.Quant(False, jitter=1e-6, seed=9000 + it))
    print('fp32 + 1e-6 perturbation done', time.time() - t0, lj[-1].sum(), flush=True)
    sd0 = O.synth_state_dict(0)
    keys = sorted(p32[SNAPS[0]])
    gr = groups(keys)
    names = sorted(gr)
    out = dict(steps=STEPS, H=H, W=W, B=B, NB=NB, SUB=SUB, lr=LR, momentum=MOM, wd=WD, clip=CLIP, snaps=np.array(SNAPS),
               loss32=l32, lossA=la, lossB=lb, lossJ=lj, group_names=np.array(names), keys=np.array(keys))
    for t in SNAPS:
        # distances of the weight UPDATES (w_t - w_0): the initial weights are common and would hide the differences
        upd = {n: {k: p[t][k] - sd0[k] for k in keys} for n, p in (('32', p32), ('A', pa), ('B', pb), ('J', pj))}
        for n in ('A', 'B', 'J'):
            out[f'd{n}_{t}'] = np.array([dist(upd[n], upd['32'], gr[g]) for g in names])
        out[f'dAB_{t}'] = np.array([dist(upd['A'], upd['B'], gr[g]) for g in names])
        out[f'update32_sub_{t}'] = torch.cat([upd['32'][k].flatten() for k in keys])[::SUB].numpy().astype(np.float16 if False else np.float32)
        print(t, 'dA', np.round(out[f'dA_{t}'], 3), 'dB', np.round(out[f'dB_{t}'], 3), 'dAB', np.round(out[f'dAB_{t}'], 3),
              'dJ', np.round(out[f'dJ_{t}'], 4))
    np.savez_compressed(os.path.join(HERE, 'trajectory.npz'), **out)
    print('groups', names)
    print('wrote trajectory.npz', os.path.getsize(os.path.join(HERE, 'trajectory.npz')) / 1e6, 'MB', time.time() - t0, 's')
