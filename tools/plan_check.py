"""Does the launch planner (conv.hip conv_choose: a cost model fitted in round 3) still pick the fastest tile for every convolution
of the training step?  Every distinct conv / data-gradient descriptor of the plan is replayed alone with the planner's choice and
with every forced (tile configuration, split-K) pair of the test hook (dsl_conv_desc.flags bits 8-15); rows where a forced pair
beats the choice by more than 8 % are printed.   Usage (GPU box): python tools/plan_check.py"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from dsl_amd import _lib as L
from dsl_amd import detectors  # noqa: F401
from dsl_amd.optim import FlatSGD
from dsl_amd.registry import build_detector

model = build_detector(bench.model_cfg()).cuda()
model.lazy_log = True
opt = FlatSGD(model, lr=0.01, momentum=0.9, weight_decay=1e-4, paramwise_cfg=dict(bias_lr_mult=2., bias_decay_mult=0.))
batch = bench.synth_batch(0, 2)
for _ in range(2):
    out = model.train_step(batch, opt)
    out['loss'].backward()
    opt.step()
torch.cuda.synchronize()
plan = [p for p in model._engine.plans.values() if p.training][0]


def timeit(op, reps=12):
    arr = (L.Op * 1)(op)
    arr[0].i[6] = 0
    for _ in range(2):
        if L.lib.dsl_run_ops(arr, 1, L.stream_ptr()) != 0:
            return None
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        L.lib.dsl_run_ops(arr, 1, L.stream_ptr())
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


seen, rows = {}, []
lists = [('fwd', plan.fwd)] + ([('prefix', plan.prefix)] if plan.prefix is not None else []) + [(f'bwd{i}', ol) for i, (ol, _) in enumerate(plan.bwd_segments)]
for lname, ol in lists:
    for op in ol.items:
        if op.kind != L.OP_CONV:
            continue
        d = C.cast(op.desc, C.POINTER(L.ConvDesc)).contents
        if d.gn_ws or (d.flags & (L.CONV_SMALL_C | L.CONV_FP8)):
            continue
        px = sum(d.n * d.gh[s] * d.gw[s] for s in range(d.nseg))
        key = (d.mode, d.kh, d.stride, d.os, d.cs, d.cd, px, d.nseg, bool(d.addend), bool(d.mask), d.flags)
        if key in seen:
            seen[key] += 1
            continue
        seen[key] = 1
        f0 = d.flags
        t0 = timeit(op)
        best = (t0, 0, 0)
        alts = []
        for cfg in range(1, 9):
            for sp in (1, 2, 4, 8):
                d.flags = (f0 & ~0xff00) | (cfg << 8) | ((sp if sp > 1 else 0) << 12)
                t = timeit(op, reps=6)
                if t is not None:
                    alts.append((t, cfg, sp))
        d.flags = f0
        alts.sort()
        desc = f"{lname:6s} {'dgrad' if d.mode else 'conv '} {d.kh}x{d.kw} s{d.stride} os{d.os} {d.cs:4d}->{d.cd:4d} px{px:6d} nseg{d.nseg} add{int(bool(d.addend))} mask{int(bool(d.mask))} f32{int(bool(d.flags & L.CONV_OUT_F32))}"
        rows.append((desc, key, t0, alts[:3]))
tot_gain = 0.0
print('# %d distinct convolution descriptors; "choice" = the planner, then the three fastest forced (tile cfg, split-K) pairs' % len(rows))
for desc, key, t0, alts in sorted(rows, key=lambda r: -(r[2] - r[3][0][0]) * seen[r[1]]):
    b = alts[0]
    flag = ' <--' if b[0] < 0.92 * t0 else ''
    if flag:
        tot_gain += (t0 - b[0]) * seen[key]
    print(f'{desc} x{seen[key]:2d}  choice {t0:6.1f} us | ' + ', '.join(f'cfg{c} sp{s} {t:6.1f}' for t, c, s in alts) + flag)
print('# sum over the flagged rows of (choice - best) x launches: %.1f us of standalone time per step' % tot_gain)
