#!/usr/bin/env python
"""Benchmark of the FCOS R50-FPN training step (BASELINE.json metric: imgs/sec per training step,
1333x800, at 1/2/4/8 MI355X).

  python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run)

A "step" = one full pass of the hot path over one synthetic batch: image pack -> ResNet-50 -> FPN ->
FCOS head -> target assignment -> focal/GIoU/centerness loss -> hand-written backward -> (N>1: RCCL
gradient all-reduce overlapped with backward) -> fused SGD step + bf16 re-pack.  Inputs are resident
in HBM before the timed region.  Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

GFLOP_PER_IMAGE_STEP = 1185.8      # SURVEY.md §8d / BASELINE.md: fwd 419.55 + bwd 766.23 GFLOP per 800x1344 image
PEAK_BF16_TFLOPS = 2500.0          # MI355X dense bf16 MFMA (guides/MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0              # HBM3E spec (same guide; ~6.3 TB/s achievable)
DOMINANT = 'conv_pipe_kernel<128, 128, 2, 4, 2>'    # the kernel class 0 of dsl_prof_* brackets (largest share of the step)


def model_cfg(dsl=False, rla=False, fp8=False):
    head = dict(type='FCOSHead', num_classes=80, in_channels=256, stacked_convs=4, feat_channels=256,
                strides=[8, 16, 32, 64, 128], norm_on_bbox=True, centerness_on_reg=True, dcn_on_last_conv=False,
                center_sampling=True, conv_bias=True,
                loss_cls=dict(type='FocalLoss', use_sigmoid=True, gamma=2.0, alpha=0.25, loss_weight=1.0),
                loss_bbox=dict(type='GIoULoss', loss_weight=1.0),
                loss_centerness=dict(type='CrossEntropyLoss', use_sigmoid=True, loss_weight=1.0))
    if dsl:
        head.update(loss_weight=3.0, soft_weight=1.0, soft_warm_up=5000)
    backbone = dict(type='ResNet', depth=50, num_stages=4, out_indices=(0, 1, 2, 3), frozen_stages=1,
                    norm_cfg=dict(type='BN', requires_grad=False), norm_eval=True, style='caffe')
    if rla:     # the backbone the reference's DSL config names (configs/fcos_semi/RLA_*.py:3-13)
        backbone = dict(type='RLA_ResNet', layers=[3, 4, 6, 3], frozen_stages=1, norm_eval=True, style='pytorch')
    return dict(type='FCOS', **(dict(fp8=dict(layers='towers')) if fp8 else {}),
                backbone=backbone,
                neck=dict(type='FPN', in_channels=[256, 512, 1024, 2048], out_channels=256, start_level=1,
                          add_extra_convs='on_output', num_outs=5, relu_before_extra_convs=True),
                bbox_head=head,
                test_cfg=dict(nms_pre=1000, min_bbox_size=0, score_thr=0.05, nms=dict(type='nms', iou_threshold=0.5),
                              max_per_img=100))


def synth_boxes(rng, n, H=800, W=1333, lo=16.0, hi=600.0):
    cx, cy = rng.uniform(0, W, n), rng.uniform(0, H, n)
    w = np.exp(rng.uniform(np.log(lo), np.log(min(hi, W)), n))
    h = np.exp(rng.uniform(np.log(lo), np.log(min(hi, H)), n))
    b = np.stack([np.clip(cx - w / 2, 0, W), np.clip(cy - h / 2, 0, H), np.clip(cx + w / 2, 0, W),
                  np.clip(cy + h / 2, 0, H)], 1).astype('float32')
    return b[((b[:, 2] - b[:, 0]) >= 1) & ((b[:, 3] - b[:, 1]) >= 1)]


def synth_batch(rank, n_img=2, H=800, W=1344, device='cuda'):
    """N(0,1) image values (bf16-representable), G ~ clip(Poisson(7), 1, 40) boxes,
    log-uniform sizes 16..600 px inside the 1333x800 area, labels U{0..79}."""
    g = torch.Generator().manual_seed(1234 + rank)
    img = torch.randn(n_img, 3, H, W, generator=g).bfloat16().float()
    rng = np.random.RandomState(2024 + rank)
    gtb, gtl = [], []
    for _ in range(n_img):
        b = synth_boxes(rng, int(np.clip(rng.poisson(7), 1, 40)))
        gtb.append(torch.from_numpy(b))
        gtl.append(torch.from_numpy(rng.randint(0, 80, len(b)).astype('int64')))
    metas = [dict(img_shape=(800, 1333, 3), pad_shape=(H, W, 3), scale_factor=1.0) for _ in range(n_img)]
    return dict(img=img.to(device), img_metas=metas, gt_bboxes=gtb, gt_labels=gtl)


def _lib_option(name):
    from dsl_amd import _lib as L
    v = C.c_int(0)
    L.check(L.lib.dsl_get_option(name.encode(), C.byref(v)), 'dsl_get_option')
    return int(v.value)


def _streams_probe():
    """How many of the library's three concurrently used streams sit on hardware queues of their own (dsl_streams_init; 3 = all)."""
    from dsl_amd import _lib as L
    v = C.c_int(0)
    L.check(L.lib.dsl_streams_init(L.stream_ptr(), C.byref(v)), 'dsl_streams_init')
    return int(v.value)


def _release_earlier_models():
    """Before an extra's model is built: collect the models earlier measurements left behind NOW.  Otherwise the garbage collector
    finds them some iterations into the next timed window, and releasing a model's device-side state (events, the library's cached
    tables) stalls the device once for ~30 ms - 375 instead of 437 img/s over a 40-step window, and never again in later windows of
    the same model (tools/second_model_probe.py, profiles/r04_second_model.txt)."""
    import gc
    gc.collect()
    if torch.cuda.is_available():
        torch.cuda.synchronize()



def fp8_step_timing(batch, steps=20, warm=5):
    """BASELINE.json configs[4], first slice (beside the bf16 headline, never instead of it): the same supervised step with the head
    towers' forward convolutions on the fp8 MFMA path (FCOS(fp8=dict(layers='towers')), delayed per-tensor activation scales: the
    GroupNorm passes write the e4m3 copies, one dsl_fp8_prep launch per step)."""
    from dsl_amd.data import mark_ready
    from dsl_amd.optim import FlatSGD
    from dsl_amd.registry import build_detector
    _release_earlier_models()
    model = build_detector(model_cfg(fp8=True)).cuda()
    opt = FlatSGD(model, lr=0.01, momentum=0.9, weight_decay=1e-4, paramwise_cfg=dict(bias_lr_mult=2., bias_decay_mult=0.))
    ev = torch.cuda.Event()
    ev.record()                       # the batch is resident: its producer's event, handed over with every batch (mark_ready is one-shot)

    def step():
        mark_ready(batch['img'], event=ev)
        out = model.train_step(batch, opt)
        out['loss'].backward()
        opt.step()
        return out
    for _ in range(warm):
        out = step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    losses = {k: round(float(v), 4) for k, v in out['log_vars'].items()}
    del model, opt
    torch.cuda.empty_cache()
    return dict(imgs_per_s=round(len(batch['img_metas']) / dt, 1), ms_per_step=round(dt * 1e3, 3), final_losses=losses,
                note='same step, batch and optimizer as the headline; forward of the 8 tower convolutions in OCP e4m3 on '
                     'v_mfma_scale_f32_32x32x64_f8f6f4, everything else (and the whole backward pass) bf16; off by default in the product')


def comm_proxy_timing(batch, steps=20, warm=5, rounds=3, wgs=32, passes=2, carriers=('lib', 'torch', 'lib_eager')):
    """Multi-GPU first contact de-risked on ONE GPU (the reference's DDP: mmdet/apis/train.py:92-96): the
    headline's step in its data-parallel schedule - named bucket events, communication stream, per-bucket optimizer steps behind each
    bucket's exchange - with dsl_comm_proxy in place of the all-reduce: `wgs` workgroups making `passes` read-modify-write passes
    over the bucket (RCCL's ring: one workgroup per channel; reduce-scatter + all-gather each read and write the bucket once), on
    (lib) the library's PLACED communication stream - the C-ABI carrier's, comm='rccl' - and (torch) a stream from torch's pool, as
    ProcessGroupNCCL runs its kernels on one of its own.  No xGMI, no peer latency: what is measured is what the collectives' device
    footprint and their stream cost the backward pass.  Alternated `rounds` times on one model; medians."""
    import statistics
    from dsl_amd import _lib as L
    from dsl_amd.data import mark_ready
    from dsl_amd.optim import FlatSGD
    from dsl_amd.registry import build_detector
    _release_earlier_models()
    model = build_detector(model_cfg()).cuda()
    opt = FlatSGD(model, lr=0.01, momentum=0.9, weight_decay=1e-4, paramwise_cfg=dict(bias_lr_mult=2., bias_decay_mult=0.))
    ev = torch.cuda.Event()
    ev.record()

    def step():
        mark_ready(batch['img'], event=ev)
        out = model.train_step(batch, opt)
        out['loss'].backward()
        opt.step()
        return out

    def run(mode):
        # '<carrier>' = the default schedule (late exchange: collectives behind the backward pass, next forward waits per stage),
        # '<carrier>_eager' = bucket by bucket behind each backward segment, everything joined at the end of the step (round 5)
        model.comm_proxy = None if mode == 'none' else dict(carrier=mode.split('_')[0], wgs=wgs, passes=passes)
        opt.late_exchange = not mode.endswith('_eager')
        opt._sync_defer()
        for _ in range(warm):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps * 1e3
    modes = ('none',) + tuple(carriers)
    res = {m: [] for m in modes}
    for _ in range(rounds):
        for m in modes:
            res[m].append(run(m))
    model.comm_proxy = None
    # the proxy alone: what each bucket's stand-in costs on an idle chip (its own "bus bandwidth")
    st = model.store
    alone = []
    for lo, hi in st.grad_buckets():
        g = st.grad[lo:hi]
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        L.check(L.lib.dsl_comm_proxy(L.ptr(g), (hi - lo) // 4 * 4, wgs, passes, L.stream_ptr()), 'dsl_comm_proxy')
        a.record()
        L.check(L.lib.dsl_comm_proxy(L.ptr(g), (hi - lo) // 4 * 4, wgs, passes, L.stream_ptr()), 'dsl_comm_proxy')
        b.record()
        torch.cuda.synchronize()
        alone.append(dict(mb=round((hi - lo) * 4 / 1e6, 1), ms=round(a.elapsed_time(b), 3)))
    med = {m: statistics.median(v) for m, v in res.items()}
    out = dict(ms_per_step={m: round(v, 3) for m, v in med.items()}, runs={m: [round(x, 3) for x in v] for m, v in res.items()},
               cost_frac={m: round(med[m] / med['none'] - 1.0, 4) for m in modes if m != 'none'},
               comm_stream_queue=int(L.lib.dsl_comm_stream_queue()), proxy=dict(workgroups=wgs, passes=passes, buckets_alone=alone),
               note='one GPU, no peer: dsl_comm_proxy (value-preserving passes over each gradient bucket) stands in for the all-reduce '
                    'behind the same bucket events; lib = the library\'s communication stream, placed on the hardware queue '
                    'comm_stream_queue names (1 weight gradients, 2 second chain, 3 frozen prefix, 4 caller, 0 = as the runtime dealt it), '
                    'torch = a torch-pool stream as ProcessGroupNCCL uses; *_eager = the round-5 schedule (exchange behind each backward '
                    'segment, joined at the end of the step) instead of the late exchange; cost_frac = median step time / median without proxy - 1')
    del model, opt
    torch.cuda.empty_cache()
    return out


class _ResidentLoader:
    """A loader (dsl_amd/data.py contract: __iter__ / __len__ yielding batch dicts) that yields ONE HBM-resident batch n times and
    stamps the wall clock (device drained) in front of batch `warm` and behind the last one."""

    def __init__(self, batch, n, warm):
        self.batch, self.n, self.warm, self.t0, self.t1 = batch, n, warm, None, None

    def __len__(self):
        return self.n

    def __iter__(self):
        from dsl_amd.data import mark_ready
        ev = torch.cuda.Event()
        ev.record()                            # the batch is resident in HBM: written (at the latest) by what is queued up to here
        for i in range(self.n):
            if i == self.warm:
                torch.cuda.synchronize()
                self.t0 = time.perf_counter()
            mark_ready(self.batch['img'], event=ev)      # the producer's event with every batch (dsl_amd.data.mark_ready is one-shot)
            yield self.batch
        torch.cuda.synchronize()
        self.t1 = time.perf_counter()


def train_detector_timing(batch, steps=20, warm=5):
    """The headline's step through the PRODUCT entry point: dsl_amd.apis.train_detector on the training sections of
    configs/fcos_semi/r50_caffe_mslonger_tricks_0.Xdata.py (restated here - the GPU box has no reference tree; tests/test_boundary_cpu.py
    builds the real file): SGD lr 0.01 momentum 0.9 wd 1e-4 bias_lr_mult 2 / bias_decay_mult 0, grad_clip None, step lr policy with
    linear warm-up, EpochBasedRunner, TextLoggerHook every 10 iterations (its host read of the log vars included), checkpoint hook
    (interval beyond the run).  Same model, batch and optimizer settings as the timed region of main()."""
    from dsl_amd import detectors  # noqa: F401
    from dsl_amd.apis import train_detector
    from dsl_amd.registry import Config, build_detector
    import tempfile
    _release_earlier_models()
    model = build_detector(model_cfg())
    loader = _ResidentLoader(batch, warm + steps, warm)
    with tempfile.TemporaryDirectory() as wd:
        cfg = Config(dict(
            model=model_cfg(), data=dict(samples_per_gpu=len(batch['img_metas']), workers_per_gpu=2),
            optimizer=dict(type='SGD', lr=0.01, momentum=0.9, weight_decay=0.0001, paramwise_cfg=dict(bias_lr_mult=2., bias_decay_mult=0.)),
            optimizer_config=dict(grad_clip=None),
            lr_config=dict(policy='step', warmup='linear', warmup_iters=500, warmup_ratio=1.0 / 3, step=[50, 80]),
            runner=dict(type='EpochBasedRunner', max_epochs=1), checkpoint_config=dict(interval=1000),
            log_config=dict(interval=10, hooks=[dict(type='TextLoggerHook')]), custom_hooks=[dict(type='NumClassCheckHook')],
            log_level='ERROR', load_from=None, resume_from=None, workflow=[('train', 1)], work_dir=wd))
        runner = train_detector(model, [loader], cfg, distributed=False, validate=False)
    dt = (loader.t1 - loader.t0) / steps
    det = runner._det(runner.model)
    out = dict(imgs_per_s=round(len(batch['img_metas']) / dt, 1), ms_per_step=round(dt * 1e3, 3),
               detector_flags=dict(lazy_log=det.lazy_log, eager_backward=det.eager_backward, pipeline_prefix=det.pipeline_prefix,
                                   deferred_head_update=bool(getattr(det.store, 'defer_head', False))),
               hooks=[type(h).__name__ for h in runner._hooks],
               note='dsl_amd.apis.train_detector + EpochBasedRunner + the config\'s hooks on the same resident batch; the lr schedule is '
                    'in its warm-up, every 10th iteration the logger reads the log vars back')
    del runner, model
    torch.cuda.empty_cache()
    return out


def datapath_timing(n_img=2, reps=20):
    """SURVEY.md section 8 row f3 beside the headline: the loader-side image work of one batch on the GPU - the labeled stream's
    pipeline (Resize -> PatchShuffle -> RandomFlip -> Normalize -> Pad: ONE launch) and the unlabeled stream's (the same +
    RandomAugmentBBox_Fast + UBAug: uint8 canvases, one launch per augmentation pass) - from decoded uint8 images already resident
    in HBM to the fp32 batch the detector takes; and the CPU restatement (oracle/datapath_oracle.py) on the same images."""
    import random
    from dsl_amd.datapath import GpuBatchPipeline
    from oracle import datapath_oracle as DO
    norm = dict(mean=[102.9801, 115.9465, 122.7717], std=[1.0, 1.0, 1.0], to_rgb=False)
    head = [dict(type='LoadImageFromFile'), dict(type='LoadAnnotations', with_bbox=True),
            dict(type='Resize', img_scale=[(1333, 640), (1333, 800)], multiscale_mode='value', keep_ratio=True),
            dict(type='PatchShuffle', ratio=0.5, ranges=[0.0, 1.0], mode=['flip', 'flop']), dict(type='RandomFlip', flip_ratio=0.5)]
    tail = [dict(type='Normalize', **norm), dict(type='Pad', size_divisor=32), dict(type='DefaultFormatBundle'),
            dict(type='Collect', keys=['img', 'gt_bboxes', 'gt_labels', 'gt_bboxes_ignore'])]
    rng = np.random.RandomState(7)
    samples = []
    for i in range(n_img):                      # COCO-sized sources (640 x 480 / 480 x 640)
        h, w = (480, 640) if i % 2 == 0 else (640, 480)
        b = synth_boxes(rng, 7)
        b[:, 0::2] = np.clip(b[:, 0::2] * w / 1333.0, 0, w - 1)
        b[:, 1::2] = np.clip(b[:, 1::2] * h / 800.0, 0, h - 1)
        samples.append(dict(img=torch.from_numpy(rng.randint(0, 256, (h, w, 3)).astype(np.uint8)).cuda(), gt_bboxes=b,
                            gt_labels=rng.randint(0, 80, len(b)), filename=f'synth{i}.jpg'))
    out = {}
    for name, pipe_cfg in (('labeled', head + tail), ('unlabeled', head + [dict(type='RandomAugmentBBox_Fast', aug_type='affine'),
                                                                              dict(type='UBAug')] + tail)):
        pipe = GpuBatchPipeline(pipe_cfg)
        np.random.seed(11)
        random.seed(11)
        for _ in range(3):
            pipe(samples)
        torch.cuda.synchronize()
        ts, passes, px = [], 0, 0
        for _ in range(reps):
            t0 = time.perf_counter()
            b = pipe(samples)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
            passes += max(len(p_) for p_ in pipe.last_passes)
            px += b['img'].shape[0] * b['img'].shape[2] * b['img'].shape[3]
        ts.sort()
        med = ts[len(ts) // 2]
        out[name] = dict(ms_per_batch=round(med * 1e3, 3), imgs_per_s=round(n_img / med, 1), mean_aug_launches=round(passes / reps, 1),
                         mean_canvas_mpix=round(px / reps / 1e6, 2))
    # the CPU restatement of the labeled stream's image work on the same two images (numpy, one thread)
    spec = [dict(img=s_['img'].cpu().numpy(), scale=(1333, 800), ps=None, flip=bool(i & 1)) for i, s_ in enumerate(samples)]
    t0 = time.perf_counter()
    DO.prepare_batch(spec, norm['mean'], norm['std'], norm['to_rgb'], 32)
    out['cpu_port_labeled'] = dict(ms_per_batch=round((time.perf_counter() - t0) * 1e3, 1), cores=1, kind='port')
    out['note'] = ('wall time per batch of %d images including the host-side parameter draws and box arithmetic, sources resident as uint8 in '
                   'HBM; algorithmic bytes of the labeled launch: 3 B read per source pixel + 12 B written per canvas pixel' % n_img)
    return out


def cpu_baseline(batch, seed=0, warmup=1, steps=2):
    """The CPU restatement of the SAME step (oracle/fcos_oracle.py, fp32 torch on the host cores), timed on a
    bounded sample: `warmup` untimed + `steps` timed full N=2 steps at 800x1344 (forward + loss + autograd backward +
    SGD), same weights and batch every step."""
    from oracle import fcos_oracle as O
    from dsl_amd.params import ParamStore
    torch.manual_seed(seed)
    store = ParamStore(80, 'cpu').init_reference_style(0)
    sd = {k: v.clone() for k, v in store.named_views().items()}
    img = batch['img'].cpu()
    cores = torch.get_num_threads()
    tk = O.trainable_keys(sd)
    times, losses = [], None
    for i in range(warmup + steps):
        t0 = time.perf_counter()
        losses, grads, _ = O.train_step(sd, img, batch['gt_bboxes'], batch['gt_labels'], None)
        O.sgd_step({k: sd[k] for k in tk}, grads, {}, first_step=True)
        if i >= warmup:
            times.append(time.perf_counter() - t0)
    n = img.shape[0]
    dt = sum(times) / len(times)
    return dict(value=n / dt, unit='imgs/s', cores=cores, kind='port',
                sample=f'{steps} training steps after {warmup} warm-up, {n} x (3,800,1344) fp32, torch CPU {cores} threads, '
                       + ', '.join(f'{t:.1f}' for t in times) + ' s'), losses


def dsl_iteration_timing(steps=12, warm=6, variants=None, extra_hook=None, raw=False):
    """BASELINE.json configs[2] beside the headline line: the semi-supervised iteration - labeled + unlabeled image and the
    half-scale copy (N = 3 through the student), ignore boxes, loss_weight 3, sisoft, clip, SGD, EMA teacher update every
    iteration - without and with the teacher's pseudo-label refresh of the upcoming unlabeled image (self-scheduled
    UnlabelPredHook, iteration mode)."""
    from dsl_amd.data import SyntheticSemiLoader
    from dsl_amd.optim import FlatSGD
    from dsl_amd.pseudo import PseudoLabelBank
    from dsl_amd.registry import build_detector
    from dsl_amd.runner import EMAOWNHook, OptimizerHook, SemiEpochBasedRunner, UnlabelPredHook
    out = {}
    for refresh, rla, asyn in variants or ((False, False, False), (True, False, False), (True, False, True), (False, True, False),
                                           (True, True, False), (True, True, True)):
        _release_earlier_models()
        student, teacher = build_detector(model_cfg(dsl=True, rla=rla)).cuda(), build_detector(model_cfg(dsl=True, rla=rla)).cuda()
        if rla:
            import warnings
            with warnings.catch_warnings():
                warnings.simplefilter('ignore')       # no pretrained checkpoint on the box: random init (reference style)
                student.init_weights()
                teacher.init_weights()
                teacher.load_state_dict(student.state_dict())
        opt = FlatSGD(student, lr=0.01, momentum=0.9, weight_decay=1e-4, paramwise_cfg=dict(bias_lr_mult=2., bias_decay_mult=0.),
                      grad_clip=dict(max_norm=35, norm_type=2))
        bank = PseudoLabelBank(num_classes=80, thres='adathres.json')
        n_it = warm + steps
        loader = SyntheticSemiLoader(bank, n_labeled=4, n_unlabeled=4, iters_per_epoch=n_it, H=800, W=1344, W_img=1333)
        runner = SemiEpochBasedRunner(student, optimizer=opt, max_epochs=1, ema_model=teacher, scale_invariant=True)
        runner.register_hook(OptimizerHook(grad_clip=dict(max_norm=35, norm_type=2)), priority=40)
        runner.register_hook(EMAOWNHook(interval=1, mode='iteration', ratio=0.99, start_point=0), priority=45)
        if refresh:
            if asyn:       # the loader asks one batch ahead (the reference: `preload` batches): the sweep runs beside the next step
                loader.unlabeled.prefetch_depth = 1
            hook = UnlabelPredHook(dict(infer_score_thre=0.1, use_ema=True, start_point=0, eval_config=dict(iou=[0.6]), async_sweep=asyn,
                                        eval_checkpoint_config=dict(interval=1, mode='iteration')), None, 'Det',
                                   interval_mode='iteration', interval=1, bank=bank)
            hook.iter_fuse_flag = True          # steady state: the initial full sweep is not part of an iteration's cost
            runner.register_hook(hook, priority=50)
        evs = []

        class Clock:
            priority = 90

            def __getattr__(self, name):
                return lambda r: None

            def after_train_iter(self, r):
                if r.iter + 1 >= warm:          # one event per iteration on the training stream (stream order = iteration order)
                    e = torch.cuda.Event(enable_timing=True)
                    e.record()
                    evs.append(e)
        runner.register_hook(Clock(), priority=90)
        if extra_hook is not None:           # (tools/dsl_outlier_probe.py: host-side bookkeeping per iteration)
            runner.register_hook(extra_hook, priority=95)
        runner.run([loader], max_epochs=1)
        torch.cuda.synchronize()
        # median of the per-iteration intervals: a one-off stall inside the window (first use of a new shape, an allocator
        # hiccup after the previous variant's teardown - the 13 <-> 18 ms spread of round 2's record) does not move it
        gaps = sorted(a.elapsed_time(b) for a, b in zip(evs[:-1], evs[1:]))
        dt = gaps[len(gaps) // 2] * 1e-3
        out.setdefault('spread', {})
        key = ('ms_per_iter' if not refresh else 'ms_per_iter_with_teacher_refresh' + ('_async' if asyn else '')) + ('_rla_backbone' if rla else '')
        out[key] = round(dt * 1e3, 3)
        out['spread'][key] = [round(gaps[0], 3), round(gaps[-1], 3)]           # min / max interval, ms
        if raw:
            out.setdefault('raw', {})[key] = [round(a.elapsed_time(b), 3) for a, b in zip(evs[:-1], evs[1:])]
        del student, teacher, runner, opt, loader
        torch.cuda.empty_cache()
    out['imgs_per_iter'] = 2
    out['note'] = ('N = 3 student step (labeled + unlabeled + half-scale copy, ignore boxes, loss_weight 3, sisoft, clip 35) + SGD + '
                   'EMA teacher every iteration; refresh = teacher sweep + fuse of the next unlabeled image every iteration; '
                   'images resident in HBM, the loader reads the refreshed labels back from the GPU before each batch; *_async: the loader asks one '
                   'batch ahead and the sweep runs on its own stream beside the next student step; *_rla_backbone: the '
                   'same iteration with the RLA_ResNet backbone of configs/fcos_semi/RLA_*.py')
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--imgs-per-gpu', type=int, default=2)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-dsl', action='store_true', help='skip the extra.dsl_iteration timing (configs[2]) after the timed region')
    ap.add_argument('--no-prof', action='store_true', help='do not bracket the conv kernels with HIP events')
    ap.add_argument('--prof-light', action='store_true', help='bracket only the dominant kernel class in the instrumented pass')
    ap.add_argument('--foreign-streams', default='', help="'before' / 'after' / 'before,after': the process creates and uses streams of its own "
                    "(as a DataLoader's copy stream or an evaluation hook would) before / after the model is built - the stream-layout robustness probe")
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', 0))
    local = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    assert world == args.gpus, f'--gpus {args.gpus} but WORLD_SIZE={world}'
    # test hooks (tests / dry runs on a 1-GPU box): DSL_BENCH_ONE_GPU=1 puts every rank on cuda:0, DSL_DIST_BACKEND=gloo
    torch.cuda.set_device(0 if os.environ.get('DSL_BENCH_ONE_GPU') else local)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group(os.environ.get('DSL_DIST_BACKEND', 'nccl'))

    from dsl_amd import _lib as L
    from dsl_amd import detectors  # noqa: F401
    from dsl_amd.optim import FlatSGD
    from dsl_amd.parallel import HipDistributedDataParallel
    from dsl_amd.registry import build_detector

    foreign = []

    def foreign_streams(n=3):
        """n streams from torch's pool, each USED (a small copy): a stream takes its hardware queue at first use."""
        src = torch.ones(1 << 16, device='cuda')
        for _ in range(n):
            st_ = torch.cuda.Stream()
            with torch.cuda.stream(st_):
                dst = src.clone()
            foreign.append((st_, dst))
        torch.cuda.synchronize()

    def foreign_tick():
        """What a loader's copy stream does beside the step: a small asynchronous copy per step on every foreign stream."""
        for st_, dst in foreign:
            with torch.cuda.stream(st_):
                dst.add_(1.0)
    if 'before' in args.foreign_streams:
        foreign_streams()
    model = build_detector(model_cfg()).cuda()      # random init, reference style, same seed on every rank
    # (no attribute is set here that dsl_amd.apis.train_detector does not set: lazy log vars, the eager backward and the pipelined
    # frozen prefix are the detector's defaults - the headline is the product path; extra.train_detector times the same step
    # through train_detector + the runner's hooks)
    if world > 1:
        model = HipDistributedDataParallel(model)
    det = model.module if world > 1 else model
    opt = FlatSGD(det, lr=0.01, momentum=0.9, weight_decay=1e-4, paramwise_cfg=dict(bias_lr_mult=2., bias_decay_mult=0.))
    batch = synth_batch(rank, args.imgs_per_gpu)
    from dsl_amd.data import mark_ready
    ready = torch.cuda.Event()
    ready.record()          # inputs are resident in HBM before the timed region: this is their producer's event; a loader hands it over
                            # with every batch (mark_ready is one-shot per batch).  (No stream of its own for the "loader": a new stream
                            # takes one of the four hardware queues and re-shuffles which of the step's streams share one - measured,
                            # profiles/r04_streams.txt: 350 instead of 425 img/s.)

    def step():
        if foreign:
            foreign_tick()
        mark_ready(batch['img'], event=ready)
        out = model.train_step(batch, opt)
        out['loss'].backward()
        opt.step()
        return out

    for _ in range(args.warmup):
        out = step()
    torch.cuda.synchronize()
    import gc
    gc.collect()
    gc.freeze()          # what dsl_amd.runner does a few iterations into a run (_gc_settle): no generation-2 walk over the op lists
    if 'after' in args.foreign_streams:
        foreign_streams()
        for _ in range(3):
            out = step()
        torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    def timed(k):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        o = None
        for i in range(k):
            o = step()
            if (i + 1) % 10 == 0:            # TextLoggerHook interval=10: one host read of the log vars
                _ = {kk: float(v) for kk, v in o['log_vars'].items()}
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        return time.perf_counter() - t0, o

    # (1) the timed region: EXACTLY --steps steps, no instrumentation
    dt, out = timed(args.steps)
    # (2) the same steps again with a HIP event pair around every launch of the dominant kernel (on its launch
    # stream).  Kept out of region (1) because the event records serialise the concurrently running streams
    # (measured: -6 % throughput with one class, -12 % with every conv / wgrad launch bracketed).
    dt_prof = None
    if not args.no_prof:
        if world > 1:
            dist.barrier()
        # (2a) phases only: two event pairs per step around the head's forward / backward data path (both towers on two streams);
        # the step is otherwise unperturbed
        NC = L.PROF_CLASSES
        L.lib.dsl_prof_reset()
        L.lib.dsl_prof_enable(3)
        timed(args.steps)
        L.lib.dsl_prof_enable(0)
        ph = [(C.c_int64 * NC)(), (C.c_double * NC)(), (C.c_double * NC)(), (C.c_double * NC)()]
        L.lib.dsl_prof_read2(*ph)
        L.lib.dsl_prof_reset()
        L.lib.dsl_prof_enable(1 if args.prof_light else 2)
        dt_prof, _ = timed(args.steps)
        L.lib.dsl_prof_enable(0)
    if world > 1:
        t = torch.tensor([dt], device='cuda')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
    log = {k: float(v) for k, v in out['log_vars'].items()}

    n_total = args.imgs_per_gpu * world * args.steps
    value = n_total / dt
    roof = None
    # HBM bytes per launch of the dominant kernel: PMC counters cannot be read from inside the process, so this is the
    # committed result of the separate rocprofv3 --pmc passes of this same command (tools/pmc_traffic.py)
    traffic = traffic_src = step_traffic = None
    tf = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles', 'traffic.json')
    if os.path.exists(tf):
        tj = json.load(open(tf))
        step_traffic = tj.get('step')          # whole-step HBM bytes (sum over every kernel of the PMC passes) and the floor they imply
        hit = [k for k in tj.get('kernels', {}) if k.startswith(DOMINANT[:-1])]     # the symbol carries further template args
        if hit:
            traffic = tj['kernels'][hit[0]]['hbm_bytes_per_launch']
            traffic_src = 'profiles/traffic.json: ' + tj.get('source', '')
    # MFMA-pipe busy fraction of the kernel classes from the SQ counters of the same separate --pmc pass (profiles/pmc_sq.json)
    mfma_busy = None
    sf = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles', 'pmc_sq.json')
    if os.path.exists(sf):
        sj = json.load(open(sf)).get('kernels', {})
        pick = lambda pre: next((v for k, v in sj.items() if k.startswith(pre)), None)
        mfma_busy = {name: (dict(mfma_busy_frac=v['mfma_busy_frac'], wait_any_frac=v['wait_any_frac'], avg_us=v['avg_us']) if v else None)
                     for name, v in (('conv_128x128', pick(DOMINANT[:-1])), ('head_tile_256x192', pick('conv_pipe_kernel<256, 192')),
                                     ('wgrad_256x256', pick('wgrad_pipe_kernel<256, 256')))}
        mfma_busy['source'] = 'profiles/pmc_sq.json (separate rocprofv3 --pmc pass of this command; SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CYCLES x 32))'
    if not args.no_prof:
        launches = (C.c_int64 * NC)()
        ms = (C.c_double * NC)()
        fl = (C.c_double * NC)()
        by = (C.c_double * NC)()
        L.lib.dsl_prof_read2(launches, ms, fl, by)

        def cls(i):
            return dict(achieved=round(fl[i] / (ms[i] * 1e-3) / 1e12, 1) if launches[i] else None,
                        avg_launch_us=round(ms[i] * 1e3 / max(launches[i], 1), 2), launches_per_step=launches[i] // args.steps,
                        ms_per_step=round(ms[i] / args.steps, 3),
                        algorithmic_mb_per_launch=round(by[i] / max(launches[i], 1) / 1e6, 2))
        if launches[0]:
            ach = fl[0] / (ms[0] * 1e-3) / 1e12
            gbs = by[0] / (ms[0] * 1e-3) / 1e9
            intensity = fl[0] / by[0]
            # Which roof?  By arithmetic intensity the class sits below the ridge (2500 TFLOP/s / 8 TB/s = 312 FLOP per algorithmic
            # byte), i.e. HBM would bind it - but a class that reaches < 25 % of BOTH roofs is bound by neither: its launches are
            # 66-132 workgroups of 4-36 K tiles, i.e. launch, pipeline fill, epilogue and kernel boundary.  The
            # review's convention ("MFMA for convs"): the MFMA view is the primary `frac` of a convolution class,
            # `bound` says "latency" in that case, and the HBM view stays beside it.
            hbm_roof = intensity < PEAK_BF16_TFLOPS * 1e12 / (PEAK_HBM_GBS * 1e9)
            latency = ach / PEAK_BF16_TFLOPS < 0.25 and gbs / PEAK_HBM_GBS < 0.25
            hbm = False
            roof = dict(bound='latency' if latency else ('hbm' if hbm_roof else 'mfma'),
                        roof_by_arithmetic_intensity='hbm' if hbm_roof else 'mfma',
                        kernel=DOMINANT + ' (forward + data-gradient implicit GEMM, 8-wave 128x128 '
                        'tile: the backbone / predictor convolutions; largest share of the step)',
                        achieved=round(gbs if hbm else ach, 1), peak=PEAK_HBM_GBS if hbm else PEAK_BF16_TFLOPS,
                        unit='GB/s' if hbm else 'TFLOP/s', frac=round(gbs / PEAK_HBM_GBS if hbm else ach / PEAK_BF16_TFLOPS, 4),
                        traffic=traffic, traffic_source=traffic_src,
                        hbm_bytes_per_step=(step_traffic or {}).get('hbm_bytes_per_step'),
                        hbm_floor_ms=(step_traffic or {}).get('hbm_floor_ms_at_8TBs'),
                        hbm_floor_ms_at_measured_6p3TBs=(step_traffic or {}).get('hbm_floor_ms_at_6p3TBs'), measured='HIP event pairs on the launch stream, second pass '
                        'of the same %d steps (%.3f ms/step while instrumented)' % (args.steps, dt_prof / args.steps * 1e3),
                        launches_per_step=launches[0] // args.steps,
                        avg_launch_us=round(ms[0] * 1e3 / launches[0], 2),
                        algorithmic_gflop_per_launch=round(fl[0] / launches[0] / 1e9, 3),
                        algorithmic_bytes_per_launch=round(by[0] / launches[0]),
                        flop_per_algorithmic_byte=round(intensity, 1),
                        mfma_tflops=round(ach, 1), mfma_frac=round(ach / PEAK_BF16_TFLOPS, 4),
                        hbm_gbs_algorithmic=round(gbs, 1), hbm_frac_algorithmic=round(gbs / PEAK_HBM_GBS, 4),
                        head_tile_256x192=cls(1), other_conv_kernels=cls(2), wgrad_kernels=cls(3),
                        head_phases={name: dict(tflops=round(ph[2][c] / (ph[1][c] * 1e-3) / 1e12, 1), ms_per_step=round(ph[1][c] / args.steps, 3),
                                                frac_of_mfma_peak=round(ph[2][c] / (ph[1][c] * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4),
                                                gflop_per_step=round(ph[2][c] / args.steps / 1e9, 1))
                                     for name, c in (('forward', 4), ('backward_data_path', 5)) if ph[0][c]} or None,
                        head_phases_note='wall time on the caller\'s stream of the head (16 tower convs + 2 predictors + 8 GroupNorm '
                                         'passes per direction, the two towers side by side on two streams), uninstrumented kernels; '
                                         'the per-launch figures above stretch when two launches overlap',
                        mfma_busy_frac=mfma_busy,
                        whole_step_frac=round(value / world * GFLOP_PER_IMAGE_STEP / 1e3 / PEAK_BF16_TFLOPS, 4))
    cpu = extra = None
    if world > 1:
        # communication trace (outside the timed region): per gradient bucket, when its all-reduce was issued behind the bucket's
        # named event and when it was complete, in ms from the start of the step - what the scaling efficiency hinges on
        det.comm_trace = []
        for _ in range(2):
            step()
        ev_end = torch.cuda.Event(enable_timing=True)
        ev_end.record()
        torch.cuda.synchronize()
        tr = det.comm_trace[-1]
        det.comm_trace = None
        bk = [dict(mb=round(b['mb'], 1), start_ms=round(tr['t0'].elapsed_time(b['start']), 3),
                   done_ms=round(tr['t0'].elapsed_time(b['done']), 3) if b['done'] is not None else None) for b in tr['buckets']]
        # attribution: the same steps with every collective skipped (gradients then wrong - timing only, after everything that is reported)
        devs = [None] * world
        dist.all_gather_object(devs, dict(rank=rank, device=torch.cuda.current_device(), name=torch.cuda.get_device_name()))
        # a scaling number is only one if every rank had a GPU of its own and the communicator spans them all: fail loudly otherwise
        # (the 1-GPU dry run of tests/test_ddp_gpu.py puts every rank on cuda:0 on purpose: DSL_BENCH_ONE_GPU)
        n_comm = int(L.lib.dsl_comm_size(det.rccl.comm)) if det.rccl is not None else dist.get_world_size()
        if n_comm != world:
            raise RuntimeError(f'bench.py --gpus {world}: the communicator has {n_comm} ranks')
        if not os.environ.get('DSL_BENCH_ONE_GPU') and len({d['device'] for d in devs}) != world:
            raise RuntimeError(f'bench.py --gpus {world}: ranks share devices {[d["device"] for d in devs]} - one process per GPU expected')
        det.comm_off = True
        for _ in range(2):
            step()
        dt_off, _ = timed(max(5, args.steps // 2))
        det.comm_off = False
        t_off = torch.tensor([dt_off], device='cuda')
        dist.all_reduce(t_off, op=dist.ReduceOp.MAX)
        order = 'layer4, layer3, layer2, head+FPN (deferred: under the next forward pass)' if getattr(det.store, 'defer_head', False) \
            else 'head+FPN, layer4, layer3, layer2'
        if getattr(det, 'late_exchange', False) and det.clip_partials is None:
            order = ('layer2, layer3, layer4, head+FPN - late exchange: queued behind the backward pass, each bucket in front of its update, the '
                     'next forward pass waits per stage (DESIGN section 6)')
        extra = dict(comm=dict(carrier='C-ABI rcclComm_t (dsl_allreduce_bucket)' if det.rccl is not None else 'torch.distributed process group',
                               backend=dist.get_backend(), rccl_ranks=int(L.lib.dsl_comm_size(det.rccl.comm)) if det.rccl is not None else dist.get_world_size(),
                               devices=devs, grad_dtype='bf16' if det.grad_bf16 else 'fp32',
                               exchange='late' if (getattr(det, 'late_exchange', False) and det.clip_partials is None) else 'eager',
                               comm_stream_queue=int(L.lib.dsl_comm_stream_queue()),
                               wgrad_slots=_lib_option('wgrad_slots'),
                               buckets=bk, step_ms=round(tr['t0'].elapsed_time(ev_end), 3),
                               ms_per_step_comm_disabled=round(float(t_off) / max(5, args.steps // 2) * 1e3, 3),
                               ms_per_step=round(dt / args.steps * 1e3, 3),
                               note='rank 0; bucket order ' + order + '; the optimizer updates a bucket as soon as its all-reduce is done '
                                    '(per-bucket SGD), so only traffic still in flight at step_ms is exposed; ms_per_step_comm_disabled = the '
                                    'same loop with every collective skipped (max over ranks): the difference to ms_per_step is what '
                                    'communication costs, the difference to the 1-GPU figure is what running beside other ranks costs'))
    layout = None
    if rank == 0 and world == 1 and not args.no_dsl and not foreign:
        # stream-layout robustness (which streams share a hardware queue is a 20 % variable): the same loop again with
        # three foreign streams of torch's pool created now and used in every step, as a loader's copy stream would be
        dt_a, _ = timed(args.steps)
        foreign_streams()
        for _ in range(3):
            step()
        dt_b, _ = timed(args.steps)
        layout = dict(ms_per_step_clean=round(dt_a / args.steps * 1e3, 3), ms_per_step_with_foreign_streams=round(dt_b / args.steps * 1e3, 3),
                      ratio=round(dt_b / dt_a, 4), streams_on_own_queues=_streams_probe(),
                      note="three torch.cuda.Stream()s created after the model and used (a small op each) in every step; the library's own "
                           'streams are created AND picked by the C library (csrc/api.hip side_init: a spin-kernel probe finds three on hardware queues of their own); tests/test_stream_layout_gpu.py runs '
                           'the before / after variants in fresh processes')
        foreign.clear()
    gc.unfreeze()        # (the extras below build and drop models of their own; their runs freeze and unfreeze by themselves)
    if rank == 0 and world == 1 and not args.no_dsl:
        del opt
        # (train_detector last: every model instance creates streams, and stream creation order decides which streams share one of
        # the four hardware queues - DESIGN 3.2i; the semi-supervised variants keep the order their round-3 numbers were taken in)
        extra = dict(dsl_iteration=dsl_iteration_timing(), fp8_towers=fp8_step_timing(batch), datapath=datapath_timing())
        extra['train_detector'] = train_detector_timing(batch, steps=60, warm=10)
        extra['comm_proxy'] = comm_proxy_timing(batch)
        extra['stream_layout_check'] = layout
        if roof is not None:        # BASELINE.json configs[2] beside the headline, also where a record that keeps `roofline` keeps it
            roof['configs2_dsl_iteration_ms'] = {k: v for k, v in extra['dsl_iteration'].items() if k.startswith('ms_per_iter')}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu, _ = cpu_baseline(batch)
    if rank == 0:
        print(json.dumps({
            'metric': 'imgs/sec training step, FCOS R50-FPN 1333x800', 'value': round(value, 2), 'unit': 'imgs/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(dt / args.steps * 1e3, 3),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'bf16', 'data': 'synthetic',
            'config': {'workload': 'FCOS R50-caffe-FPN supervised training step (BASELINE.json configs[1]), '
                                   f'{args.imgs_per_gpu} x (3,800,1344) per GPU, synthetic COCO-shaped boxes, '
                                   'random-init weights', 'global_batch': args.imgs_per_gpu * world,
                       'parallelism': f'dp{world}', 'optimizer': 'SGD momentum 0.9 wd 1e-4'},
            'roofline': roof, 'cpu_baseline': cpu, 'extra': extra, 'final_losses': log,
            'streams_on_own_queues': _streams_probe(), 'comm_stream_queue': int(L.lib.dsl_comm_stream_queue())}))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
