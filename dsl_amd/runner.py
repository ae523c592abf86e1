"""Semi-supervised runner and hooks with the reference's registry names, driving the HIP step.

  SemiEpochBasedRunner   mmdet/runner/hooks/semi_epoch_based_runner.py:49-509
  EMAOWNHook             mmdet/runner/hooks/ema.py:4-44
  UnlabelPredHook        mmdet/runner/hooks/unlabel_pred_hook.py:370-562 (+ adathres :295-367,
                         parse_det_results :20-38, SemiCOCODataset._parse_ann_info datasets/semicoco.py:232-264)
  OptimizerHook / LrUpdaterHook (step + linear warm-up)   mmcv 1.3.10 semantics, configs/fcos_semi/*.py

What changes w.r.t. the reference: the scale-invariant third image is built on the
GPU; EMA is one fused lerp over the flat parameter buffer (no deepcopy, no barriers); the pseudo-label
refresh runs the teacher and its post-processing on the GPU and keeps the labels in an in-memory
bank (the reference's per-image JSON layout is kept as an optional export)."""
import gc
import json
import os
import time

import numpy as np
import torch
import torch.nn.functional as F

from . import _lib as L
from .optim import build_optimizer  # noqa: F401
from .registry import HOOKS, RUNNERS


class Hook:
    def every_n_iters(self, runner, n):
        return (runner.iter + 1) % n == 0 if n > 0 else False

    def every_n_epochs(self, runner, n):
        return (runner.epoch + 1) % n == 0 if n > 0 else False


for _m in ('before_run', 'after_run', 'before_train_epoch', 'after_train_epoch', 'before_train_iter', 'after_train_iter'):
    setattr(Hook, _m, lambda self, runner: None)


def append_half_scale(img, gt_bboxes, gt_labels, gt_bboxes_ignore, img_metas, materialize=True):
    """semi_epoch_based_runner.py:186-204, on device: bilinear half-size copy of the LAST image pasted at the
    top-left of a zero canvas; boxes and ignore boxes halved; meta shapes halved (int).
    materialize=False: `img` is returned as it came (the lists still get their extra entry): the detector's stem kernel samples
    the copy from img[-1] (FCOS.forward_train(half_scale_copy=True))."""
    B, _, H, W = img.shape
    if materialize:
        small = F.interpolate(img[B - 1:], size=(int(H / 2), int(W / 2)), mode='bilinear')
        canvas = torch.zeros_like(img[B - 1:])
        canvas[:, :, :small.shape[2], :small.shape[3]] = small
        img = torch.cat([img, canvas], 0)
    meta = dict(img_metas[-1])
    for k in ('img_shape', 'pad_shape'):
        if k in meta:
            s = meta[k]
            meta[k] = (int(s[0] / 2), int(s[1] / 2)) + tuple(s[2:])
    if 'scale_factor' in meta:
        meta['scale_factor'] = np.asarray(meta['scale_factor']) / 2
    gt_bboxes = list(gt_bboxes) + [gt_bboxes[-1].clone() / 2]
    gt_labels = list(gt_labels) + [gt_labels[-1].clone()]
    if gt_bboxes_ignore is not None:
        last = gt_bboxes_ignore[-1].clone()
        gt_bboxes_ignore = list(gt_bboxes_ignore) + [last / 2 if len(last) > 0 else last]
    return img, gt_bboxes, gt_labels, gt_bboxes_ignore, list(img_metas) + [meta]


@RUNNERS.register_module()
class SemiEpochBasedRunner:
    def __init__(self, model, batch_processor=None, optimizer=None, work_dir=None, logger=None, meta=None,
                 max_iters=None, max_epochs=None, ema_model=None, scale_invariant=False):
        assert batch_processor is None and hasattr(model, 'train_step')
        self.model, self.optimizer, self.work_dir, self.logger, self.meta = model, optimizer, work_dir, logger, meta
        self._max_epochs, self._max_iters = max_epochs, max_iters
        self.ema_model = ema_model
        self.ema_flag = False
        self.scale_invariant = scale_invariant
        self._hooks = []
        self._epoch = self._iter = self._inner_iter = 0
        self.outputs = None
        self.log_buffer = []
        self.mode = None

    epoch = property(lambda self: self._epoch)
    iter = property(lambda self: self._iter)
    inner_iter = property(lambda self: self._inner_iter)
    max_epochs = property(lambda self: self._max_epochs)

    def _det(self, m):
        return m.module if hasattr(m, 'module') else m

    PRIORITIES = dict(HIGHEST=0, VERY_HIGH=10, HIGH=30, ABOVE_NORMAL=40, NORMAL=50, BELOW_NORMAL=60, LOW=70, VERY_LOW=90,
                      LOWEST=100)         # mmcv/runner/priority.py

    def register_hook(self, hook, priority=50):
        if isinstance(priority, str):
            priority = self.PRIORITIES[priority.upper()]
        hook.priority = priority
        i = len(self._hooks)
        while i > 0 and self._hooks[i - 1].priority > priority:
            i -= 1
        self._hooks.insert(i, hook)

    def call_hook(self, name):
        for h in self._hooks:
            getattr(h, name)(self)

    # -- config-driven hook registration (semi_epoch_based_runner.py:460-509, mmcv BaseRunner.register_*_hook) ----
    def register_training_hooks(self, lr_config, optimizer_config=None, ema_config=None, checkpoint_config=None,
                                log_config=None, momentum_config=None, timer_config=None, custom_hooks_config=None):
        if lr_config is not None:
            if isinstance(lr_config, dict):
                c = dict(lr_config)
                policy = c.pop('policy')
                assert policy == 'step', f'lr policy {policy!r}: only the step policy of configs/fcos_semi is built'
                hook = StepLrUpdaterHook(**c)
            else:
                hook = lr_config
            self.register_hook(hook, priority='VERY_HIGH')
        assert momentum_config is None, 'momentum schedules are not used by configs/fcos_semi'
        if optimizer_config is not None:
            hook = OptimizerHook(**optimizer_config) if isinstance(optimizer_config, dict) else optimizer_config
            self.register_hook(hook, priority='ABOVE_NORMAL')
        if ema_config is not None:
            if isinstance(ema_config, dict):
                c = dict(ema_config)
                c.setdefault('type', 'EMAOWNHook')
                hook = HOOKS.build(c)
            else:
                hook = ema_config
            self.register_hook(hook, priority=45)
        if checkpoint_config is not None:
            hook = CheckpointHook(**checkpoint_config) if isinstance(checkpoint_config, dict) else checkpoint_config
            self.register_hook(hook, priority='NORMAL')
        if log_config is not None:
            interval = log_config.get('interval', 10)
            for info in log_config.get('hooks', []):
                c = dict(info)
                if c['type'] not in HOOKS:
                    raise KeyError(f"logger hook {c['type']} is not available")
                c.setdefault('interval', interval)
                self.register_hook(HOOKS.build(c), priority='VERY_LOW')
        for c in custom_hooks_config or []:
            c = dict(c)
            prio = c.pop('priority', 'NORMAL')
            self.register_hook(HOOKS.build(c), priority=prio)

    def current_lr(self):
        return [g['lr'] for g in self.optimizer.param_groups]

    def resume(self, checkpoint, resume_optimizer=True, map_location='cpu'):
        """mmcv BaseRunner.resume: weights (student and teacher, :350-366), epoch / iter counters, optimizer state."""
        ck = self.load_checkpoint(checkpoint, map_location=map_location)
        self._epoch = ck['meta']['epoch']
        self._iter = ck['meta']['iter']
        if 'optimizer' in ck and resume_optimizer and self.optimizer is not None:
            self.optimizer.load_state_dict(ck['optimizer'])
        ema = checkpoint + '_ema'
        if self.ema_model is not None and os.path.exists(ema):
            self._det(self.ema_model).load_state_dict(torch.load(ema, map_location=map_location)['state_dict'], strict=False)
            self.ema_flag = True
        if self.logger:
            self.logger.info('resumed epoch %d, iter %d', self.epoch, self.iter)

    def run_iter(self, data_batch, train_mode=True, **kw):
        m = self.model if train_mode else (self.ema_model if self.ema_flag else self.model)
        outputs = m.train_step(data_batch, self.optimizer, **kw) if train_mode else m.val_step(data_batch, self.optimizer, **kw)
        if not isinstance(outputs, dict):
            raise TypeError('model.train_step() must return a dict')
        if 'log_vars' in outputs:
            self.log_buffer.append((outputs['log_vars'], outputs['num_samples']))
            self.log_buffer = self.log_buffer[-50:]
        self.outputs = outputs

    def train(self, data_loader, **kw):
        self.mode = 'train'
        self.data_loader = data_loader
        self._max_iters = self._max_epochs * len(data_loader) if self._max_epochs else self._max_iters
        self.iter_tol_epoch = len(data_loader)
        self.call_hook('before_train_epoch')
        for i, data_batch in enumerate(data_loader):
            self._inner_iter = i
            if self.scale_invariant:
                d = data_batch
                det = getattr(self.model, 'module', self.model)
                im = d['img']
                # the HIP detector reads the half-scale copy out of the last image inside its stem kernel: no interpolate / zeros /
                # cat passes on the training stream, the loader's own tensor (and its mark_ready event) reaches forward_train
                in_stem = (getattr(det, 'half_scale_in_stem', False) and torch.is_tensor(im) and im.is_cuda
                           and im.shape[2] % 2 == 0 and im.shape[3] % 2 == 0)
                img, gb, gl, gi, metas = append_half_scale(im, d['gt_bboxes'], d['gt_labels'],
                                                           d.get('gt_bboxes_ignore'), d['img_metas'], materialize=not in_stem)
                data_batch = dict(d, img=img, gt_bboxes=gb, gt_labels=gl, img_metas=metas)
                if in_stem:
                    data_batch['half_scale_copy'] = True
                if gi is not None:
                    data_batch['gt_bboxes_ignore'] = gi
            self.call_hook('before_train_iter')
            self.run_iter(data_batch, train_mode=True, **kw)
            self.call_hook('after_train_iter')
            self._iter += 1
            self._gc_settle()
        self.call_hook('after_train_epoch')
        self._epoch += 1

    GC_FREEZE_AFTER = 5        # training iterations of a run after which the long-lived heap is frozen (see _gc_settle)

    def _gc_settle(self):
        """The engine's op lists, descriptors, plans and the modules are some 10^5 long-lived Python objects.  A generation-2
        collection that walks them stalls the host for tens of milliseconds - longer than its 2 - 3 ms lead over the GPU, so the GPU
        idles (the 19 / 55 ms intervals of `extra.dsl_iteration.spread` in rounds 4 / 5; the collection below itself takes 40 - 100 ms
        on this heap: profiles/r05_bench_full_gcfreeze_in_window.log, where it fell into bench.py's timed window).  Once everything
        is built - the plans are through by iteration 4 - they move to the permanent generation; run() undoes it."""
        self._gc_iters = getattr(self, '_gc_iters', 0) + 1
        if self.GC_FREEZE_AFTER <= 0:          # (0 disables the freeze)
            return
        if self._gc_iters == self.GC_FREEZE_AFTER and not getattr(self, '_gc_frozen', False):
            gc.collect()
            gc.freeze()
            self._gc_frozen = True

    def run(self, data_loaders, workflow=(('train', 1),), max_epochs=None, **kw):
        if max_epochs is not None:
            self._max_epochs = max_epochs
        self._gc_iters = 0
        self.call_hook('before_run')
        try:
            while self._epoch < self._max_epochs:
                for (mode, epochs), loader in zip(workflow, data_loaders):
                    assert mode == 'train'
                    for _ in range(epochs):
                        if self._epoch >= self._max_epochs:
                            break
                        self.train(loader, **kw)
        finally:
            # also when train() or a hook raises: a heap left frozen would keep every later cycle of the process uncollected
            if getattr(self, '_gc_frozen', False):
                gc.unfreeze()
                self._gc_frozen = False
        self.call_hook('after_run')

    @torch.no_grad()
    def EMA(self, keep_rate=0.1, mode='epoch', start_point=5):
        """semi_epoch_based_runner.py:368-409: teacher = keep*teacher + (1-keep)*student over the whole
        state_dict.  One fused pass over the flat trainable buffer; the frozen tensors (stem, layer1, BN
        statistics) are identical in student and teacher, so their lerp is the identity and is skipped; the
        int64 num_batches_tracked entries stay 0 (norm_eval).  Stream-ordered: no barriers needed."""
        s, t = self._det(self.model).store, self._det(self.ema_model).store
        if t.device != s.device:
            t.to(s.device)
        s.wait_pending()          # (a deferred head update of the optimizer step just taken)
        ev = getattr(self, '_sweep_event', None)          # an asynchronous teacher sweep (UnlabelPredHook.async_sweep) still reads the
        if ev is not None:                                # teacher's weights: the update waits for it
            torch.cuda.current_stream().wait_event(ev)
            self._sweep_event = None
        if t.train16 is None or t.dirty:
            L.check(L.lib.dsl_ema_lerp(L.ptr(t.train), L.ptr(s.train), s.n_train, float(keep_rate), L.stream_ptr()), 'dsl_ema_lerp')
            t.refresh()
        else:
            # one pass: the lerp and the teacher's bf16 forward copy (384 + 64 MB instead of 384 + 192)
            L.check(L.lib.dsl_ema_lerp_bf16(L.ptr(t.train), L.ptr(s.train), L.ptr(t.train16), s.n_train, float(keep_rate), L.stream_ptr()),
                    'dsl_ema_lerp_bf16')
            # RLA_ResNet keeps the affine parameters of its eval-mode BatchNorms trainable: the teacher's forward reads the
            # FOLDED (scale, bias), so they are re-made from the lerped gamma / beta (no-op for the plain ResNet)
            t.refold_bn(L.stream_ptr())
        self.ema_flag = True

    def save_checkpoint(self, out_dir, filename_tmpl='epoch_{}.pth', save_optimizer=True, meta=None, create_symlink=True):
        """:411-458: student (+ optimizer state) to <file>, teacher to <file>_ema once it exists (ema_flag), latest.pth."""
        os.makedirs(out_dir, exist_ok=True)
        meta = dict(meta or {})
        if self.meta is not None:
            meta.update(self.meta)
        meta.update(epoch=self.epoch + 1, iter=self.iter)
        filename = filename_tmpl.format(self.epoch + 1)
        fn = os.path.join(out_dir, filename)
        ck = dict(meta=meta, state_dict={k: v.detach().cpu().clone() for k, v in self._det(self.model).state_dict().items()})
        if save_optimizer and self.optimizer is not None:
            osd = self.optimizer.state_dict()
            ck['optimizer'] = {k: (v.detach().cpu().clone() if isinstance(v, torch.Tensor) else v) for k, v in osd.items()}
        torch.save(ck, fn)
        if self.ema_flag and self.ema_model is not None:
            sd = {k: v.detach().cpu().clone() for k, v in self._det(self.ema_model).state_dict().items()}
            torch.save(dict(meta=meta, state_dict=sd), fn + '_ema')
        if create_symlink:
            dst = os.path.join(out_dir, 'latest.pth')
            if os.path.lexists(dst):
                os.remove(dst)
            try:
                os.symlink(filename, dst)
            except OSError:
                import shutil
                shutil.copy(fn, dst)
        return fn

    def load_checkpoint(self, filename, map_location='cpu', strict=False):
        """:350-366: the same file goes into both student and teacher."""
        ck = torch.load(filename, map_location=map_location)
        sd = ck.get('state_dict', ck)
        self._det(self.model).load_state_dict(sd, strict=strict)
        if self.ema_model is not None:
            self._det(self.ema_model).load_state_dict(sd, strict=strict)
        return ck


RUNNERS.register_module(name='EpochBasedRunner', module=SemiEpochBasedRunner)     # the supervised config's runner: same loop, no teacher


@HOOKS.register_module()
class OptimizerHook(Hook):
    """mmcv OptimizerHook: zero_grad; loss.backward(); [clip]; step.  Clipping is fused into the SGD kernel."""

    def __init__(self, grad_clip=None, **kw):
        self.grad_clip = grad_clip

    def after_train_iter(self, runner):
        opt = runner.optimizer
        if self.grad_clip and getattr(opt, 'max_norm', None) is None:
            opt.max_norm = float(self.grad_clip['max_norm'])
        opt.zero_grad()
        runner.outputs['loss'].backward()
        opt.step()


@HOOKS.register_module()
class StepLrUpdaterHook(Hook):
    """policy='step', by_epoch, gamma 0.1, linear warm-up from warmup_ratio over warmup_iters."""

    def __init__(self, step, gamma=0.1, warmup=None, warmup_iters=0, warmup_ratio=0.1, by_epoch=True, **kw):
        self.step = [step] if isinstance(step, int) else list(step)
        self.gamma, self.warmup, self.warmup_iters, self.warmup_ratio = gamma, warmup, warmup_iters, warmup_ratio

    def _factor(self, runner):
        f = self.gamma ** sum(1 for s in self.step if runner.epoch >= s)
        if self.warmup == 'linear' and runner.iter < self.warmup_iters:
            k = (1 - runner.iter / self.warmup_iters) * (1 - self.warmup_ratio)
            f = f * (1 - k)
        return f

    def before_train_iter(self, runner):
        f = self._factor(runner)
        for g in runner.optimizer.param_groups:
            g['lr'] = g['initial_lr'] * f


@HOOKS.register_module()
class EMAOWNHook(Hook):
    """runner/hooks/ema.py:4-44."""

    def __init__(self, interval=-1, mode='epoch', ratio=0.99, start_point=-1, step_decay=None, decay_ratio=0.1, **kw):
        self.interval, self.mode, self.start_point, self.ratio = interval, mode, start_point, ratio
        self.step_decay, self.decay_ratio, self.args = step_decay, decay_ratio, kw

    def after_train_epoch(self, runner):
        if self.step_decay is not None and runner.epoch + 1 in self.step_decay:
            self.ratio = max(1.0 - (1.0 - self.ratio) / self.decay_ratio, 0.01)
        if self.mode != 'epoch' or self.interval == -1 or self.start_point > runner.epoch + 1:
            return
        if self.every_n_epochs(runner, self.interval):
            runner.EMA(keep_rate=self.ratio, mode=self.mode, start_point=self.start_point, **self.args)

    def after_train_iter(self, runner):
        if self.mode != 'iteration' or self.interval == -1 or self.start_point > runner.iter + 1:
            return
        if self.every_n_iters(runner, self.interval):
            runner.EMA(keep_rate=self.ratio, mode=self.mode, start_point=self.start_point, **self.args)


@HOOKS.register_module()
class CheckpointHook(Hook):
    """mmcv CheckpointHook (by_epoch): runner.save_checkpoint every `interval` epochs into work_dir, rank 0 only."""

    def __init__(self, interval=-1, by_epoch=True, save_optimizer=True, out_dir=None, max_keep_ckpts=-1, **kw):
        self.interval, self.by_epoch, self.save_optimizer, self.out_dir = interval, by_epoch, save_optimizer, out_dir

    def _rank0(self):
        import torch.distributed as dist
        return not (dist.is_available() and dist.is_initialized()) or dist.get_rank() == 0

    def after_train_epoch(self, runner):
        if not self.by_epoch or not self.every_n_epochs(runner, self.interval):
            return
        out = self.out_dir or runner.work_dir
        if out and self._rank0():
            runner.save_checkpoint(out, save_optimizer=self.save_optimizer)


@HOOKS.register_module()
class TextLoggerHook(Hook):
    """mmcv TextLoggerHook, reduced: every `interval` iterations one line with epoch / iter / lr and the log vars
    averaged over the interval (the only host read of the step's device scalars), also appended to
    <work_dir>/<timestamp>.log.json."""

    def __init__(self, interval=10, by_epoch=True, **kw):
        self.interval, self.by_epoch = interval, by_epoch

    def after_train_iter(self, runner):
        if not self.every_n_iters(runner, self.interval):
            return
        buf = runner.log_buffer[-self.interval:]
        if not buf:
            return
        keys = list(buf[0][0].keys())
        # ONE device-to-host read for the whole interval (lazy log vars are 0-dim device tensors: float() on each of interval x keys
        # of them is as many stream synchronisations - 40 per line at the default interval, ≈ 2 % of the supervised step's throughput)
        flat = [lv[k] for lv, _ in buf for k in keys]
        if flat and all(torch.is_tensor(v) and v.is_cuda for v in flat):
            host = torch.stack([v.detach().reshape(()) for v in flat]).tolist()
        else:
            host = [float(v) for v in flat]
        nk = len(keys)
        tot = sum(n for _, n in buf)
        avg = {k: float(sum(host[i * nk + j] * buf[i][1] for i in range(len(buf))) / tot) for j, k in enumerate(keys)}
        rec = dict(mode='train', epoch=runner.epoch + 1, iter=runner.inner_iter + 1, lr=runner.current_lr()[0], **avg)
        if runner.logger:
            runner.logger.info('Epoch [%d][%d/%d]\tlr: %.3e, %s', rec['epoch'], rec['iter'], len(runner.data_loader),
                               rec['lr'], ', '.join(f'{k}: {v:.4f}' for k, v in avg.items()))
        if runner.work_dir:
            os.makedirs(runner.work_dir, exist_ok=True)
            with open(os.path.join(runner.work_dir, f"{getattr(runner, 'timestamp', None) or 'train'}.log.json"), 'a') as f:
                f.write(json.dumps(rec) + '\n')
        runner.log_buffer = []


@HOOKS.register_module()
class NumClassCheckHook(Hook):
    """mmdet/datasets/utils.py:115-160: the head's num_classes must equal len(dataset.CLASSES)."""

    def before_train_epoch(self, runner):
        loader = getattr(runner, 'data_loader', None)
        ds = getattr(loader, 'dataset', loader)
        classes = getattr(ds, 'CLASSES', None)
        if classes is None:
            if runner.logger:
                runner.logger.warning('Please set `CLASSES` in the %s and check if it is consistent with the `num_classes` '
                                      'of head', ds.__class__.__name__)
            return
        assert type(classes) is not str, f'`CLASSES` in {ds.__class__.__name__} should be a tuple of str.'
        head = runner._det(runner.model).bbox_head
        assert head.num_classes == len(classes), (
            f'The `num_classes` ({head.num_classes}) in {head.__class__.__name__} of '
            f'{runner._det(runner.model).__class__.__name__} does not matches the length of `CLASSES` {len(classes)}) in '
            f'{ds.__class__.__name__}')


@HOOKS.register_module()
class DistSamplerSeedHook_semi(Hook):
    def before_train_epoch(self, runner):
        loader = getattr(runner, 'data_loader', None)
        if hasattr(getattr(loader, 'sampler', None), 'set_epoch'):
            loader.sampler.set_epoch(runner.epoch)
        elif hasattr(loader, 'set_epoch'):
            loader.set_epoch(runner.epoch)


# ------------------------------------------------------------------------------------------------
# pseudo labels
# ------------------------------------------------------------------------------------------------
from .pseudo import (PseudoLabelBank, adaptive_thresholds, fuse_host, parse_det_results,  # noqa: E402,F401
                     split_pseudo_labels)


@HOOKS.register_module()
class UnlabelPredHook(Hook):
    """On-GPU, self-scheduling pseudo-label refresh (unlabel_pred_hook.py:370-562).

    Constructed as the reference does it (apis/train.py:193-196): `UnlabelPredHook(cfg.data.unlabel_pred, cfg, 'Det',
    interval_mode=..., interval=...)`; the dict's keys may also be given as keyword arguments.  Keys read:
    infer_score_thre, first_score_thre, use_ema, start_point, preload, fuse_history / first_fuse, eval_config['iou'],
    category_info_path (dict or file with id2cat / cat2id), ada_thres_weight_settings, anno_root_path (only with
    export=True: the reference's JSON files are then written there as well).  `eval_flip` is accepted and has no effect, as in
    the reference: inference_model (:210-236,243) runs the three mirrored copies through the model and returns the
    unflipped image's result only.  `async_sweep` (not a reference key): the per-iteration sweep on its own stream beside the
    student's next step; default = whenever the loader looks at least one batch ahead (same labels, same weights:
    tests/test_runner_gpu.py::test_async_sweep_gives_the_same_labels).

    Schedule (mirrors :446-469): in iteration mode the hook wakes up once `runner.iter + 1 >= start_point *
    iters_per_epoch + 1` and `(runner.iter + 1 - start_point) % interval == 0`; the first time it sweeps EVERY
    unlabeled image (sharded over the ranks, :281), afterwards it refreshes the unlabeled image(s) this rank will
    consume `lookahead` iterations after the next one - the loader says which (`loader.unlabeled.upcoming`), where the
    reference guesses it from the sampler order and the DataLoader prefetch depth (`preload`).  Epoch mode sweeps
    everything every `interval` epochs from `start_point` on.  At every epoch end the class thresholds / weights are
    recomputed (adathres, :295-367,448).

    One refresh = teacher forward on the image's test view -> dsl_fcos_detect (top-k, decode, class-aware NMS, top 100)
    -> dsl_pseudo_label_fuse (score >= infer_score_thre, int() truncation, 6-decimal scores, second per-class NMS at
    eval_config['iou'][0] with score_threshold 0.1, :20-57,150-166; with fuse_history=True the image's stored labels join
    the candidates, :131-141, first_fuse=False keeps them out of the first sweep) -> PseudoLabelBank; nothing leaves the GPU until the
    loader asks the bank for that image's annotations."""

    def __init__(self, kwargs=None, config=None, task_type='Det', interval_mode=None, interval=None, bank=None,
                 source=None, export=False, **kw):
        k = dict(kwargs or {})
        k.update(kw)
        assert task_type == 'Det'
        self.cfg, self.config = k, config
        self.infer_score_thre = float(k.get('infer_score_thre', 0.1))
        self.first_score_thre = k.get('first_score_thre', None)
        if self.first_score_thre is None and config is not None:
            self.first_score_thre = config.get('infer_score_thre', 0.1) if hasattr(config, 'get') else 0.1
        self.use_ema = bool(k.get('use_ema', True))
        self.start_point = int(k.get('start_point', 0))
        self.fuse = bool(k.get('fuse_history', False))
        self.first_ignore = not k.get('first_fuse', True)
        self.iou = float((k.get('eval_config') or {}).get('iou', [0.6])[0])
        ec = k.get('eval_checkpoint_config') or {}
        self.interval_mode = interval_mode or ec.get('mode', 'epoch')
        self.interval = int(interval if interval is not None else ec.get('interval', 1))
        self.preload_num = int(k.get('preload', 10))
        cat = k.get('category_info_path')
        if isinstance(cat, str):
            cat = json.load(open(cat))
        self.id2cat = dict(cat['id2cat']) if cat else None
        # save_results2file loops `for i in range(0, len(id2cat) - 1)` (:152): every class id but the last entry of the
        # category file (COCO files list the 80 classes + background)
        self.num_classes = (len(self.id2cat) - 1) if self.id2cat else int(k.get('num_classes', 80))
        names = [self.id2cat[str(i)] for i in range(self.num_classes)] if self.id2cat else k.get('class_names')
        ada = None
        if config is not None:
            data = config.get('data', {}) if hasattr(config, 'get') else {}
            ada = (data.get('unlabel_train') or {}).get('thres') if data else None
        # adathres runs when the unlabeled dataset's `thres` is a (file) name (:427-435)
        self.adathres_compute = bool(isinstance(ada, str) or k.get('adathres', False) or (config is None and (bank is None or bank.mode == 'ada')))
        self.bank = bank if bank is not None else PseudoLabelBank(
            num_classes=self.num_classes, class_names=names, thres=(ada if ada is not None else 'adathres.json'),
            ada_settings=k.get('ada_thres_weight_settings'))
        self.source = source
        self.export_dir = k.get('anno_root_path') if export else k.get('export_dir')
        self.iter_fuse_flag = False
        self.n_refreshed = 0
        # not a key of the reference (see refresh()): True / False, or None = asynchronous whenever the labels being refreshed are
        # not needed by the very next batch (the loader looks >= 1 batch ahead, as the reference's `preload` does)
        self.async_sweep = k.get('async_sweep', None)
        self._lookahead = 0
        self._sweep_stream = None
        self._buf = {}

    # -- schedule (:439-469) ------------------------------------------------------------------------------
    def every_n_epochs_with_startpoint(self, runner, n, start):
        if runner.epoch + 1 < start:
            return False
        return (runner.epoch + 1 - start) % n == 0 if n > 0 else False

    def every_n_iters_with_startpoint(self, runner, n, start):
        if runner.iter + 1 < start:
            return False
        return (runner.iter + 1 - start) % n == 0 if n > 0 else False

    def _source(self, runner):
        src = self.source or getattr(getattr(runner, 'data_loader', None), 'unlabeled', None)
        if src is None:
            raise RuntimeError('UnlabelPredHook: the data loader exposes no `.unlabeled` source (dsl_amd/data.py)')
        return src

    def after_train_epoch(self, runner):
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            self.bank_all_gather()            # the ranks' per-iteration refreshes of the epoch (a shared directory in the reference)
        if self.adathres_compute:
            self.update_thresholds()
        if self.interval_mode == 'epoch' and runner.epoch + 1 >= self.start_point:
            if self.every_n_epochs_with_startpoint(runner, self.interval, self.start_point):
                self.refresh_all(runner)

    def after_train_iter(self, runner):
        if self.interval_mode != 'iteration' or runner.iter + 1 < self.start_point * runner.iter_tol_epoch + 1:
            return
        if not self.every_n_iters_with_startpoint(runner, self.interval, self.start_point):
            return
        if not self.iter_fuse_flag:          # the first fuse is the same as the epoch manner (:461-463)
            self.refresh_all(runner)
            self.iter_fuse_flag = True
            return
        src = self._source(runner)
        depth = getattr(src, 'prefetch_depth', None)
        self._lookahead = self.preload_num if depth is None else depth
        names = src.upcoming(self._lookahead)
        if names:
            self.refresh_names(runner, names)

    # -- refresh ---------------------------------------------------------------------------------------
    def _thr(self):
        if self.first_score_thre is not None:     # the first sweep may use its own threshold (:487-492)
            t, self.first_score_thre = float(self.first_score_thre), None
            return t
        return self.infer_score_thre

    def refresh_all(self, runner):
        import torch.distributed as dist
        src = self._source(runner)
        rank, world = (dist.get_rank(), dist.get_world_size()) if dist.is_available() and dist.is_initialized() else (0, 1)
        thr = self._thr()
        self.refresh_names(runner, src.names[rank::world], thr=thr)
        if world > 1:
            self.bank_all_gather()
        if self.fuse and self.first_ignore:        # after the first fuse the initial labels take part (:508-510)
            self.first_ignore = False

    def refresh_names(self, runner, names, thr=None):
        src = self._source(runner)
        thr = self._thr() if thr is None else thr
        for name in names:
            img, meta = src.test_view(name)
            self.refresh(runner, img, [meta], [name], thr=thr)

    def refresh(self, runner, imgs, img_metas, names, thr=None):
        """Teacher sweep of `imgs` ([n,3,H,W], already normalised / padded) -> fused pseudo labels into the bank."""
        from .sweep import detect_device
        det = runner._det(runner.model)
        teacher = runner._det(runner.ema_model) if (self.use_ema and runner.ema_flag and runner.ema_model is not None) else det
        thr = self.infer_score_thre if thr is None else thr
        asyn = self.async_sweep if self.async_sweep is not None else self._lookahead >= 1
        if asyn and self.iter_fuse_flag and teacher is not det:
            # (teacher is det - no EMA teacher yet, or use_ema=False - sweeps synchronously: the student's next SGD step and
            # weight re-pack would rewrite the bf16 weights / BatchNorm folds under the side stream's reads)
            # the sweep of an image the loader hands out `prefetch_depth` batches from now runs on its own stream beside the
            # student's next step (the reference refreshes `preload` iterations ahead of the loader for the same reason); it
            # starts behind everything queued so far (the teacher's EMA update included), the next EMA update waits for it
            if self._sweep_stream is None:
                from .detectors import role_stream
                self._sweep_stream = role_stream('sweep')
            ss = self._sweep_stream
            ss.wait_stream(torch.cuda.current_stream())
            imgs.record_stream(ss)
            with torch.cuda.stream(ss):
                out = self._sweep(runner, det, teacher, imgs, img_metas, names, thr, single_stream=True)
                runner._sweep_event = torch.cuda.Event()
                runner._sweep_event.record()
            return out
        return self._sweep(runner, det, teacher, imgs, img_metas, names, thr)

    def _sweep(self, runner, det, teacher, imgs, img_metas, names, thr, single_stream=False):
        from .sweep import detect_device
        dets, labels, count = detect_device(det, imgs, img_metas, rescale=True, store=teacher.store, single_stream=single_stream)
        n, maxk = dets.shape[0], dets.shape[1]
        # fresh output buffers per call: the bank reads them lazily (a ring would need the events below to bound reuse)
        ob = torch.empty(n, maxk, 4, device=dets.device)
        osc = torch.empty(n, maxk, device=dets.device)
        ol = torch.empty(n, maxk, dtype=torch.int64, device=dets.device)
        oc = torch.empty(n, dtype=torch.int32, device=dets.device)
        if self.fuse and not self.first_ignore:
            ob, osc, ol = self._fuse_with_history(dets, labels, count, names, thr, oc)
        else:
            L.check(L.lib.dsl_pseudo_label_fuse(L.ptr(dets), L.ptr(labels), L.ptr(count), n, maxk, self.num_classes, float(thr),
                                                float(self.iou), 0.1, L.ptr(ob), L.ptr(osc), L.ptr(ol), L.ptr(oc), L.stream_ptr()),
                    'dsl_pseudo_label_fuse')
        ev = torch.cuda.Event()
        ev.record()
        for i, name in enumerate(names):
            self.bank.put_device(name, ob, osc, ol, oc, i, ev, stamp=runner.iter + 1)
            if self.export_dir:
                self.bank.export_json(name, self.export_dir)
        self.n_refreshed += len(names)
        return self.bank

    def _fuse_with_history(self, dets, labels, count, names, thr, oc):
        """fuse_history=True (:131-141): every image's stored labels join its new detections before the per-class NMS.  The old
        record is read through the bank (its producing sweep is at least one refresh interval old) and goes up as one small
        table per call; the label lists grow from round to round, so the output rows are sized old + new."""
        n, maxk, dev = dets.shape[0], dets.shape[1], dets.device
        olds = [self.bank[nm] if nm in self.bank else None for nm in names]
        max_old = max([len(o['scores']) for o in olds if o is not None] + [0])
        cap = 1024 - maxk          # the fuse kernel holds 1024 candidates per image (detect.hip FUSE_MAX)
        if max_old > cap:
            # the reference has no such limit and the lists grow from round to round: keep an image's `cap` best-scoring stored labels
            # (the per-class NMS favours high scores anyway) and say so once, instead of aborting a training run
            trimmed = []
            dropped = sum(max(len(o['scores']) - cap, 0) for o in olds if o is not None)
            self.fuse_dropped_total = getattr(self, 'fuse_dropped_total', 0) + dropped       # visible divergence from the reference
            self.fuse_dropped_sweeps = getattr(self, 'fuse_dropped_sweeps', 0) + 1
            for o in olds:
                if o is not None and len(o['scores']) > cap:
                    keep = np.sort(np.argsort(-np.asarray(o['scores'], np.float32), kind='stable')[:cap])
                    o = dict(o, rects=[np.asarray(o['rects'], np.float32).reshape(-1, 4)[j].tolist() for j in keep],
                             scores=[o['scores'][j] for j in keep], tags=[o['tags'][j] for j in keep])
                trimmed.append(o)
            olds = trimmed
            if self.fuse_dropped_sweeps == 1 or self.fuse_dropped_sweeps % 100 == 0:       # the first time, then every 100th sweep that trims
                import logging
                logging.getLogger('dsl_amd').warning('fuse_history: %d stored labels + %d detections exceed the fuse step\'s 1024 candidates; keeping '
                                                      'the %d best-scoring stored labels per image (%d labels dropped in this sweep, %d in %d sweeps '
                                                      'so far - the reference keeps them all)', max_old, maxk, cap, dropped, self.fuse_dropped_total,
                                                      self.fuse_dropped_sweeps)
            max_old = cap
        mo = max(max_old, 1)
        hb, hs = torch.zeros(n, mo, 4), torch.zeros(n, mo)
        hl, hc = torch.zeros(n, mo, dtype=torch.int64), torch.zeros(n, dtype=torch.int32)
        for i, o in enumerate(olds):
            if o is not None and len(o['scores']):
                k = len(o['scores'])
                hb[i, :k] = torch.from_numpy(np.asarray(o['rects'], np.float32).reshape(-1, 4))
                hs[i, :k] = torch.from_numpy(np.asarray(o['scores'], np.float32))
                hl[i, :k] = torch.from_numpy(np.asarray(o['tags'], np.int64))
                hc[i] = k
        db, ds, dl, dc = (t.to(dev, non_blocking=True) for t in (hb, hs, hl, hc))
        mout = max_old + maxk
        ob = torch.empty(n, mout, 4, device=dev)
        osc = torch.empty(n, mout, device=dev)
        ol = torch.empty(n, mout, dtype=torch.int64, device=dev)
        L.check(L.lib.dsl_pseudo_label_fuse_history(L.ptr(dets), L.ptr(labels), L.ptr(count), n, maxk, L.ptr(db), L.ptr(ds), L.ptr(dl),
                                                    L.ptr(dc), max_old, self.num_classes, float(thr), float(self.iou), 0.1, L.ptr(ob),
                                                    L.ptr(osc), L.ptr(ol), L.ptr(oc), mout, L.stream_ptr()),
                'dsl_pseudo_label_fuse_history')
        return ob, osc, ol

    def bank_all_gather(self):
        """Every rank ends up with every rank's new records (the reference's ranks share them through the file system)."""
        import torch.distributed as dist
        mine = {n: self.bank[n] for n in self.bank.names()}
        out = [None] * dist.get_world_size()
        dist.all_gather_object(out, mine)
        for d in out:
            self.bank.merge(d)

    def update_thresholds(self):
        return self.bank.update_thresholds()

    thres = property(lambda self: self.bank.thres)
    class_weights = property(lambda self: self.bank.class_weights)

    def targets_for(self, name, img_wh=None):
        return self.bank.ann_info(name, img_wh=img_wh)
