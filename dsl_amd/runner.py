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


def append_half_scale(img, gt_bboxes, gt_labels, gt_bboxes_ignore, img_metas):
    """semi_epoch_based_runner.py:186-204, on device: bilinear half-size copy of the LAST image pasted at the
    top-left of a zero canvas; boxes and ignore boxes halved; meta shapes halved (int)."""
    B, _, H, W = img.shape
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

    def register_hook(self, hook, priority=50):
        hook.priority = priority
        i = len(self._hooks)
        while i > 0 and self._hooks[i - 1].priority > priority:
            i -= 1
        self._hooks.insert(i, hook)

    def call_hook(self, name):
        for h in self._hooks:
            getattr(h, name)(self)

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
                img, gb, gl, gi, metas = append_half_scale(d['img'], d['gt_bboxes'], d['gt_labels'],
                                                           d.get('gt_bboxes_ignore'), d['img_metas'])
                data_batch = dict(d, img=img, gt_bboxes=gb, gt_labels=gl, img_metas=metas)
                if gi is not None:
                    data_batch['gt_bboxes_ignore'] = gi
            self.call_hook('before_train_iter')
            self.run_iter(data_batch, train_mode=True, **kw)
            self.call_hook('after_train_iter')
            self._iter += 1
        self.call_hook('after_train_epoch')
        self._epoch += 1

    def run(self, data_loaders, workflow=(('train', 1),), max_epochs=None, **kw):
        if max_epochs is not None:
            self._max_epochs = max_epochs
        self.call_hook('before_run')
        while self._epoch < self._max_epochs:
            for (mode, epochs), loader in zip(workflow, data_loaders):
                assert mode == 'train'
                for _ in range(epochs):
                    if self._epoch >= self._max_epochs:
                        break
                    self.train(loader, **kw)
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
        L.check(L.lib.dsl_ema_lerp(L.ptr(t.train), L.ptr(s.train), s.n_train, float(keep_rate), L.stream_ptr()), 'dsl_ema_lerp')
        if t.train16 is None or t.dirty:
            t.refresh()
        else:
            L.check(L.lib.dsl_cast_bf16(L.ptr(t.train), L.ptr(t.train16), t.n_train, L.stream_ptr()), 'dsl_cast_bf16')
        self.ema_flag = True

    def save_checkpoint(self, out_dir, filename_tmpl='epoch_{}.pth', meta=None):
        """:411-458: student to <file>, teacher to <file>_ema."""
        os.makedirs(out_dir, exist_ok=True)
        fn = os.path.join(out_dir, filename_tmpl.format(self.epoch + 1))
        meta = dict(meta or {}, epoch=self.epoch + 1, iter=self.iter)
        sd = {k: v.detach().cpu().clone() for k, v in self._det(self.model).state_dict().items()}
        torch.save(dict(meta=meta, state_dict=sd), fn)
        if self.ema_model is not None:
            sd = {k: v.detach().cpu().clone() for k, v in self._det(self.ema_model).state_dict().items()}
            torch.save(dict(meta=meta, state_dict=sd), fn + '_ema')
        return fn

    def load_checkpoint(self, filename, map_location='cpu', strict=False):
        """:350-366: the same file goes into both student and teacher."""
        ck = torch.load(filename, map_location=map_location)
        sd = ck.get('state_dict', ck)
        self._det(self.model).load_state_dict(sd, strict=strict)
        if self.ema_model is not None:
            self._det(self.ema_model).load_state_dict(sd, strict=strict)
        return ck


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
class NumClassCheckHook(Hook):
    pass


@HOOKS.register_module()
class DistSamplerSeedHook_semi(Hook):
    def before_train_epoch(self, runner):
        loader = getattr(runner, 'data_loader', None)
        if hasattr(getattr(loader, 'sampler', None), 'set_epoch'):
            loader.sampler.set_epoch(runner.epoch)


# ------------------------------------------------------------------------------------------------
# pseudo labels
# ------------------------------------------------------------------------------------------------
def adaptive_thresholds(scores_by_class, prev_thres=None, ranges=(0.3, 0.35), gamma1=0.05, gamma2=0.6, base=0.3):
    """unlabel_pred_hook.py:295-367: from the per-class lists of pseudo-label scores, count / accumulate those
    above the previous class threshold (0.3 the first time), then
      thres_c  = clip((cum_c / (avg / n_cls)) ** gamma1 * base, ranges)
      weight_c = (avg / n_cls / cum_c) ** gamma2."""
    dis, cum = {}, {}
    for c, sc in scores_by_class.items():
        sc = np.asarray(sc, dtype=np.float64)
        thr = 0.3 if prev_thres is None else prev_thres.get(c)
        sel = sc if thr is None else sc[sc >= thr]
        if len(sel):
            dis[c], cum[c] = len(sel), float(sel.sum())
    if not dis:
        return {}, {}
    avg = sum(dis.values())
    weights = {c: (avg / len(dis) / cum[c]) ** gamma2 for c in dis}
    thres = {c: max(min((cum[c] / (avg / len(dis))) ** gamma1 * base, ranges[1]), ranges[0]) for c in dis}
    return thres, weights


def split_pseudo_labels(boxes, labels, scores, thres_by_class, default_thres=(0.1, 0.3), img_wh=None):
    """datasets/semicoco.py:184-291: score in [default_lo, thr_c) -> gt_bboxes_ignore, otherwise -> gt box; classes
    without a threshold of their own use the dataset's default band [0.1, 0.3) (semicoco.py:56, semivoc.py:43)."""
    gt, gl, ig = [], [], []
    for b, l, s in zip(boxes, labels, scores):
        x1, y1, x2, y2 = (float(v) for v in b)
        if img_wh is not None:
            if max(0, min(x2, img_wh[0]) - max(x1, 0)) * max(0, min(y2, img_wh[1]) - max(y1, 0)) == 0:
                continue
        if x2 - x1 < 1 or y2 - y1 < 1:
            continue
        hi = thres_by_class.get(int(l), default_thres[1])
        if default_thres[0] <= s < hi:
            ig.append([x1, y1, x2, y2])
        else:
            gt.append([x1, y1, x2, y2])
            gl.append(int(l))
    f = lambda a: torch.tensor(a, dtype=torch.float32).reshape(-1, 4)
    return f(gt), torch.tensor(gl, dtype=torch.int64), f(ig)


def parse_det_results(dets, labels, score_thr):
    """unlabel_pred_hook.py:20-38 + the score sort of gen_save_json_dict (:40-57): keep detections with
    score >= score_thr, integer-truncate the coordinates (int()), round the score to 6 decimals, highest score first.
    dets [k, 5] (x1, y1, x2, y2, score), labels [k]."""
    dets, labels = np.asarray(dets), np.asarray(labels)
    keep = dets[:, 4] >= score_thr
    b, l = dets[keep], labels[keep]
    scores = np.array([round(float(v), 6) for v in b[:, 4]], dtype=np.float64)
    order = np.argsort(-scores, kind='stable')
    return dict(rects=np.trunc(b[order, :4]).astype(np.int64), tags=l[order].astype(np.int64), scores=scores[order])


@HOOKS.register_module()
class UnlabelPredHook(Hook):
    """On-GPU pseudo-label refresh.  Each call runs the (EMA) teacher on the given unlabeled images with the
    HIP sweep, keeps detections with score >= infer_score_thre with integer-truncated coordinates
    (parse_det_results :20-38) in an in-memory bank keyed by image name, and optionally exports the
    reference's per-image JSON {imageName,targetNum,rects,tags,masks,scores}."""

    def __init__(self, infer_score_thre=0.1, use_ema=True, start_point=8, export_dir=None, class_names=None,
                 eval_checkpoint_config=None, **kw):
        self.score_thr, self.use_ema, self.start_point = infer_score_thre, use_ema, start_point
        self.export_dir, self.class_names = export_dir, class_names
        self.bank = {}
        self.thres, self.class_weights = None, None
        self.cfg = kw

    def refresh(self, runner, imgs, img_metas, names):
        from .sweep import detect_device
        det = runner._det(runner.model)
        teacher = runner._det(runner.ema_model) if (self.use_ema and runner.ema_flag) else det
        dets, labels, count = detect_device(det, imgs, img_metas, rescale=True, store=teacher.store)
        dets, labels, count = dets.cpu().numpy(), labels.cpu().numpy(), count.cpu().numpy()
        for i, name in enumerate(names):
            k = int(count[i])
            self.bank[name] = parse_det_results(dets[i, :k], labels[i, :k], self.score_thr)
            if self.export_dir:
                self._export(name)
        return self.bank

    def _export(self, name):
        e = self.bank[name]
        os.makedirs(self.export_dir, exist_ok=True)
        tags = [self.class_names[t] if self.class_names else int(t) for t in e['tags']]
        with open(os.path.join(self.export_dir, os.path.basename(name) + '.json'), 'w') as f:
            json.dump(dict(imageName=name, targetNum=len(tags), rects=e['rects'].tolist(), tags=tags,
                           masks=[[] for _ in tags], scores=e['scores'].tolist()), f)

    def update_thresholds(self):
        by_c = {}
        for e in self.bank.values():
            for t, s in zip(e['tags'], e['scores']):
                by_c.setdefault(int(t), []).append(float(s))
        self.thres, self.class_weights = adaptive_thresholds(by_c, self.thres)
        return self.thres

    def targets_for(self, name, img_wh=None):
        e = self.bank[name]
        return split_pseudo_labels(e['rects'], e['tags'], e['scores'], self.thres or {}, img_wh=img_wh)

    def after_train_epoch(self, runner):
        self.update_thresholds()
