"""Evaluation / interop side of the detector:

  results2json / format_results   mmdet/datasets/coco.py:179-360 (COCO detection-result json: xywh boxes, category ids)
  single_gpu_test / multi_gpu_test mmdet/apis/test.py (the loops EvalHook drives); detections come from the HIP sweep
  EvalHook                         mmdet/core/evaluation/eval_hooks.py:9-66: evaluates the EMA teacher once it exists
                                   (DistEvalHook :24-66 `if runner.ema_flag: model = runner.ema_model`), every `interval` epochs
  coco_bbox_eval                   the bbox protocol of pycocotools' COCOeval (10 IoU thresholds .50:.05:.95, 101 recall points,
                                   maxDets 100, crowd = ignore, area ranges) restated in numpy - pycocotools is not in this
                                   image, so this evaluator is checked on constructed cases only: PARITY UNPINNED against
                                   pycocotools; the json files are the interop path to the real tool.

The dataset side is a plain object (or dict) with `img_ids`, `cat_ids` and, for the built-in evaluator, `annotations`:
one list per image of dict(bbox=[x, y, w, h], category_id, iscrowd=0, area=optional).
"""
import json
import os
import tempfile
from collections import OrderedDict

import numpy as np
import torch


def xyxy2xywh(bbox):
    """coco.py:179-196."""
    b = np.asarray(bbox).tolist()
    return [b[0], b[1], b[2] - b[0], b[3] - b[1]]


def det2json(results, img_ids, cat_ids):
    """coco.py:213-229 (_det2json): results[idx][label] = ndarray (k, 5) xyxy + score."""
    out = []
    for idx, img_id in enumerate(img_ids):
        for label, bboxes in enumerate(results[idx]):
            for i in range(bboxes.shape[0]):
                out.append(dict(image_id=img_id, bbox=xyxy2xywh(bboxes[i]), score=float(bboxes[i][4]), category_id=cat_ids[label]))
    return out


def results2json(results, img_ids, cat_ids, outfile_prefix):
    """coco.py:269-304 for detection results: writes <prefix>.bbox.json, returns {'bbox': path, 'proposal': path}."""
    if not isinstance(results[0], list):
        raise TypeError('invalid type of results')
    files = dict(bbox=f'{outfile_prefix}.bbox.json', proposal=f'{outfile_prefix}.bbox.json')
    os.makedirs(os.path.dirname(os.path.abspath(files['bbox'])), exist_ok=True)
    with open(files['bbox'], 'w') as f:
        json.dump(det2json(results, img_ids, cat_ids), f)
    return files


def format_results(results, img_ids, cat_ids, jsonfile_prefix=None):
    """coco.py:334-360."""
    assert isinstance(results, list), 'results must be a list'
    assert len(results) == len(img_ids), f'The length of results is not equal to the dataset len: {len(results)} != {len(img_ids)}'
    tmp_dir = None
    if jsonfile_prefix is None:
        tmp_dir = tempfile.TemporaryDirectory()
        jsonfile_prefix = os.path.join(tmp_dir.name, 'results')
    return results2json(results, img_ids, cat_ids, jsonfile_prefix), tmp_dir


# ----------------------------------------------------------------------------------------------------------------------
def _iou_xywh(d, g, crowd):
    """[D, G] IoU of xywh boxes; against a crowd box the union is the detection's area (COCO maskUtils.iou)."""
    d, g = np.asarray(d, np.float64).reshape(-1, 4), np.asarray(g, np.float64).reshape(-1, 4)
    if not len(d) or not len(g):
        return np.zeros((len(d), len(g)))
    dx2, dy2, gx2, gy2 = d[:, 0] + d[:, 2], d[:, 1] + d[:, 3], g[:, 0] + g[:, 2], g[:, 1] + g[:, 3]
    iw = np.clip(np.minimum(dx2[:, None], gx2[None]) - np.maximum(d[:, None, 0], g[None, :, 0]), 0, None)
    ih = np.clip(np.minimum(dy2[:, None], gy2[None]) - np.maximum(d[:, None, 1], g[None, :, 1]), 0, None)
    inter = iw * ih
    da, ga = (d[:, 2] * d[:, 3])[:, None], (g[:, 2] * g[:, 3])[None]
    union = np.where(np.asarray(crowd, bool)[None], da, da + ga - inter)
    return inter / np.maximum(union, 1e-12)


def coco_bbox_eval(dets, img_ids, cat_ids, annotations, iou_thrs=None, max_dets=100):
    """dets: the list det2json produces; annotations[idx]: list of dict(bbox xywh, category_id, iscrowd, [area]).
    Returns OrderedDict(mAP, mAP_50, mAP_75, mAP_s, mAP_m, mAP_l) (-1 where no ground truth falls in the range)."""
    iou_thrs = np.linspace(.5, .95, 10) if iou_thrs is None else np.asarray(iou_thrs, np.float64)
    rec_thrs = np.linspace(0, 1, 101)
    ranges = OrderedDict(all=(0, 1e10), small=(0, 32 ** 2), medium=(32 ** 2, 96 ** 2), large=(96 ** 2, 1e10))
    by = {}
    for d in dets:
        by.setdefault((d['image_id'], d['category_id']), []).append(d)
    prec = {r: -np.ones((len(iou_thrs), len(rec_thrs), len(cat_ids))) for r in ranges}
    for ci, cat in enumerate(cat_ids):
        for rname, (lo, hi) in ranges.items():
            scores, matched, ignored, npos = [], [], [], 0
            for idx, img_id in enumerate(img_ids):
                gts = [a for a in annotations[idx] if a['category_id'] == cat]
                garea = np.array([a.get('area', a['bbox'][2] * a['bbox'][3]) for a in gts], np.float64)
                crowd = np.array([bool(a.get('iscrowd', 0)) for a in gts], bool)
                gig = crowd | (garea < lo) | (garea > hi)
                order = np.argsort(gig, kind='mergesort')            # non-ignored ground truth first
                gts, crowd, gig = [gts[i] for i in order], crowd[order], gig[order]
                dd = sorted(by.get((img_id, cat), []), key=lambda d: -d['score'])[:max_dets]
                ious = _iou_xywh([d['bbox'] for d in dd], [a['bbox'] for a in gts], crowd)
                npos += int((~gig).sum())
                dm = -np.ones((len(iou_thrs), len(dd)), int)
                dig = np.zeros((len(iou_thrs), len(dd)), bool)
                for ti, t in enumerate(iou_thrs):
                    gm = -np.ones(len(gts), int)
                    for di in range(len(dd)):
                        best, m = min(t, 1 - 1e-10), -1
                        for gi in range(len(gts)):
                            if gm[gi] >= 0 and not crowd[gi]:
                                continue
                            if m > -1 and not gig[m] and gig[gi]:
                                break                                  # already matched to a regular gt: do not trade it for an ignored one
                            if ious[di, gi] < best:
                                continue
                            best, m = ious[di, gi], gi
                        if m >= 0:
                            dm[ti, di], gm[m], dig[ti, di] = m, di, gig[m]
                darea = np.array([d['bbox'][2] * d['bbox'][3] for d in dd], np.float64)
                dig |= (dm < 0) & ((darea < lo) | (darea > hi))[None]
                scores += [d['score'] for d in dd]
                matched.append(dm >= 0)
                ignored.append(dig)
            if npos == 0:
                continue
            sc = np.array(scores)
            order = np.argsort(-sc, kind='mergesort')
            mt = np.concatenate(matched, 1)[:, order] if matched else np.zeros((len(iou_thrs), 0), bool)
            ig = np.concatenate(ignored, 1)[:, order] if ignored else np.zeros((len(iou_thrs), 0), bool)
            for ti in range(len(iou_thrs)):
                tp = np.cumsum(mt[ti] & ~ig[ti])
                fp = np.cumsum(~mt[ti] & ~ig[ti])
                rc = tp / npos
                pr = tp / np.maximum(tp + fp, np.spacing(1))
                for i in range(len(pr) - 1, 0, -1):                    # precision envelope
                    pr[i - 1] = max(pr[i - 1], pr[i])
                q = np.zeros(len(rec_thrs))
                inds = np.searchsorted(rc, rec_thrs, side='left')
                ok = inds < len(pr)
                q[ok] = pr[inds[ok]]
                prec[rname][ti, :, ci] = q

    def mean(p, ti=None):
        p = p if ti is None else p[ti:ti + 1]
        v = p[p > -1]
        return float(v.mean()) if v.size else -1.0
    i50 = int(np.argmin(np.abs(iou_thrs - .5)))
    i75 = int(np.argmin(np.abs(iou_thrs - .75)))
    return OrderedDict(mAP=mean(prec['all']), mAP_50=mean(prec['all'], i50), mAP_75=mean(prec['all'], i75), mAP_s=mean(prec['small']),
                       mAP_m=mean(prec['medium']), mAP_l=mean(prec['large']))


# ----------------------------------------------------------------------------------------------------------------------
def _unwrap(x):
    return x[0] if isinstance(x, (list, tuple)) and len(x) == 1 and isinstance(x[0], (list, tuple, torch.Tensor)) else x


@torch.no_grad()
def single_gpu_test(model, data_loader, store=None):
    """mmdet/apis/test.py single_gpu_test without the visualisation: per image the list of per-class (k, 5) arrays."""
    det = model.module if hasattr(model, 'module') else model
    results = []
    for data in data_loader:
        img, metas = _unwrap(data['img']), _unwrap(data['img_metas'])
        from .sweep import simple_test
        results.extend(simple_test(det, img.to(det.store.device), metas, rescale=True, store=store))
    return results


@torch.no_grad()
def multi_gpu_test(model, data_loader, store=None):
    """Every rank tests its loader's shard (the loader yields rank-local batches, index = rank + k * world as a
    DistributedSampler(shuffle=False) does); results are gathered in dataset order on every rank."""
    import torch.distributed as dist
    part = single_gpu_test(model, data_loader, store)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return part
    parts = [None] * dist.get_world_size()
    dist.all_gather_object(parts, part)
    out = []
    for i in range(max(len(p) for p in parts)):
        out += [p[i] for p in parts if i < len(p)]
    n = getattr(getattr(data_loader, 'dataset', data_loader), 'num_images', len(out))
    return out[:n]


class EvalHook:
    """eval_hooks.py:9-66 on top of mmcv's EvalHook: every `interval` epochs (from `start`), test `dataloader` with the EMA
    teacher if the runner has one (`runner.ema_flag`), else the student; evaluate with the dataset's own `evaluate()` or, for
    datasets that carry `annotations`, the built-in COCO bbox protocol; log the metrics; optionally keep the best checkpoint."""
    priority = 75

    def __init__(self, dataloader, start=None, interval=1, by_epoch=True, save_best=None, metric='bbox', jsonfile_prefix=None,
                 **eval_kwargs):
        self.dataloader, self.start, self.interval, self.by_epoch = dataloader, start, interval, by_epoch
        self.save_best = 'mAP' if save_best == 'auto' else save_best
        self.metric, self.jsonfile_prefix, self.eval_kwargs = metric, jsonfile_prefix, eval_kwargs
        self.best, self.history = None, []

    def __getattr__(self, name):
        if name.startswith(('before_', 'after_')):
            return lambda runner: None
        raise AttributeError(name)

    def _should_evaluate(self, runner):
        cur = runner.epoch + 1
        if self.start is not None and cur < self.start:
            return False
        return (cur - (self.start or 0)) % self.interval == 0 if self.start is not None else cur % self.interval == 0

    def after_train_epoch(self, runner):
        if self.by_epoch and self._should_evaluate(runner):
            self._do_evaluate(runner)

    def _do_evaluate(self, runner):
        det = runner._det(runner.model)
        store = runner._det(runner.ema_model).store if (runner.ema_flag and runner.ema_model is not None) else None
        if store is not None and runner.logger:
            runner.logger.info('Using ema model for eval')
        results = multi_gpu_test(det, self.dataloader, store)
        # DistEvalHook (mmdet/core/evaluation/eval_hooks.py): every rank takes part in the gather above, but only rank 0 formats,
        # evaluates, logs and saves - the other ranks would race on the same json / checkpoint paths - and the metrics are
        # broadcast so every rank's history / best agree
        import torch.distributed as dist
        multi = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        rank = dist.get_rank() if multi else 0
        metrics = None
        if rank == 0:
            ds = getattr(self.dataloader, 'dataset', self.dataloader)
            if hasattr(ds, 'evaluate'):
                metrics = ds.evaluate(results, metric=self.metric, **self.eval_kwargs)
            else:
                prefix = self.jsonfile_prefix or (os.path.join(runner.work_dir, f'eval_epoch_{runner.epoch + 1}') if runner.work_dir else None)
                files, tmp = format_results(results, ds.img_ids, ds.cat_ids, prefix)
                metrics = coco_bbox_eval(json.load(open(files['bbox'])), ds.img_ids, ds.cat_ids, ds.annotations)
                metrics = OrderedDict((f'bbox_{k}', v) for k, v in metrics.items())
                if tmp is not None:
                    tmp.cleanup()
        if multi:
            box = [metrics]
            dist.broadcast_object_list(box, src=0)
            metrics = box[0]
        self.history.append((runner.epoch + 1, dict(metrics)))
        if runner.logger and rank == 0:
            runner.logger.info('Epoch(val) [%d]\t%s', runner.epoch + 1, ', '.join(f'{k}: {v:.4f}' for k, v in metrics.items()))
        key = next((k for k in metrics if self.save_best and k.endswith(self.save_best)), None)
        if key is not None and (self.best is None or metrics[key] > self.best) and runner.work_dir:
            self.best = metrics[key]
            if rank == 0:
                runner.save_checkpoint(runner.work_dir, filename_tmpl='best_' + key + '_epoch_{}.pth', create_symlink=False)
        if multi:
            dist.barrier()
        return metrics


def load_checkpoint(model, filename, map_location='cpu', strict=False, revise_keys=((r'^module\.', ''),)):
    """mmcv.runner.load_checkpoint for upstream .pth files: optional 'state_dict' wrapper, key rewriting
    (default: strip DDP's 'module.'), non-strict by default.  Returns the checkpoint dict."""
    import re
    ck = torch.load(filename, map_location=map_location)
    sd = ck.get('state_dict', ck) if isinstance(ck, dict) else ck
    for pat, rep in revise_keys:
        sd = OrderedDict((re.sub(pat, rep, k), v) for k, v in sd.items())
    det = model.module if hasattr(model, 'module') else model
    det.load_state_dict(sd, strict=strict)
    return ck
