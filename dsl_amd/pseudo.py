"""Pseudo labels of the unlabeled stream: the in-memory replacement of the reference's per-image JSON annotation
files, with the reference's semantics.

  parse_det_results / fuse   mmdet/runner/hooks/unlabel_pred_hook.py:20-57, 84-171 (save_results2file, fuse_history False / True)
  adaptive_thresholds        unlabel_pred_hook.py:295-367 (adathres)
  split_pseudo_labels        mmdet/datasets/semicoco.py:184-291 (SemiCOCODataset._parse_ann_info)
  file formats               tools/generate_unlabel_annos_coco.py:36-58 ({imageName,targetNum,rects,tags,masks,scores}),
                             adathres file {cat, id, thres}

The reference writes one JSON per unlabeled image, re-reads all of them at every epoch end for the class statistics
and re-reads one per training sample; `PseudoLabelBank` keeps the same records in memory (device results are pulled to
the host lazily, so a refresh does not stall the training stream) and can export / import the reference's files.
"""
import json
import os

import numpy as np
import torch


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


def fuse_host(dets, labels, parse_thr, iou_thr, nms_thr=0.1, num_classes=80, old=None):
    """Host restatement of the label-file step (what dsl_pseudo_label_fuse does on the GPU): parse_det_results, then per
    class 0..num_classes-1 mmcv.ops.nms(iou_threshold, score_threshold=nms_thr) on the truncated boxes
    (unlabel_pred_hook.py:150-166).  old = dict(rects, tags, scores): fuse_history=True, the stored labels precede the new
    detections (:131-141).  Used by the CPU tests and as the checker of the kernel."""
    e = parse_det_results(dets, labels, parse_thr)
    rects = e['rects'].astype(np.float32)
    scores = e['scores'].astype(np.float32)
    if old is not None and len(old['scores']):
        rects = np.concatenate([np.asarray(old['rects'], np.float32).reshape(-1, 4), rects])
        scores = np.concatenate([np.asarray(old['scores'], np.float32), scores])
        e = dict(tags=np.concatenate([np.asarray(old['tags'], np.int64), e['tags']]))
    out_b, out_s, out_l = [], [], []
    for c in range(num_classes):
        idx = np.nonzero((e['tags'] == c) & (scores > np.float32(nms_thr)))[0]
        if not len(idx):
            continue
        idx = idx[np.argsort(-scores[idx], kind='stable')]
        kept = []
        for i in idx:
            a = rects[i]
            ok = True
            for j in kept:
                b = rects[j]
                w = max(np.float32(0), min(a[2], b[2]) - max(a[0], b[0]))
                h = max(np.float32(0), min(a[3], b[3]) - max(a[1], b[1]))
                inter = np.float32(w * h)
                union = np.float32((a[2] - a[0]) * (a[3] - a[1]) + (b[2] - b[0]) * (b[3] - b[1]) - inter)
                with np.errstate(divide='ignore', invalid='ignore'):
                    if np.float32(inter / union) > np.float32(iou_thr):
                        ok = False
                        break
            if ok:
                kept.append(i)
        out_b += [rects[i] for i in kept]
        out_s += [scores[i] for i in kept]
        out_l += [c] * len(kept)
    return dict(rects=np.array(out_b, np.float32).reshape(-1, 4), tags=np.array(out_l, np.int64),
                scores=np.array(out_s, np.float32))


class PseudoLabelBank:
    """name -> {rects [k,4], tags [k] (class index), scores [k]}; per-class thresholds / weights (adathres)."""

    def __init__(self, num_classes=80, class_names=None, default_thres=(0.1, 0.3), thres=None, ada_settings=None):
        self.num_classes = num_classes
        self.class_names = list(class_names) if class_names is not None else None
        self.default_thres = tuple(default_thres)
        # `thres` of the dataset config (semicoco.py:55): None = every stored box is a gt box, a [lo, hi] pair = fixed band,
        # a str = adaptive thresholds (the name of the reference's threshold file)
        self.mode = None if thres is None else ('ada' if isinstance(thres, str) else 'fixed')
        self.fixed_band = tuple(thres) if self.mode == 'fixed' else None
        self.ada_settings = dict(ada_settings or {})
        self.entries, self._pending = {}, {}
        self.thres, self.class_weights = None, None          # None until the first adathres computation ("file absent")

    # -- records ------------------------------------------------------------------------------------
    def put(self, name, rects, tags, scores, stamp=0):
        """stamp: training iteration that produced the record; merge() keeps the newest."""
        self._pending.pop(name, None)
        self.entries[name] = dict(rects=np.asarray(rects).reshape(-1, 4), tags=np.asarray(tags, np.int64).reshape(-1),
                                  scores=np.asarray(scores, np.float64).reshape(-1), stamp=int(stamp))

    def put_device(self, name, boxes, scores, labels, count, index, event=None, stamp=0):
        """Results still on the GPU (row `index` of the fuse outputs): pulled to the host when first read."""
        self.entries.pop(name, None)
        self._pending[name] = (boxes, scores, labels, count, index, event, stamp)

    def _materialize(self, name):
        p = self._pending.pop(name, None)
        if p is not None:
            boxes, scores, labels, count, i, ev, stamp = p
            if ev is not None:
                ev.synchronize()
            k = int(count[i])
            self.put(name, boxes[i, :k].cpu().numpy(), labels[i, :k].cpu().numpy(), scores[i, :k].cpu().numpy(), stamp)

    def merge(self, records):
        """records: name -> entry dict (another rank's bank): the newer record of a name wins, ties keep ours."""
        for n, e in records.items():
            if n not in self or e.get('stamp', 0) > self[n].get('stamp', 0):
                self.put(n, e['rects'], e['tags'], e['scores'], e.get('stamp', 0))

    def __contains__(self, name):
        return name in self.entries or name in self._pending

    def __getitem__(self, name):
        self._materialize(name)
        return self.entries[name]

    def names(self):
        return list(dict.fromkeys(list(self.entries) + list(self._pending)))

    def __len__(self):
        return len(self.names())

    # -- what the training pipeline loads (semicoco.py:184-291) -------------------------------------------
    def ann_info(self, name, img_wh=None):
        """(gt_bboxes [g,4], gt_labels [g], gt_bboxes_ignore [k,4]) of an unlabeled image from its stored pseudo labels."""
        if name not in self:
            z = torch.zeros(0, 4)
            return z, torch.zeros(0, dtype=torch.int64), z.clone()
        e = self[name]
        if self.mode is None:
            return split_pseudo_labels(e['rects'], e['tags'], e['scores'], {}, default_thres=(2.0, 2.0), img_wh=img_wh)
        if self.mode == 'fixed':
            return split_pseudo_labels(e['rects'], e['tags'], e['scores'], {}, default_thres=self.fixed_band, img_wh=img_wh)
        return split_pseudo_labels(e['rects'], e['tags'], e['scores'], self.thres or {}, default_thres=self.default_thres,
                                   img_wh=img_wh)

    # -- adathres (unlabel_pred_hook.py:295-367) ----------------------------------------------------------
    def class_scores(self):
        by_c = {}
        for n in self.names():
            e = self[n]
            for t, s in zip(e['tags'], e['scores']):
                by_c.setdefault(int(t), []).append(float(s))
        return by_c

    def update_thresholds(self):
        s = self.ada_settings
        self.thres, self.class_weights = adaptive_thresholds(
            self.class_scores(), self.thres, ranges=tuple(s.get('ranges', (0.3, 0.35))), gamma1=s.get('gamma1', 0.05),
            gamma2=s.get('gamma2', 0.6), base=s.get('base', 0.3))
        return self.thres

    # -- the reference's files ---------------------------------------------------------------------------
    def _tag(self, t):
        return self.class_names[int(t)] if self.class_names else int(t)

    def export_json(self, name, out_dir):
        e = self[name]
        os.makedirs(out_dir, exist_ok=True)
        tags = [self._tag(t) for t in e['tags']]
        path = os.path.join(out_dir, os.path.basename(name) + '.json')
        with open(path, 'w', encoding='utf-8') as f:
            json.dump(dict(imageName=name, targetNum=len(tags), rects=np.asarray(e['rects']).tolist(), tags=tags,
                           masks=[[] for _ in tags], scores=np.asarray(e['scores']).tolist()), f, indent=4, ensure_ascii=False)
        return path

    def import_json(self, path, name=None):
        d = json.load(open(path))
        ids = {n: i for i, n in enumerate(self.class_names)} if self.class_names else {}
        keep = [i for i, t in enumerate(d['tags']) if not ids or t in ids]
        tags = [ids[d['tags'][i]] if ids else int(d['tags'][i]) for i in keep]
        scores = d.get('scores') or [1.0] * len(d['tags'])
        self.put(name or d.get('imageName') or os.path.basename(path)[:-5], [d['rects'][i] for i in keep], tags,
                 [scores[i] for i in keep])

    def export_thres(self, path):
        t, w = self.thres or {}, self.class_weights or {}
        cat = {self._tag(c): w[c] for c in sorted(w, key=lambda c: str(self._tag(c)))}
        with open(path, 'w', encoding='utf-8') as f:
            json.dump(dict(cat=cat, id={int(c): w[c] for c in w}, thres={self._tag(c): t[c] for c in t}), f, indent=4,
                      ensure_ascii=False)
