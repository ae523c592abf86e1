"""Row f4: COCO result export, the built-in bbox evaluator, upstream checkpoint loading - CPU side."""
import json
import os

import numpy as np
import pytest
import torch

from dsl_amd import evaluation as E


def test_xyxy2xywh_and_det2json():
    # coco.py:179-229: [x1, y1, x2, y2] -> [x1, y1, w, h]; one entry per box, category id through cat_ids
    assert E.xyxy2xywh(np.array([10., 20., 50., 80., .9])) == [10., 20., 40., 60.]
    res = [[np.array([[10., 20., 50., 80., .9]], np.float32), np.zeros((0, 5), np.float32)],
           [np.zeros((0, 5), np.float32), np.array([[0., 0., 4., 4., .5], [1., 1., 3., 3., .25]], np.float32)]]
    js = E.det2json(res, [7, 9], [3, 17])
    assert js == [dict(image_id=7, bbox=[10., 20., 40., 60.], score=pytest.approx(.9), category_id=3),
                  dict(image_id=9, bbox=[0., 0., 4., 4.], score=.5, category_id=17),
                  dict(image_id=9, bbox=[1., 1., 2., 2.], score=.25, category_id=17)]


def test_format_results(tmp_path):
    res = [[np.array([[1., 2., 3., 4., .5]], np.float32)]]
    files, tmp = E.format_results(res, [1], [1], str(tmp_path / 'out'))
    assert tmp is None and files['bbox'].endswith('out.bbox.json') and files['proposal'] == files['bbox']
    assert json.load(open(files['bbox']))[0]['bbox'] == [1., 2., 2., 2.]
    files, tmp = E.format_results(res, [1], [1])
    assert os.path.exists(files['bbox'])
    tmp.cleanup()
    with pytest.raises(AssertionError, match='length of results'):
        E.format_results(res, [1, 2], [1])
    with pytest.raises(TypeError):
        E.results2json([np.zeros((0, 5))], [1], [1], str(tmp_path / 'x'))


def _ann(x, y, w, h, c, crowd=0):
    return dict(bbox=[x, y, w, h], category_id=c, iscrowd=crowd)


def _det(i, x, y, w, h, s, c):
    return dict(image_id=i, bbox=[x, y, w, h], score=s, category_id=c)


def test_eval_perfect_and_empty():
    anns = [[_ann(10, 10, 50, 50, 1), _ann(100, 100, 200, 200, 2)], [_ann(0, 0, 20, 20, 1)]]
    dets = [_det(1, 10, 10, 50, 50, .9, 1), _det(1, 100, 100, 200, 200, .8, 2), _det(2, 0, 0, 20, 20, .7, 1)]
    m = E.coco_bbox_eval(dets, [1, 2], [1, 2, 3], anns)
    assert m['mAP'] == pytest.approx(1.0) and m['mAP_50'] == pytest.approx(1.0) and m['mAP_75'] == pytest.approx(1.0)
    assert m['mAP_s'] == pytest.approx(1.0) and m['mAP_m'] == pytest.approx(1.0) and m['mAP_l'] == pytest.approx(1.0)
    m = E.coco_bbox_eval([], [1, 2], [1, 2, 3], anns)
    assert m['mAP'] == 0.0
    m = E.coco_bbox_eval(dets, [1, 2], [3], anns)          # a category without ground truth does not count
    assert m['mAP'] == -1.0


def test_eval_iou_thresholds():
    # one gt 100x100; detection shifted by 10 px in x: IoU = 90/110 = 0.818 -> a hit at thresholds .50 ... .80 (7 of 10)
    anns = [[_ann(0, 0, 100, 100, 1)]]
    m = E.coco_bbox_eval([_det(1, 10, 0, 100, 100, .9, 1)], [1], [1], anns)
    assert m['mAP'] == pytest.approx(0.7) and m['mAP_50'] == pytest.approx(1.0) and m['mAP_75'] == pytest.approx(1.0)
    m = E.coco_bbox_eval([_det(1, 30, 0, 100, 100, .9, 1)], [1], [1], anns)     # IoU 70/130 = .538: only the .50 threshold
    assert m['mAP'] == pytest.approx(0.1) and m['mAP_75'] == 0.0


def test_eval_ranking_and_duplicates():
    # two gts of one class; detections by score: hit, false positive, hit -> precision at recall .5 = 1, at recall 1 = 2/3
    anns = [[_ann(0, 0, 50, 50, 1), _ann(200, 200, 50, 50, 1)]]
    dets = [_det(1, 0, 0, 50, 50, .9, 1), _det(1, 400, 400, 50, 50, .8, 1), _det(1, 200, 200, 50, 50, .7, 1)]
    m = E.coco_bbox_eval(dets, [1], [1], anns)
    want = (51 * 1.0 + 50 * (2 / 3)) / 101                   # recall thresholds 0 ... .50 -> 1.0; .51 ... 1.0 -> 2/3
    assert m['mAP_50'] == pytest.approx(want)
    # a second detection of an already matched gt is a false positive
    dets = [_det(1, 0, 0, 50, 50, .9, 1), _det(1, 1, 0, 50, 50, .8, 1), _det(1, 200, 200, 50, 50, .7, 1)]
    assert E.coco_bbox_eval(dets, [1], [1], anns)['mAP_50'] == pytest.approx(want)


def test_eval_crowd_is_ignored():
    # a detection inside a crowd region is neither a hit nor a false positive; crowd gts do not count as positives
    anns = [[_ann(0, 0, 50, 50, 1), _ann(100, 100, 300, 300, 1, crowd=1)]]
    dets = [_det(1, 150, 150, 40, 40, .95, 1), _det(1, 160, 160, 40, 40, .9, 1), _det(1, 0, 0, 50, 50, .5, 1)]
    assert E.coco_bbox_eval(dets, [1], [1], anns)['mAP'] == pytest.approx(1.0)


def test_eval_max_dets():
    anns = [[_ann(0, 0, 50, 50, 1)]]
    dets = [_det(1, 300 + i, 300, 10, 10, .9 - .001 * i, 1) for i in range(100)] + [_det(1, 0, 0, 50, 50, .1, 1)]
    assert E.coco_bbox_eval(dets, [1], [1], anns)['mAP'] == 0.0           # the hit is detection 101
    assert E.coco_bbox_eval(dets, [1], [1], anns, max_dets=101)['mAP'] > 0.0


def test_load_checkpoint_strips_module_prefix(tmp_path):
    from util import fcos_model_cfg
    import dsl_amd.detectors  # noqa: F401  (registers the model classes)
    from dsl_amd.registry import build_detector
    det = build_detector(fcos_model_cfg())
    sd = {k: v.clone() for k, v in det.state_dict().items()}
    k0 = next(k for k in sd if k.endswith('conv_cls.weight'))
    new = {('module.' + k): (v + 1.0 if k == k0 else v) for k, v in sd.items()}
    fn = str(tmp_path / 'up.pth')
    torch.save(dict(state_dict=new, meta=dict(epoch=3)), fn)
    ck = E.load_checkpoint(det, fn)
    assert ck['meta']['epoch'] == 3
    assert torch.allclose(det.state_dict()[k0].float(), sd[k0].float() + 1.0, atol=1e-2)


def _reference_ap(dets, img_ids, cat_ids, anns, thr, lo=0.0, hi=1e10, max_dets=100):
    """An independent, deliberately naive restatement of COCOeval's bbox protocol for ONE IoU threshold and area range
    (pycocotools cocoeval.py evaluateImg + accumulate as published): used only to cross-check coco_bbox_eval."""
    def iou(d, g, crowd):
        ix = max(0.0, min(d[0] + d[2], g[0] + g[2]) - max(d[0], g[0]))
        iy = max(0.0, min(d[1] + d[3], g[1] + g[3]) - max(d[1], g[1]))
        inter = ix * iy
        union = d[2] * d[3] if crowd else d[2] * d[3] + g[2] * g[3] - inter
        return inter / union if union > 0 else 0.0
    aps = []
    for cat in cat_ids:
        rows, npos = [], 0                      # rows: (score, is_tp, is_ignored)
        for idx, img in enumerate(img_ids):
            gts = [a for a in anns[idx] if a['category_id'] == cat]
            g_ign = [bool(a.get('iscrowd', 0)) or not (lo <= a['bbox'][2] * a['bbox'][3] <= hi) for a in gts]
            npos += sum(1 for x in g_ign if not x)
            order = sorted(range(len(gts)), key=lambda i: g_ign[i])           # stable: regular gts first
            gts, g_ign = [gts[i] for i in order], [g_ign[i] for i in order]
            dd = sorted([d for d in dets if d['image_id'] == img and d['category_id'] == cat], key=lambda d: -d['score'])[:max_dets]
            taken = [False] * len(gts)
            for d in dd:
                best, m = min(thr, 1 - 1e-10), -1
                for gi, g in enumerate(gts):
                    if taken[gi] and not g.get('iscrowd', 0):
                        continue
                    if m >= 0 and not g_ign[m] and g_ign[gi]:
                        break
                    v = iou(d['bbox'], g['bbox'], bool(g.get('iscrowd', 0)))
                    if v < best:
                        continue
                    best, m = v, gi
                if m >= 0:
                    taken[m] = True
                    rows.append((d['score'], True, g_ign[m]))
                else:
                    a = d['bbox'][2] * d['bbox'][3]
                    rows.append((d['score'], False, not (lo <= a <= hi)))
        if npos == 0:
            continue
        rows = [r for r in sorted(rows, key=lambda r: -r[0]) if not r[2]]      # python's sort is stable, like mergesort
        tp = fp = 0
        rc, pr = [], []
        for _, hit, _ in rows:
            tp, fp = tp + hit, fp + (not hit)
            rc.append(tp / npos)
            pr.append(tp / (tp + fp))
        for i in range(len(pr) - 2, -1, -1):
            pr[i] = max(pr[i], pr[i + 1])
        q = []
        for k in range(101):
            t = k / 100.0
            j = next((i for i, r in enumerate(rc) if r >= t - 1e-12), None)
            q.append(pr[j] if j is not None else 0.0)
        aps.append(sum(q) / 101)
    return sum(aps) / len(aps) if aps else -1.0


@pytest.mark.parametrize('seed', [0, 1, 2])
def test_eval_random_case_vs_independent_restatement(seed):
    """~25 detections over 3 images / 3 categories with crowd boxes, duplicates, misses and all three area ranges: every
    figure of coco_bbox_eval against the naive restatement above (both follow the published COCOeval protocol; pycocotools
    itself is not installable here - parity with it stays unpinned)."""
    rng = np.random.RandomState(seed)
    img_ids, cat_ids = [11, 12, 13], [1, 2, 5]
    anns, dets = [], []
    for img in img_ids:
        a = []
        for _ in range(rng.randint(2, 6)):
            w, h = rng.choice([20, 50, 150]) * rng.uniform(.7, 1.3), rng.choice([20, 50, 150]) * rng.uniform(.7, 1.3)
            x, y = rng.uniform(0, 400), rng.uniform(0, 300)
            a.append(dict(bbox=[float(x), float(y), float(w), float(h)], category_id=int(rng.choice(cat_ids)), iscrowd=int(rng.rand() < .15)))
        anns.append(a)
        for g in a:
            for _ in range(rng.randint(0, 3)):           # jittered copies of the gt (hits at some thresholds, duplicates)
                j = rng.normal(0, .12, 4) * [g['bbox'][2], g['bbox'][3], g['bbox'][2], g['bbox'][3]]
                b = [g['bbox'][0] + j[0], g['bbox'][1] + j[1], max(g['bbox'][2] + j[2], 2.0), max(g['bbox'][3] + j[3], 2.0)]
                dets.append(dict(image_id=img, bbox=[float(v) for v in b], score=float(np.round(rng.uniform(.05, 1), 3)),
                                 category_id=g['category_id'] if rng.rand() < .85 else int(rng.choice(cat_ids))))
        for _ in range(2):                               # pure false positives
            dets.append(dict(image_id=img, bbox=[float(rng.uniform(0, 400)), float(rng.uniform(0, 300)), 30.0, 30.0],
                             score=float(np.round(rng.uniform(.05, 1), 3)), category_id=int(rng.choice(cat_ids))))
    m = E.coco_bbox_eval(dets, img_ids, cat_ids, anns)
    thrs = np.linspace(.5, .95, 10)
    ref_all = [_reference_ap(dets, img_ids, cat_ids, anns, t) for t in thrs]
    assert m['mAP'] == pytest.approx(np.mean(ref_all), abs=1e-9)
    assert m['mAP_50'] == pytest.approx(ref_all[0], abs=1e-9) and m['mAP_75'] == pytest.approx(ref_all[5], abs=1e-9)
    for key, (lo, hi) in (('mAP_s', (0, 32 ** 2)), ('mAP_m', (32 ** 2, 96 ** 2)), ('mAP_l', (96 ** 2, 1e10))):
        vals = [_reference_ap(dets, img_ids, cat_ids, anns, t, lo, hi) for t in thrs]
        want = np.mean(vals) if vals[0] >= 0 else -1.0
        assert m[key] == pytest.approx(want, abs=1e-9), key
    assert 0.0 < m['mAP'] < 1.0
