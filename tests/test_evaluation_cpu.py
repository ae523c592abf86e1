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
