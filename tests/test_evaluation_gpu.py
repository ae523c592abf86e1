"""Row f4 on the GPU: EvalHook drives the HIP sweep over a validation loader with the EMA teacher, exports the COCO result
json and evaluates it; train_detector(validate=True) wires it in."""
import json
import os

import numpy as np
import pytest
import torch

from util import fcos_model_cfg

pytestmark = pytest.mark.gpu


def build(bias=0.0):
    from dsl_amd import detectors  # noqa: F401
    from dsl_amd.registry import build_detector
    from oracle import fcos_oracle as O
    m = build_detector(fcos_model_cfg())
    sd = O.synth_state_dict(0)
    sd['bbox_head.conv_cls.bias'] = torch.full_like(sd['bbox_head.conv_cls.bias'], bias)
    m.load_state_dict(sd)
    return m.cuda()


def test_single_gpu_test_matches_simple_test_and_json_round_trip(tmp_path):
    from dsl_amd import evaluation as E
    from dsl_amd.data import SyntheticValLoader
    from dsl_amd.sweep import simple_test
    det = build()
    loader = SyntheticValLoader(n_images=3, H=128, W=192, W_img=190, samples_per_gpu=2)
    res = E.single_gpu_test(det, loader)
    assert len(res) == 3 and all(len(r) == 80 for r in res)
    batches = list(loader)
    direct = simple_test(det, batches[0]['img'][0], batches[0]['img_metas'][0], rescale=True)
    for a, b in zip(res[:2], direct):
        for ca, cb in zip(a, b):
            np.testing.assert_array_equal(ca, cb)
    assert sum(len(c) for r in res for c in r) > 0
    files, _ = E.format_results(res, loader.img_ids, loader.cat_ids, str(tmp_path / 'val'))
    js = json.load(open(files['bbox']))
    assert len(js) == sum(len(c) for r in res for c in r)
    # detections used as ground truth evaluate to a perfect score: the export and the evaluator agree on boxes and ids
    anns = [[] for _ in loader.img_ids]
    for d in js:
        anns[loader.img_ids.index(d['image_id'])].append(dict(bbox=d['bbox'], category_id=d['category_id'], iscrowd=0))
    # identical boxes of one class within an image would steal each other's match; NMS has removed those
    m = E.coco_bbox_eval(js, loader.img_ids, loader.cat_ids, anns)
    assert m['mAP'] == pytest.approx(1.0, abs=1e-6)
    m = E.coco_bbox_eval(js, loader.img_ids, loader.cat_ids, loader.annotations)
    assert 0.0 <= m['mAP'] < 0.2                              # random weights on random boxes


def test_eval_hook_uses_the_teacher_once_it_exists(tmp_path):
    from dsl_amd import evaluation as E
    from dsl_amd.data import SyntheticValLoader
    from dsl_amd.runner import SemiEpochBasedRunner
    student, teacher = build(0.0), build(-20.0)               # the teacher's scores are all below score_thr: no detections
    runner = SemiEpochBasedRunner(student, optimizer=None, max_epochs=1, ema_model=teacher, work_dir=str(tmp_path))
    loader = SyntheticValLoader(n_images=2, H=128, W=192, W_img=190)
    hook = E.EvalHook(loader, interval=1, metric='bbox')
    runner.ema_flag = False
    m_student = hook._do_evaluate(runner)
    n_student = len(json.load(open(os.path.join(str(tmp_path), 'eval_epoch_1.bbox.json'))))
    runner.ema_flag = True
    m_teacher = hook._do_evaluate(runner)
    n_teacher = len(json.load(open(os.path.join(str(tmp_path), 'eval_epoch_1.bbox.json'))))
    assert n_student > 0 and n_teacher == 0
    assert set(m_student) == {'bbox_mAP', 'bbox_mAP_50', 'bbox_mAP_75', 'bbox_mAP_s', 'bbox_mAP_m', 'bbox_mAP_l'}
    assert m_teacher['bbox_mAP'] == 0.0 and len(hook.history) == 2
    # interval / start logic of the epoch hook
    hook2 = E.EvalHook(loader, interval=2)
    fired = []
    hook2._do_evaluate = lambda r: fired.append(r.epoch + 1)
    for ep in range(6):
        runner._epoch = ep
        hook2.after_train_epoch(runner)
    assert fired == [2, 4, 6]
