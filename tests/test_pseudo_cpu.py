"""Pseudo-label bookkeeping on the host: the label-file step against the reference's own save_results2file (golden),
the bank's annotation semantics, thresholds, the reference's file formats, and the refresh schedule of the hook."""
import json
import os

import numpy as np
import pytest
import torch

from dsl_amd.pseudo import PseudoLabelBank, fuse_host

GOLDEN = os.path.join(os.path.dirname(__file__), 'golden')


def test_fuse_matches_reference_save_results2file():
    """parse_det_results + score sort + per-class second NMS == save_results2file (unlabel_pred_hook.py:84-171,
    fuse=False) run from the reference's source (tests/golden/make_golden.py fuse), bit for bit."""
    d = json.load(open(os.path.join(GOLDEN, 'fuse.json')))
    for c in d['cases']:
        got = fuse_host(np.array(c['dets'], np.float32), np.array(c['labels']), d['infer_score_thre'], c['iou'],
                        d['nms_score_thr'], num_classes=len(d['id2cat']) - 1)
        assert got['rects'].tolist() == c['rects'] and got['tags'].tolist() == c['tags']
        assert got['scores'].tolist() == c['scores'] and len(got['tags']) == c['targetNum']


def test_fuse_history_matches_reference_save_results2file():
    """fuse=True (unlabel_pred_hook.py:131-141): three successive refreshes of one label file, with and without
    first_ignore, as the reference's own save_results2file wrote them (tests/golden/make_golden.py fuse_hist)."""
    d = json.load(open(os.path.join(GOLDEN, 'fuse_hist.json')))
    for c in d['cases']:
        for r in c['rounds']:
            old = None if r['first_ignore'] else dict(rects=r['old_rects'], tags=r['old_tags'], scores=r['old_scores'])
            got = fuse_host(np.array(r['dets'], np.float32), np.array(r['labels']), d['infer_score_thre'], c['iou'],
                            d['nms_score_thr'], num_classes=len(d['id2cat']) - 1, old=old)
            assert got['rects'].tolist() == r['rects'] and got['tags'].tolist() == r['tags']
            assert got['scores'].tolist() == r['scores'] and len(got['tags']) == r['targetNum']


def test_bank_annotation_modes_and_files(tmp_path):
    names = [f'c{i}' for i in range(4)]
    rects = [[0, 0, 50, 40], [10, 10, 30, 30], [5, 5, 5.5, 30], [200, 200, 260, 260]]
    tags, scores = [0, 1, 2, 3], [0.9, 0.2, 0.8, 0.31]
    for thres, want_gt, want_ig in ((None, 3, 0), ([0.1, 0.4], 1, 2), ('adathres.json', 2, 1)):
        bank = PseudoLabelBank(num_classes=4, class_names=names, thres=thres)
        bank.put('a.jpg', rects, tags, scores)
        gt, gl, ig = bank.ann_info('a.jpg', img_wh=(300, 300))            # the 0.5 px wide box is dropped (w < 1)
        assert (len(gt), len(ig)) == (want_gt, want_ig) and len(gl) == len(gt), thres
    # adaptive mode: per-class thresholds replace the default band once computed
    bank.thres = {3: 0.35}
    gt, gl, ig = bank.ann_info('a.jpg', img_wh=(300, 300))
    assert len(gt) == 1 and len(ig) == 2 and gl.tolist() == [0]
    assert bank.ann_info('unknown.jpg')[0].shape == (0, 4)
    # the reference's files: per-image JSON and threshold file, round trip
    p = bank.export_json('a.jpg', str(tmp_path))
    j = json.load(open(p))
    assert set(j) == {'imageName', 'targetNum', 'rects', 'tags', 'masks', 'scores'} and j['tags'] == names and j['targetNum'] == 4
    b2 = PseudoLabelBank(num_classes=4, class_names=names, thres='adathres.json')
    b2.import_json(p)
    assert b2['a.jpg']['tags'].tolist() == tags and np.allclose(b2['a.jpg']['scores'], scores)
    bank.update_thresholds()
    bank.export_thres(str(tmp_path / 'adathres.json'))
    t = json.load(open(tmp_path / 'adathres.json'))
    assert set(t) == {'cat', 'id', 'thres'} and all(0.3 <= v <= 0.35 for v in t['thres'].values())
    # merge keeps the newest record of a name
    b2.put('a.jpg', rects[:1], tags[:1], scores[:1], stamp=5)
    bank.merge({'a.jpg': b2['a.jpg'], 'b.jpg': b2['a.jpg']})
    assert len(bank['a.jpg']['tags']) == 1 and 'b.jpg' in bank
    bank.merge({'a.jpg': dict(rects=rects, tags=tags, scores=scores, stamp=2)})
    assert len(bank['a.jpg']['tags']) == 1


class _Runner:
    def __init__(self, loader):
        self.iter = self.epoch = 0
        self.data_loader = loader
        self.iter_tol_epoch = len(loader)
        self.ema_flag, self.ema_model, self.model = False, None, None


def test_refresh_schedule_matches_reference_rule():
    """after_train_iter wakes up as unlabel_pred_hook.py:455-469: iteration mode, from start_point epochs on, every
    `interval` iterations counted from start_point; first a full sweep, then one upcoming image per firing."""
    from dsl_amd.data import SyntheticSemiLoader
    from dsl_amd.runner import UnlabelPredHook
    bank = PseudoLabelBank(num_classes=80, thres='adathres.json')
    loader = SyntheticSemiLoader(bank, n_labeled=3, n_unlabeled=5, iters_per_epoch=5, H=32, W=64, device='cpu')
    hook = UnlabelPredHook(dict(infer_score_thre=0.1, use_ema=True, start_point=1, preload=6, first_fuse=False,
                                eval_config=dict(iou=[0.6]), eval_checkpoint_config=dict(interval=2, mode='iteration')),
                           None, 'Det', interval_mode='iteration', interval=2, bank=bank)
    calls = []
    hook.refresh_all = lambda r: calls.append(('all', r.iter))
    hook.refresh_names = lambda r, names, thr=None: calls.append(('one', r.iter, tuple(names)))
    r = _Runner(loader)
    for ep in range(3):
        for _ in loader:
            hook.after_train_iter(r)
            r.iter += 1
        r.epoch += 1
    # eligible from iter + 1 >= 1 * 5 + 1 (the second epoch); fires when (iter + 1 - 1) % 2 == 0
    # (the last iteration of an epoch has no upcoming image: the reference's runner.ITER is exhausted there, :521-526)
    fired = [c[1] for c in calls]
    assert fired == [it for it in range(15) if it + 1 >= 6 and (it + 1 - 1) % 2 == 0 and it % 5 != 4], fired
    assert calls[0][0] == 'all' and all(c[0] == 'one' for c in calls[1:])
    # the image named is the one the loader hands out next (prefetch depth 0)
    bank2 = PseudoLabelBank(num_classes=80, thres='adathres.json')
    l2 = SyntheticSemiLoader(bank2, n_labeled=3, n_unlabeled=5, iters_per_epoch=5, H=32, W=64, device='cpu')
    it = iter(l2)
    next(it)
    upcoming = l2.unlabeled.upcoming(0)
    assert next(it)['img_metas'][1]['filename'] == upcoming[0]


def test_loader_injects_bank_labels():
    from dsl_amd.data import SyntheticSemiLoader
    bank = PseudoLabelBank(num_classes=80, thres='adathres.json')
    loader = SyntheticSemiLoader(bank, n_labeled=2, n_unlabeled=3, iters_per_epoch=3, H=64, W=96, device='cpu')
    first = next(iter(loader))
    assert first['gt_bboxes'][1].shape == (0, 4) and first['gt_bboxes_ignore'][1].shape == (0, 4)     # nothing stored yet
    name = loader.unlabeled.upcoming(0)[0]
    bank.put(name, [[4, 4, 40, 40], [10, 10, 60, 50], [1, 1, 2, 2.5]], [7, 9, 3], [0.8, 0.2, 0.05])
    b = next(loader)
    assert b['img_metas'][1]['filename'] == name
    assert b['gt_bboxes'][1].tolist() == [[4, 4, 40, 40], [1, 1, 2, 2.5]] and b['gt_labels'][1].tolist() == [7, 3]   # 0.05 < band: a gt box, as the reference
    assert b['gt_bboxes_ignore'][1].tolist() == [[10, 10, 60, 50]]
    assert b['gt_bboxes_ignore'][0].shape == (0, 4) and b['img'].shape == (2, 3, 64, 96)
