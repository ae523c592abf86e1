"""Pin the CPU oracle (oracle/fcos_oracle.py) against
  (1) the reference's own known-answer tests for this path, and
  (2) vectors produced by running the reference's python files (tests/golden/make_golden.py).
CPU only."""
import numpy as np
import pytest
import torch

from oracle import fcos_oracle as O

T = torch.from_numpy


def test_kat_giou_reference_test_box_overlap():
    # /root/reference/tests/test_metrics/test_box_overlap.py:83-97
    b1 = torch.tensor([[0., 0., 10., 10.], [10., 10., 20., 20.], [32., 32., 38., 42.]])
    b2 = torch.tensor([[0., 0., 10., 20.], [0., 10., 10., 19.], [10., 10., 20., 20.]])
    g = O.giou_aligned(b1, b2)
    assert torch.allclose(g, torch.tensor([0.5000, -0.0500, -0.8214]), atol=1e-4)


def test_kat_distance2bbox_reference_test_misc():
    # /root/reference/tests/test_utils/test_misc.py:51-92
    point = torch.tensor([[74., 61.], [-29., 106.], [138., 61.], [29., 170.]])
    distance = torch.tensor([[0., 0, 1., 1.], [1., 2., 10., 6.], [22., -29., 138., 61.],
                             [54., -29., 170., 61.]])
    expected = torch.tensor([[74., 61., 75., 62.], [0., 104., 0., 112.], [100., 90., 100., 120.],
                             [0., 120., 100., 120.]])
    out = O.distance2bbox(point, distance, max_shape=(120, 100))
    assert expected.allclose(out)
    out = O.distance2bbox(point, distance, max_shape=torch.tensor((120, 100)))
    assert expected.allclose(out)
    batch = O.distance2bbox(point.unsqueeze(0).repeat(2, 1, 1), distance.unsqueeze(0).repeat(2, 1, 1),
                            max_shape=[(120, 100)] * 2)
    assert expected.unsqueeze(0).repeat(2, 1, 1).allclose(batch)
    rois = torch.zeros((0, 4))
    deltas = torch.zeros((0, 4))
    assert O.distance2bbox(rois, deltas, max_shape=(120, 100)).shape == (0, 4)


def test_focal_zero_weight_is_zero():
    # /root/reference/tests/test_models/test_loss.py:17-23 (weights = 0 => loss 0)
    pred = torch.rand(10, 80)
    lab = torch.randint(0, 81, (10,))
    assert float((O.focal_loss_elem(pred, lab) * torch.zeros(10, 1)).sum()) == 0.0


@pytest.mark.parametrize('name', ['assign_small.npz', 'assign_full.npz'])
def test_assign_matches_reference(golden, name):
    d = golden(name)
    sizes = [tuple(s) for s in d['sizes']]
    n = int(d['n_img'])
    pts = O.get_points(sizes)
    gtb = [T(d[f'gt{i}']) for i in range(n)]
    gtl = [T(d[f'gl{i}']) for i in range(n)]
    labels, tg, _ = O.get_targets(pts, gtb, gtl)
    assert torch.equal(torch.cat(labels), T(d['labels']).long())
    assert torch.equal(torch.cat(tg), T(d['bbox_targets']))     # bit exact
    if 'points' in d:
        assert torch.equal(torch.cat(pts), T(d['points']))


@pytest.mark.parametrize('name', ['loss_sup', 'loss_sup_ig', 'loss_dsl', 'loss_dsl_warm',
                                  'loss_dsl_even', 'loss_nopos'])
def test_loss_matches_reference(golden, name):
    d = golden(name + '.npz')
    B = int(d['B'])
    cls = [T(d[f'cls{i}']).requires_grad_() for i in range(5)]
    reg = [T(d[f'reg{i}']).requires_grad_() for i in range(5)]
    ctr = [T(d[f'ctr{i}']).requires_grad_() for i in range(5)]
    gtb = [T(d[f'gt{i}']) for i in range(B)]
    gtl = [T(d[f'gl{i}']) for i in range(B)]
    ig = [T(d[f'ig{i}']) for i in range(B)] if int(d['with_ig']) else None
    # the fixtures were generated with head.cur_iter = 0, so `soft_warm_up >= cur_iter` holds on the
    # first call for every warm-up value >= 0 and the reference scales by 1/1000 (fcos_head.py:323-326)
    out = O.fcos_loss(cls, reg, ctr, gtb, gtl, ig, loss_weight=float(d['loss_weight']),
                      soft_weight=float(d['soft_weight']), soft_scale=1 / 1000.0)
    keys = [k for k in ('loss_cls', 'loss_bbox', 'loss_centerness', 'loss_sisoft') if k in d]
    for k in keys:
        assert float(out[k].detach()) == pytest.approx(float(d[k]), rel=2e-6, abs=1e-7), k
    assert set(keys) == set(out.keys())
    sum(out.values()).backward()
    for i in range(5):
        for nm, t in (('gcls', cls), ('greg', reg), ('gctr', ctr)):
            ref = T(d[f'{nm}{i}'])
            got = t[i].grad if t[i].grad is not None else torch.zeros_like(ref)
            assert torch.allclose(got, ref, rtol=1e-5, atol=1e-8), (nm, i)


@pytest.mark.parametrize('name', ['net_tiny', 'net_small_dsl'])
def test_whole_step_matches_reference(golden, name):
    d = golden(name + '.npz')
    B = int(d['B'])
    sd = O.synth_state_dict(0)
    img = T(d['img'])
    gtb = [T(d[f'gt{i}']) for i in range(B)]
    gtl = [T(d[f'gl{i}']) for i in range(B)]
    dsl = bool(int(d['dsl']))
    ig = [T(d[f'ig{i}']) for i in range(B)] if dsl else None
    kw = dict(loss_weight=3.0, soft_weight=1.0, soft_scale=1.0) if dsl else {}
    losses, grads, aux = O.train_step(sd, img, gtb, gtl, ig, **kw)
    for k in losses:
        assert losses[k] == pytest.approx(float(d[k]), rel=1e-4), k
    for i in range(5):
        assert torch.allclose(aux['cls'][i], T(d[f'cls{i}']), rtol=1e-3, atol=1e-4)
        assert torch.allclose(aux['reg'][i], T(d[f'reg{i}']), rtol=1e-3, atol=1e-4)
        assert torch.allclose(aux['ctr'][i], T(d[f'ctr{i}']), rtol=1e-3, atol=1e-4)
    keys = [str(k) for k in d['grad_keys']]
    assert keys == O.trainable_keys(sd)
    norms = np.array([float(grads[k].norm()) for k in keys])
    assert np.allclose(norms, d['grad_norms'], rtol=2e-3, atol=1e-6)
    for k in d.files:
        if k.startswith('grad/'):
            assert torch.allclose(grads[k[5:]], T(d[k]), rtol=1e-3, atol=1e-5 * float(T(d[k]).abs().max())), k


def _check_dets(dets, d, n):
    for i in range(n):
        b, l = dets[i]
        rb, rl = T(d[f'det{i}']), T(d[f'lab{i}'])
        assert b.shape == rb.shape
        assert torch.equal(l, rl)
        assert torch.allclose(b, rb, rtol=1e-5, atol=1e-4)


def test_get_bboxes_matches_reference(golden):
    d = golden('bboxes_synth.npz')
    cls = [T(d[f'cls{i}']) for i in range(5)]
    reg = [T(d[f'reg{i}']) for i in range(5)]
    ctr = [T(d[f'ctr{i}']) for i in range(5)]
    shp = tuple(int(x) for x in d['img_shape'])
    dets = O.get_bboxes(cls, reg, ctr, [shp] * 2, [d['scale_factor']] * 2)
    _check_dets(dets, d, 2)


def test_sweep_matches_reference(golden):
    d = golden('sweep_tiny.npz')
    sd = O.synth_state_dict(0)
    sd['bbox_head.conv_cls.bias'] = torch.full((80,), float(d['cls_bias']))
    with torch.no_grad():
        cls, reg, ctr = O.extract_and_head(sd, T(d['img']), O.Quant(False), training=False)
    for i in range(5):
        assert torch.allclose(cls[i], T(d[f'cls{i}']), rtol=1e-3, atol=1e-4)
        assert torch.allclose(reg[i], T(d[f'reg{i}']), rtol=1e-3, atol=1e-3)
    shp = tuple(int(x) for x in d['img_shape'])
    # NMS on the reference's own head outputs (bit-identical inputs -> identical keep set)
    dets = O.get_bboxes([T(d[f'cls{i}']) for i in range(5)], [T(d[f'reg{i}']) for i in range(5)],
                        [T(d[f'ctr{i}']) for i in range(5)], [shp] * 2, [d['scale_factor']] * 2)
    _check_dets(dets, d, 2)


def test_ema_and_sgd_formulas():
    t = {'w': torch.ones(3), 'n': torch.tensor(5)}
    s = {'w': torch.zeros(3), 'n': torch.tensor(105)}
    e = O.ema_update(t, s, 0.99)
    assert torch.allclose(e['w'], torch.full((3,), 0.99))
    assert int(e['n']) == 6 and e['n'].dtype == torch.long
    # one torch.optim.SGD step as the known answer for sgd_step
    p = torch.nn.Parameter(torch.tensor([1.0, -2.0]))
    opt = torch.optim.SGD([p], lr=0.01, momentum=0.9, weight_decay=1e-4)
    params, bufs = {'x.weight': p.detach().clone()}, {}
    for step in range(3):
        g = torch.tensor([0.5, 0.25]) * (step + 1)
        p.grad = g.clone()
        opt.step()
        params, bufs = O.sgd_step(params, {'x.weight': g}, bufs, first_step=(step == 0))
        assert torch.allclose(params['x.weight'], p.detach(), rtol=1e-6)


def test_adaptive_thresholds_match_reference_adathres():
    """Pseudo-label refresh statistics: per-class thresholds and class weights equal the
    reference's `adathres` (unlabel_pred_hook.py:295-367) on the same pseudo-label files, for the first call and for
    a call with the previous call's history (fixture: tests/golden/adathres.json, made by make_golden.py adathres)."""
    import json
    import os
    from dsl_amd.runner import UnlabelPredHook, adaptive_thresholds
    d = json.load(open(os.path.join(os.path.dirname(__file__), 'golden', 'adathres.json')))
    idx = {n: i for i, n in enumerate(d['names'])}
    prev = None
    hook = UnlabelPredHook()
    for rnd in d['rounds']:
        by_c = {}
        hook.bank.entries.clear()              # the round's files replace the previous round's; the thresholds carry over
        for k, (tags, scores) in enumerate(rnd['per_img']):
            for t, s in zip(tags, scores):
                by_c.setdefault(idx[t], []).append(s)
            hook.bank.put(f'im{k}', np.zeros((len(tags), 4), np.int64), [idx[t] for t in tags], scores)
        thr, w = adaptive_thresholds(by_c, prev)
        assert {d['names'][c] for c in thr} == set(rnd['thres'])
        for name, v in rnd['thres'].items():
            assert thr[idx[name]] == pytest.approx(v, rel=1e-9), name
        for name, v in rnd['weights'].items():
            assert w[idx[name]] == pytest.approx(v, rel=1e-9), name
        # the hook's own bookkeeping gives the same numbers
        hook.update_thresholds()
        assert hook.thres == pytest.approx(thr) and hook.class_weights == pytest.approx(w)
        prev = thr


def test_pseudo_label_split_matches_reference_parse_ann_info():
    """(gt, ignore) split of stored pseudo labels == SemiCOCODataset._parse_ann_info (semicoco.py:184-291) on the same
    files: default band before any threshold file exists, per-class thresholds (with unseen classes) afterwards."""
    import json
    import os
    from dsl_amd.runner import split_pseudo_labels
    d = json.load(open(os.path.join(os.path.dirname(__file__), 'golden', 'pseudo_split.json')))
    idx = {n: i for i, n in enumerate(d['names'])}
    for mode in d['modes']:
        thr = {idx[k]: v for k, v in mode['thres'].items()} if mode['thres'] else {}
        for img, ref in zip(d['imgs'], mode['outs']):
            gt, gl, ig = split_pseudo_labels(np.array(img['rects'], np.float64).reshape(-1, 4), [idx[t] for t in img['tags']],
                                             img['scores'], thr, img_wh=tuple(d['wh']))
            assert gt.tolist() == ref['bboxes'] and gl.tolist() == ref['labels'] and ig.tolist() == ref['ignore'], mode['mode']


def test_parse_det_results_matches_reference():
    """Stored pseudo labels == parse_det_results + score sort (unlabel_pred_hook.py:20-57): threshold, int()
    truncation, 6-decimal scores, highest score first."""
    import json
    import os
    from dsl_amd.runner import parse_det_results
    d = json.load(open(os.path.join(os.path.dirname(__file__), 'golden', 'parse_dets.json')))
    for case in d['cases']:
        dets = np.concatenate([np.array(r, np.float32).reshape(-1, 5) for r in case['results']])
        labels = np.concatenate([np.full(len(r), c) for c, r in enumerate(case['results'])])
        got = parse_det_results(dets, labels, d['score_thr'])
        assert got['scores'].tolist() == [o['score'] for o in case['out']]
        assert got['tags'].tolist() == [o['c'] for o in case['out']]
        assert got['rects'].tolist() == [o['bbox'] for o in case['out']]


def test_rla_resnet_oracle_matches_reference(golden):
    """oracle/rla_oracle.py == the reference's RLA_ResNet (resnet_rla.py, imported unmodified by make_golden.py rla): stage
    outputs, the set of trainable tensors, every gradient norm and the full gradients of thirteen representative parameters
    (conv / trainable eval-mode BN / downsample / shared conv_out and recurrent conv / per-block stage BN)."""
    from oracle import fcos_oracle as O
    from oracle import rla_oracle as RO
    d = golden('rla_tiny.npz')
    sd = RO.synth_state_dict(0)
    tk = [k for k in RO.trainable_keys(sd) if k.startswith('backbone.')]
    assert len(tk) == int(d['n_train']) and sorted(tk) == sorted(str(k) for k in d['grad_keys'])
    p = {k: (v.clone().requires_grad_(k in tk) if v.is_floating_point() else v) for k, v in sd.items()}
    outs = RO.rla_resnet_forward(p, torch.from_numpy(d['x']), O.Quant(False))
    for i, o in enumerate(outs):
        assert torch.allclose(o, torch.from_numpy(d[f'out{i}']), rtol=1e-4, atol=1e-5), i
    sum((o * torch.from_numpy(d[f'r{i}'])).sum() for i, o in enumerate(outs)).backward()
    norms = dict(zip((str(k) for k in d['grad_keys']), d['grad_norms']))
    for k in tk:
        assert float(p[k].grad.norm()) == pytest.approx(float(norms[k]), rel=2e-3, abs=1e-6), k
    full = [k for k in d.files if k.startswith('grad:')]
    assert len(full) == 13
    for k in full:
        ref = torch.from_numpy(d[k])
        assert torch.allclose(p[k[5:]].grad, ref, rtol=2e-3, atol=2e-4 * float(ref.abs().max())), k
