"""Teacher sweep on the GPU vs the reference's get_bboxes / multiclass_nms outputs (golden vectors)."""
import numpy as np
import pytest
import torch

from util import fcos_model_cfg, levels_to_flat

pytestmark = pytest.mark.gpu
T = torch.from_numpy


def _match(dets, labels, count, d, n):
    for i in range(n):
        k = int(count[i])
        rb, rl = T(d[f'det{i}']), T(d[f'lab{i}'])
        assert k == rb.shape[0], (k, rb.shape)
        got_b, got_l = dets[i, :k].cpu(), labels[i, :k].cpu()
        # same set, same order up to ties in score
        assert torch.allclose(got_b[:, 4], rb[:, 4], rtol=1e-4, atol=1e-6)
        order_ref = np.lexsort((rb[:, 0].numpy(), rl.numpy(), -rb[:, 4].numpy()))
        order_got = np.lexsort((got_b[:, 0].numpy(), got_l.numpy(), -got_b[:, 4].numpy()))
        assert torch.equal(got_l[order_got], rl[order_ref])
        assert torch.allclose(got_b[order_got], rb[order_ref], rtol=1e-4, atol=1e-3)


def test_detect_vs_reference_synth(golden):
    from dsl_amd.sweep import DetectPlan
    d = golden('bboxes_synth.npz')
    sizes = [tuple(int(v) for v in s) for s in d['sizes']]
    strides = (8, 16, 32, 64, 128)
    B = 2
    cls = levels_to_flat([T(d[f'cls{i}']) for i in range(5)]).contiguous().cuda()
    M = cls.shape[0]
    rc = torch.zeros(M, 8)
    rc[:, :4] = levels_to_flat([T(d[f'reg{i}']) / s for i, s in zip(range(5), strides)])
    rc[:, 4] = levels_to_flat([T(d[f'ctr{i}']) for i in range(5)])[:, 0]
    rc = rc.cuda()
    scales = torch.ones(5, device='cuda')
    dp = DetectPlan(B, sizes, strides, 'cuda')
    dp.bind(cls, rc, scales)
    shp = tuple(int(x) for x in d['img_shape'])
    dp.set_meta([shp] * B, [d['scale_factor']] * B, True)
    dp.run()
    torch.cuda.synchronize()
    _match(dp.dets, dp.labels, dp.count, d, B)


def test_simple_test_tiny_net(golden):
    """Whole sweep (bf16 network forward + detect) against the reference's detections: box sets agree for
    the confident detections; exact parity of the post-processing alone is the test above."""
    from dsl_amd import detectors  # noqa: F401
    from dsl_amd.registry import build_detector
    from oracle import fcos_oracle as O
    d = golden('sweep_tiny.npz')
    sd = O.synth_state_dict(0)
    sd['bbox_head.conv_cls.bias'] = torch.full((80,), float(d['cls_bias']))
    model = build_detector(fcos_model_cfg())
    model.load_state_dict(sd)
    model = model.cuda()
    shp = tuple(int(x) for x in d['img_shape'])
    metas = [dict(img_shape=shp, scale_factor=d['scale_factor'])] * 2
    res = model.simple_test(T(d['img']).cuda(), metas, rescale=True)
    assert len(res) == 2 and len(res[0]) == 80
    for i in range(2):
        got = np.concatenate(res[i])
        ref = d[f'det{i}']
        assert got.shape[1] == 5 and 80 <= got.shape[0] <= 100
        # top-20 reference detections are found with near-identical boxes
        top = ref[np.argsort(-ref[:, 4])[:20]]
        for r in top:
            dist = np.abs(got[:, :4] - r[:4]).max(1)
            j = dist.argmin()
            assert dist[j] < 1.0 and abs(got[j, 4] - r[4]) < 0.02, (r, got[j])


def test_detect_with_more_valid_pairs_than_the_candidate_buffer(golden):
    """bbox_nms.py:54-62 hands EVERY (location, class) pair above score_thr to the NMS; dsl_fcos_detect keeps the best 16 384 by
    final score first (detect.hip CAND_CAP).  Greedy NMS only ever lets a higher-scored box suppress a lower-scored one, so the
    survivors among the best 16 384 are exactly the uncapped NMS's survivors among them - the output can differ only if fewer than
    max_per_img boxes survive there.  Here ~80 000 of 81 920 pairs are valid (an untrained head looks like this): the kernel's
    detections equal the oracle's uncapped multiclass_nms, box for box (the oracle's result is a committed fixture,
    tests/golden/make_detect_many.py: its pure-Python NMS needs minutes on a busy host)."""
    import importlib.util
    import os
    from dsl_amd.sweep import DetectPlan
    spec = importlib.util.spec_from_file_location('make_detect_many', os.path.join(os.path.dirname(__file__), 'golden', 'make_detect_many.py'))
    M = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(M)
    d = golden('detect_many.npz')
    cls, reg, ctr = M.inputs()
    assert M.digest(cls + reg + ctr) == pytest.approx(float(d['digest']), rel=1e-7), 'the seeded inputs differ from the fixture\'s'      # (exp() differs in the last bits between hosts)
    sizes, strides, B, shp = M.SIZES, M.STRIDES, M.B, M.SHAPE
    n_valid = [int((torch.cat([c[i].permute(1, 2, 0).reshape(-1, 80) for c in cls]).sigmoid() > 0.05).sum()) for i in range(B)]
    assert min(n_valid) > 4 * 16384, n_valid
    cls_f = levels_to_flat(cls).contiguous().cuda()
    rc = torch.zeros(cls_f.shape[0], 8)
    rc[:, :4] = levels_to_flat([r / s for r, s in zip(reg, strides)])
    rc[:, 4] = levels_to_flat(ctr)[:, 0]
    dp = DetectPlan(B, sizes, strides, 'cuda')
    dp.bind(cls_f, rc.cuda(), torch.ones(5, device='cuda'))
    dp.set_meta([shp + (3,)] * B, [1.0] * B, True)
    dp.run()
    torch.cuda.synchronize()
    for i in range(B):
        k = int(dp.count[i])
        rb, rl = T(d[f'det{i}']), T(d[f'lab{i}'])
        assert k == rb.shape[0] == 100
        got_b, got_l = dp.dets[i, :k].cpu(), dp.labels[i, :k].cpu()
        assert torch.allclose(got_b[:, 4], rb[:, 4], rtol=1e-4, atol=1e-6)          # same scores in the same (descending) order
        assert torch.equal(got_l, rl) and torch.allclose(got_b[:, :4], rb[:, :4], rtol=1e-4, atol=1e-3)
