"""GPU parity of the fused FCOS assignment + loss kernels against the golden vectors produced by the
reference (tests/golden/*.npz) and against the CPU oracle on seeded inputs."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
T = torch.from_numpy


def flat_levels(ts):     # list of (B,C,h,w) -> (M, C) level-major
    return torch.cat([t.permute(0, 2, 3, 1).reshape(-1, t.shape[1]) for t in ts])


def run_plan(sizes, B, cls, reg, ctr, gtb, gtl, ig, loss_weight=1.0, soft_weight=0.0, scales=None):
    from dsl_amd.head_loss import FcosLossPlan
    plan = FcosLossPlan(B, sizes, 'cuda')
    plan.set_targets(gtb, gtl, ig)
    plan.configure(loss_weight=loss_weight, soft_weight=soft_weight)
    M = plan.M
    cls_d = flat_levels(cls).contiguous().cuda()
    rc = torch.zeros(M, 8)
    rc[:, :4] = flat_levels(reg)
    rc[:, 4] = flat_levels(ctr)[:, 0]
    rc_d = rc.cuda()
    sc = (torch.ones(5) if scales is None else scales).cuda()
    plan.bind_outputs(cls_d, rc_d, sc)
    plan.assign()
    plan.loss()
    torch.cuda.synchronize()
    return plan


@pytest.mark.parametrize('name', ['assign_small.npz', 'assign_full.npz'])
def test_assign_bit_exact_vs_reference(golden, name):
    from dsl_amd.head_loss import FcosLossPlan
    d = golden(name)
    sizes = [tuple(int(v) for v in s) for s in d['sizes']]
    n = int(d['n_img'])
    plan = FcosLossPlan(n, sizes, 'cuda')
    plan.set_targets([T(d[f'gt{i}']) for i in range(n)], [T(d[f'gl{i}']) for i in range(n)], None)
    plan.assign()
    torch.cuda.synchronize()
    assert torch.equal(plan.labels.cpu(), T(d['labels']).long())
    assert torch.equal(plan.bbox_targets.cpu(), T(d['bbox_targets']))     # bit exact
    npos = int(((T(d['labels']) >= 0) & (T(d['labels']) < 80)).sum())
    assert int(plan.stats[0]) == npos


def test_assign_indices_vs_oracle_random():
    from dsl_amd.head_loss import FcosLossPlan
    from oracle import fcos_oracle as O
    sizes = [(100, 168), (50, 84), (25, 42), (13, 21), (7, 11)]
    rng = np.random.RandomState(5)
    gtb, gtl, igb = [], [], []
    for n_ in (1, 13, 90):
        b = O.synth_boxes(rng, n_)
        gtb.append(T(b))
        gtl.append(T(rng.randint(0, 80, len(b)).astype('int64')))
        igb.append(T(O.synth_boxes(rng, 3)))
    plan = FcosLossPlan(3, sizes, 'cuda')
    plan.set_targets(gtb, gtl, igb)
    plan.configure(loss_weight=3.0)
    plan.assign()
    torch.cuda.synchronize()
    pts = O.get_points(sizes)
    labels, tg, idx = O.get_targets(pts, gtb, gtl)
    assert torch.equal(plan.labels.cpu(), torch.cat(labels))
    assert torch.equal(plan.assign_idx.cpu().long(), torch.cat(idx))
    pos = torch.cat(labels) < 80
    assert torch.equal(plan.bbox_targets.cpu()[pos], torch.cat(tg)[pos])
    cls = [torch.zeros(3, 80, h, w) for h, w in sizes]
    _, aux = O.fcos_loss(cls, [torch.ones(3, 4, h, w) for h, w in sizes], [torch.zeros(3, 1, h, w) for h, w in sizes],
                         gtb, gtl, igb, loss_weight=3.0, return_aux=True)
    assert torch.equal(plan.cls_weight.cpu(), aux['cls_weight'])
    assert float(plan.stats[1]) == pytest.approx(float(aux['ctr_targets'].sum()), rel=1e-5)


@pytest.mark.parametrize('name', ['loss_sup', 'loss_sup_ig', 'loss_dsl', 'loss_dsl_warm', 'loss_dsl_even',
                                  'loss_nopos'])
def test_loss_vs_reference_golden(golden, name):
    d = golden(name + '.npz')
    B = int(d['B'])
    sizes = [tuple(int(v) for v in s) for s in d['sizes']]
    cls = [T(d[f'cls{i}']) for i in range(5)]
    reg = [T(d[f'reg{i}']) for i in range(5)]
    ctr = [T(d[f'ctr{i}']) for i in range(5)]
    gtb = [T(d[f'gt{i}']) for i in range(B)]
    gtl = [T(d[f'gl{i}']) for i in range(B)]
    ig = [T(d[f'ig{i}']) for i in range(B)] if int(d['with_ig']) else None
    sw = float(d['soft_weight']) / 1000.0          # fixtures were generated inside the warm-up window
    plan = run_plan(sizes, B, cls, reg, ctr, gtb, gtl, ig, float(d['loss_weight']), sw)
    got = plan.losses.cpu()
    for i, k in enumerate(('loss_cls', 'loss_bbox', 'loss_centerness', 'loss_sisoft')):
        if k in d:
            assert float(got[i]) == pytest.approx(float(d[k]), rel=1e-4, abs=1e-6), k      # bar: 1e-3
    gc = flat_levels([T(d[f'gcls{i}']) for i in range(5)])
    gr = flat_levels([T(d[f'greg{i}']) for i in range(5)])
    gt_ = flat_levels([T(d[f'gctr{i}']) for i in range(5)])[:, 0]
    mine_c = plan.g_cls.float().cpu()
    assert float(mine_c[:, 80:].abs().max()) == 0.0
    tol = 2 ** -8
    assert torch.allclose(mine_c[:, :80], gc, rtol=tol, atol=tol * float(gc.abs().max()) * 0.05 + 1e-9)
    mine_r = plan.g_rc.float().cpu()
    raw = flat_levels(reg)
    on = raw > 0          # the fused kernel differentiates through ReLU(scale * conv_reg); the fixture stops at bbox_pred
    assert torch.allclose(mine_r[:, :4][on], gr[on], rtol=tol, atol=tol * float(gr.abs().max()) * 0.05 + 1e-9)
    assert float(mine_r[:, :4][~on].abs().max()) == 0.0
    assert torch.allclose(mine_r[:, 4], gt_, rtol=tol, atol=tol * float(gt_.abs().max()) * 0.05 + 1e-9)
    assert float(mine_r[:, 5:].abs().max()) == 0.0


def test_loss_scale_relu_chain_vs_oracle():
    """Gradient through Scale + ReLU (fcos_head.py:159-163) including d/dScale."""
    from oracle import fcos_oracle as O
    sizes = [(16, 24), (8, 12), (4, 6), (2, 3), (1, 2)]
    g = torch.Generator().manual_seed(0)
    B = 2
    rng = np.random.RandomState(3)
    gtb = [T(O.synth_boxes(rng, 4, H=128, W=192, lo=8, hi=150)) for _ in range(B)]
    gtl = [T(rng.randint(0, 80, len(b)).astype('int64')) for b in gtb]
    cls = [torch.randn(B, 80, h, w, generator=g) - 2 for h, w in sizes]
    raw = [(torch.randn(B, 4, h, w, generator=g) * 3 + 2).requires_grad_() for h, w in sizes]
    ctr = [torch.randn(B, 1, h, w, generator=g) for h, w in sizes]
    scales = (torch.rand(5, generator=g) + 0.5).requires_grad_()
    reg = [torch.relu(r * scales[i]) for i, r in enumerate(raw)]
    out = O.fcos_loss(cls, reg, ctr, gtb, gtl, None)
    sum(out.values()).backward()
    plan = run_plan(sizes, B, cls, [r.detach() for r in raw], ctr, gtb, gtl, None, scales=scales.detach())
    got = plan.losses.cpu()
    assert float(got[1]) == pytest.approx(float(out['loss_bbox']), rel=1e-4)
    assert torch.allclose(plan.g_scales.cpu(), scales.grad, rtol=1e-3, atol=1e-5)
    gr = flat_levels([r.grad for r in raw])
    assert torch.allclose(plan.g_rc.float().cpu()[:, :4], gr, rtol=2 ** -7, atol=1e-5)


def test_points_and_deterministic_sums():
    """dsl_fcos_points == get_points (anchor_free_head.py:287-321, fcos_head.py:550-560); the loss / num_pos sums are added
    in a fixed order (block records + one finalize pass): bit-identical on a re-run."""
    import ctypes as C
    from dsl_amd import _lib as L
    from dsl_amd.head_loss import FcosLossPlan
    from oracle import fcos_oracle as O
    sizes = [(25, 42), (13, 21), (7, 11), (4, 6), (2, 3)]
    plan = FcosLossPlan(2, sizes, 'cuda')
    P = sum(h * w for h, w in sizes)
    pts = torch.empty(P, 2, device='cuda')
    L.check(L.lib.dsl_fcos_points(C.byref(plan.desc), L.ptr(pts), L.stream_ptr()))
    torch.cuda.synchronize()
    assert torch.equal(pts.cpu(), torch.cat(O.get_points(sizes)))
    g = torch.Generator().manual_seed(2)
    rng = np.random.RandomState(2)
    cls = [torch.randn(2, 80, h, w, generator=g) - 2 for h, w in sizes]
    reg = [torch.rand(2, 4, h, w, generator=g) * 4 for h, w in sizes]
    ctr = [torch.randn(2, 1, h, w, generator=g) for h, w in sizes]
    gtb = [T(O.synth_boxes(rng, 6, H=200, W=336, lo=8, hi=150)) for _ in range(2)]
    gtl = [T(rng.randint(0, 80, len(b)).astype('int64')) for b in gtb]
    runs = []
    for _ in range(2):
        p = run_plan(sizes, 2, cls, reg, ctr, gtb, gtl, None)
        runs.append((p.losses.clone(), p.stats.clone(), p.g_scales.clone(), p.g_cls.clone(), p.g_rc.clone()))
    for a, b in zip(*runs):
        assert torch.equal(a, b)
    assert float(runs[0][1][0]) > 0 and float(runs[0][1][2:].abs().sum()) == 0
