"""Whole training step on the GPU (forward + loss + hand-written backward through the C ABI) against
(1) golden vectors produced by the reference itself and (2) the CPU oracle, fp32 and bf16-emulating."""
import numpy as np
import pytest
import torch

from util import fcos_model_cfg, levels_to_flat, rel_l2

pytestmark = pytest.mark.gpu
T = torch.from_numpy


def build(**head):
    from dsl_amd import detectors  # noqa: F401  (registers the classes)
    from dsl_amd.registry import build_detector
    from oracle import fcos_oracle as O
    cfg = fcos_model_cfg(**head)
    model = build_detector(cfg)
    model.load_state_dict(O.synth_state_dict(0))
    return model.cuda()


_KNOBS = [('net_tiny', {}), ('net_small_dsl', {}), ('net_tiny', dict(side='0')), ('net_small_dsl', dict(side='0')),
          ('net_tiny', dict(img_split='', img_split_bwd='')), ('net_tiny', dict(bneck_fwd='')), ('net_tiny', dict(bneck_fwd='23')),
          ('net_tiny', dict(pipe_prefix='0')), ('net_small_dsl', dict(tower_slots='128', tail_slots='128'))]


@pytest.mark.parametrize('name,knobs', _KNOBS, ids=[n + ''.join(f'-{k}={v}' for k, v in kn.items()) for n, kn in _KNOBS])
def test_train_step_vs_reference_and_oracle(golden, monkeypatch, name, knobs):
    """Also under every schedule knob of dsl_amd/tuning.py set away from its default (side=0: every launch on the caller's stream -
    ADVICE round 4 asked for a gradient check of the single-stream schedule; no image-split chains; unfused / all fused bottleneck
    stages; inline prefix; other weight-gradient budgets): the same checks against the reference's vectors."""
    from oracle import fcos_oracle as O
    from dsl_amd import tuning
    tuning.tune('side')                                    # (DSL_TUNE parsed before the overrides below)
    for k_, v_ in knobs.items():
        monkeypatch.setitem(tuning._values, k_, v_)
    side = knobs.get('side', '1')
    d = golden(name + '.npz')
    B, dsl = int(d['B']), bool(int(d['dsl']))
    head = dict(loss_weight=3.0, soft_weight=1.0, soft_warm_up=0) if dsl else {}
    model = build(**head)
    if dsl:
        model.bbox_head.cur_iter = 1          # as in the fixture: past the warm-up window
    img = T(d['img'])
    gtb = [T(d[f'gt{i}']) for i in range(B)]
    gtl = [T(d[f'gl{i}']) for i in range(B)]
    ig = [T(d[f'ig{i}']) for i in range(B)] if dsl else None
    metas = [dict(img_shape=tuple(img.shape[2:]) + (3,), pad_shape=tuple(img.shape[2:]) + (3,), scale_factor=1.0)] * B
    losses = model.forward_train(img.cuda(), metas, gtb, gtl, ig)
    total = sum(losses.values())
    total.backward()
    torch.cuda.synchronize()
    got = {k: float(v.detach()) for k, v in losses.items()}
    # (a) the reference itself (fp32): documented bf16 tolerance
    for k in got:
        assert got[k] == pytest.approx(float(d[k]), rel=3e-2), (k, got[k], float(d[k]))
    # (b) oracle with bf16 storage emulation at the same points: the 1e-3 bar of north_star
    sd = O.synth_state_dict(0)
    kw = dict(loss_weight=3.0, soft_weight=1.0, soft_scale=1.0) if dsl else {}
    ol, og, aux = O.train_step(sd, img, gtb, gtl, ig, emulate_bf16=True, **kw)
    print(name, 'losses hip', got, 'oracle-bf16', ol, 'ref-fp32', {k: float(d[k]) for k in got})
    for k in got:
        assert got[k] == pytest.approx(ol[k], rel=3e-3), (k, got[k], ol[k])
    plan = next(iter(model._engine.plans.values()))
    assert bool(plan.BR) == (side == '1')
    cls = plan.bufs['cls_logits'].cpu()
    ref_cls = levels_to_flat([c.detach() for c in aux['cls']])
    assert rel_l2(cls, ref_cls) < 5e-3
    rc = plan.bufs['regctr'].cpu()
    # assignment indices are identical to the fp32 reference (exact arithmetic, independent of bf16)
    _, raux = O.fcos_loss(aux['cls'], aux['reg'], aux['ctr'], gtb, gtl, ig, return_aux=True, **kw)
    assert torch.equal(plan.lossplan.labels.cpu(), raux['labels'])
    assert torch.equal(plan.lossplan.assign_idx.cpu().long(), raux['assign_idx'])
    # gradients.  Two bf16-storage computations of the same network decorrelate to the bf16 noise floor
    # (a rounding flips whenever fp32 summation order differs), so the HIP gradients are judged against
    # the fp32 oracle RELATIVE to the error the bf16-emulating oracle itself makes against fp32.
    _, g32, _ = O.train_step(sd, img, gtb, gtl, ig, emulate_bf16=False, **kw)
    named = dict(model.named_parameters())
    keys = [str(k) for k in d['grad_keys']]
    bad = []
    for k in keys:
        e_hip, e_emu = rel_l2(named[k].grad.cpu(), g32[k]), rel_l2(og[k], g32[k])
        if float(g32[k].norm()) > 0 and e_hip > 1.6 * e_emu + 5e-3:
            bad.append((k, e_hip, e_emu))
    assert not bad, bad[:10]
    ref_norms = dict(zip(keys, d['grad_norms']))       # |grad| recorded from the reference itself
    nerr = {k: abs(float(named[k].grad.norm()) - ref_norms[k]) / (ref_norms[k] + 1e-12) for k in keys}
    print('worst grad-norm err vs fp32 reference:', sorted(nerr.items(), key=lambda kv: -kv[1])[:5])
    assert max(nerr.values()) < 0.2


def test_losses_within_1e3_of_fp32_oracle_256x320():
    """north_star bar: losses within 1e-3 relative of the reference's fp32 CPU arithmetic."""
    from oracle import fcos_oracle as O
    model = build()
    rng = np.random.RandomState(1)
    g = torch.Generator().manual_seed(3)
    H, W, B = 256, 320, 2
    img = torch.randn(B, 3, H, W, generator=g) * 40
    gtb = [T(O.synth_boxes(rng, 4, H=H, W=W, lo=8, hi=min(H, W))) for _ in range(B)]
    gtl = [T(rng.randint(0, 80, len(b)).astype('int64')) for b in gtb]
    losses = model.forward_train(img.cuda(), [dict()] * B, gtb, gtl)
    torch.cuda.synchronize()
    l32, _, aux = O.train_step(O.synth_state_dict(0), img, gtb, gtl, None, emulate_bf16=False, want_grads=False)
    for k, v in losses.items():
        assert float(v.detach()) == pytest.approx(l32[k], rel=1e-3), (k, float(v.detach()), l32[k])
    plan = next(iter(model._engine.plans.values()))
    _, raux = O.fcos_loss(aux['cls'], aux['reg'], aux['ctr'], gtb, gtl, None, return_aux=True)
    assert torch.equal(plan.lossplan.assign_idx.cpu().long(), raux['assign_idx'])      # identical indices


@pytest.mark.parametrize('B,H,W', [(1, 128, 192), (3, 160, 224), (4, 96, 160), (2, 224, 160)])
def test_train_step_vs_oracle_other_batch_sizes_and_shapes(B, H, W):
    """Batch sizes other than the benchmark's 2 (1: no image-split chains; 3 and 4: uneven / even halves) and image sizes whose level
    grids leave ragged tiles: losses within 3e-3 of the bf16-emulating oracle, identical assignment, every parameter gradient no
    farther from the fp32 gradient than 1.6 x the bf16-emulating oracle's own distance (the golden test's noise model)."""
    from oracle import fcos_oracle as O
    model = build()
    rng = np.random.RandomState(10 + B)
    g = torch.Generator().manual_seed(7 + B)
    img = (torch.randn(B, 3, H, W, generator=g) * 40).bfloat16().float()
    gtb = [T(O.synth_boxes(rng, 3, H=H, W=W, lo=8, hi=min(H, W))) for _ in range(B)]
    gtl = [T(rng.randint(0, 80, len(b)).astype('int64')) for b in gtb]
    losses = model.forward_train(img.cuda(), [dict()] * B, gtb, gtl)
    sum(losses.values()).backward()
    torch.cuda.synchronize()
    got = {k: float(v.detach()) for k, v in losses.items()}
    sd = O.synth_state_dict(0)
    ol, og, aux = O.train_step(sd, img, gtb, gtl, None, emulate_bf16=True)
    for k in got:
        assert got[k] == pytest.approx(ol[k], rel=3e-3), (k, got[k], ol[k])
    plan = next(iter(model._engine.plans.values()))
    _, raux = O.fcos_loss(aux['cls'], aux['reg'], aux['ctr'], gtb, gtl, None, return_aux=True)
    assert torch.equal(plan.lossplan.labels.cpu(), raux['labels'])
    assert torch.equal(plan.lossplan.assign_idx.cpu().long(), raux['assign_idx'])
    _, g32, _ = O.train_step(sd, img, gtb, gtl, None, emulate_bf16=False)
    named = dict(model.named_parameters())
    bad = []
    for k, gref in g32.items():
        if k not in named or named[k].grad is None or float(gref.norm()) == 0:
            continue
        e_hip, e_emu = rel_l2(named[k].grad.cpu(), gref), rel_l2(og[k], gref)
        if e_hip > 1.6 * e_emu + 5e-3:
            bad.append((k, float(e_hip), float(e_emu)))
    assert not bad, bad[:10]


def test_sgd_step_and_ema_on_flat_store():
    from dsl_amd.optim import FlatSGD
    from oracle import fcos_oracle as O
    model = build()
    rng = np.random.RandomState(0)
    g = torch.Generator().manual_seed(0)
    img = torch.randn(2, 3, 64, 96, generator=g) * 40
    gtb = [T(O.synth_boxes(rng, 2, H=64, W=96, lo=8, hi=60)) for _ in range(2)]
    gtl = [T(rng.randint(0, 80, len(b)).astype('int64')) for b in gtb]
    metas = [dict(img_shape=(64, 96, 3))] * 2
    opt = FlatSGD(model, lr=0.01, momentum=0.9, weight_decay=1e-4, paramwise_cfg=dict(bias_lr_mult=2., bias_decay_mult=0.),
                  grad_clip=dict(max_norm=35, norm_type=2))
    before = {k: v.clone().cpu() for k, v in model.state_dict().items()}
    losses = model.forward_train(img.cuda(), metas, gtb, gtl)
    sum(losses.values()).backward()
    grads = {k: p.grad.clone().cpu() for k, p in model.named_parameters() if p.grad is not None}
    opt.step()
    torch.cuda.synchronize()
    after = {k: v.cpu() for k, v in model.state_dict().items()}
    tk = O.trainable_keys(before)
    params, bufs = O.sgd_step({k: before[k] for k in tk}, {k: grads[k] for k in tk}, {}, base_lr=0.01, max_norm=35,
                              first_step=True)
    for k in tk:
        assert torch.allclose(after[k], params[k], rtol=1e-5, atol=1e-7), k
    for k in before:
        if k not in tk:
            assert torch.equal(after[k], before[k]), k      # frozen tensors untouched
    # bf16 pack follows the master weights
    st = model.store
    assert torch.equal(st.train16.float().cpu(), st.train.cpu().bfloat16().float())


def test_eager_backward_same_gradients():
    """FCOS.eager_backward only moves the launch of the backward lists behind the loss kernel: same losses and the
    same gradients as loss.backward() (up to the float-atomic reduction order inside the GN / loss kernels)."""
    from oracle import fcos_oracle as O
    torch.manual_seed(3)
    img = torch.randn(2, 3, 128, 192).bfloat16().float()
    gtb = [torch.tensor([[10., 12., 90., 100.], [40., 30., 150., 120.]]), torch.tensor([[5., 5., 60., 70.]])]
    gtl = [torch.tensor([3, 17]), torch.tensor([60])]
    metas = [dict(img_shape=(128, 192, 3), pad_shape=(128, 192, 3), scale_factor=1.0)] * 2
    grads, losses = [], []
    for eager in (False, True):
        model = build()
        model.eager_backward = eager
        out = model.train_step(dict(img=img.cuda(), img_metas=metas, gt_bboxes=gtb, gt_labels=gtl), None)
        out['loss'].backward()
        torch.cuda.synchronize()
        grads.append(model.store.grad.clone())
        losses.append(float(out['loss']))
    assert abs(losses[0] - losses[1]) < 1e-4 * abs(losses[0])
    assert torch.isfinite(grads[0]).all() and float(grads[0].abs().max()) > 0
    assert rel_l2(grads[1], grads[0]) < 1e-2


def test_full_size_losses_and_assignment_vs_oracle():
    """BASELINE.json configs[1] at its real size (2 x 3 x 800 x 1344, the bench.py workload): the three losses within
    1e-3 relative of the fp32 CPU oracle, and bit-identical target assignment on all 44 800 locations."""
    import sys
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from oracle import fcos_oracle as O
    model = build()
    b = bench.synth_batch(0, 2)
    losses = model.forward_train(b['img'], b['img_metas'], b['gt_bboxes'], b['gt_labels'])
    torch.cuda.synchronize()
    l32, _, aux = O.train_step(O.synth_state_dict(0), b['img'].cpu(), b['gt_bboxes'], b['gt_labels'], None,
                               emulate_bf16=False, want_grads=False)
    for k, v in losses.items():
        assert float(v.detach()) == pytest.approx(l32[k], rel=1e-3), (k, float(v.detach()), l32[k])
    plan = next(iter(model._engine.plans.values()))
    with torch.no_grad():
        _, raux = O.fcos_loss([t.detach() for t in aux['cls']], [t.detach() for t in aux['reg']],
                              [t.detach() for t in aux['ctr']], b['gt_bboxes'], b['gt_labels'], None, return_aux=True)
    assert plan.lossplan.assign_idx.numel() == 2 * 22400
    assert torch.equal(plan.lossplan.assign_idx.cpu().long(), raux['assign_idx'])
    assert torch.equal(plan.lossplan.labels.cpu(), raux['labels'])


def test_full_size_dsl_iteration_vs_oracle():
    """BASELINE.json configs[2] at its real size: the semi-supervised batch 3 x (3, 800, 1344) - labeled image, unlabeled
    image with ignore boxes, its half-scale copy built by append_half_scale - with loss_weight 3, the sisoft term at full
    weight: the four losses within 1e-3 relative of the fp32 CPU oracle and bit-identical assignment (gt and ignore
    passes) on all 67 200 locations."""
    import sys
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from dsl_amd.runner import append_half_scale
    from oracle import fcos_oracle as O
    kw = dict(loss_weight=3.0, soft_weight=1.0)
    model = build(soft_warm_up=0, **kw)
    model.bbox_head.cur_iter = 1                      # past the warm-up window
    b = bench.synth_batch(0, 2)
    rng = np.random.RandomState(77)
    ig = [torch.zeros(0, 4), T(bench.synth_boxes(rng, 3))]        # K ~ Poisson(3) ignore boxes on the unlabeled image
    # the three-image batch is built once, on the CPU, by the oracle's restatement of semi_epoch_based_runner.py:186-204
    # (the device version is compared with it in tests/test_boundary_cpu.py) and handed to both sides
    oimg, ogb, ogl, ogi = O.append_half_scale(b['img'].cpu(), b['gt_bboxes'], b['gt_labels'], ig)
    _, _, _, _, metas = append_half_scale(b['img'][:, :, :8, :8].cpu(), b['gt_bboxes'], b['gt_labels'], ig, b['img_metas'])
    assert oimg.shape[0] == 3 and len(metas) == 3
    losses = model.forward_train(oimg.cuda(), metas, ogb, ogl, ogi)
    torch.cuda.synchronize()
    assert set(losses) == {'loss_cls', 'loss_bbox', 'loss_centerness', 'loss_sisoft'}
    l32, _, aux = O.train_step(O.synth_state_dict(0), oimg, ogb, ogl, ogi, emulate_bf16=False, want_grads=False,
                               soft_scale=1.0, **kw)
    for k, v in losses.items():
        assert float(v.detach()) == pytest.approx(l32[k], rel=1e-3), (k, float(v.detach()), l32[k])
    plan = [p for p in model._engine.plans.values() if p.N == 3][0]
    with torch.no_grad():
        _, raux = O.fcos_loss([t.detach() for t in aux['cls']], [t.detach() for t in aux['reg']],
                              [t.detach() for t in aux['ctr']], ogb, ogl, ogi, return_aux=True, soft_scale=1.0, **kw)
    assert plan.lossplan.assign_idx.numel() == 3 * 22400
    assert torch.equal(plan.lossplan.assign_idx.cpu().long(), raux['assign_idx'])
    assert torch.equal(plan.lossplan.labels.cpu(), raux['labels'])
    if 'cls_weight' in raux:
        assert torch.equal(plan.lossplan.cls_weight.cpu(), raux['cls_weight'].float())


_FULL_ORACLE = {}


def _full_size_oracle(kind):
    """fp32 and bf16-emulating oracle gradients of the benchmark's own batch (kind 'sup': bench.synth_batch(0, 2); 'dsl': the N = 3
    semi-supervised batch of test_full_size_dsl_iteration_vs_oracle), computed once per session (about 12 s per backward pass on the
    GPU box's host) and shared by the schedule legs below."""
    if kind in _FULL_ORACLE:
        return _FULL_ORACLE[kind]
    import sys
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from oracle import fcos_oracle as O
    b = bench.synth_batch(0, 2)
    sd = O.synth_state_dict(0)
    if kind == 'sup':
        img, gtb, gtl, ig, metas, kw = b['img'].cpu(), b['gt_bboxes'], b['gt_labels'], None, b['img_metas'], {}
    else:
        from dsl_amd.runner import append_half_scale
        rng = np.random.RandomState(77)
        ig0 = [torch.zeros(0, 4), T(bench.synth_boxes(rng, 3))]
        img, gtb, gtl, ig = O.append_half_scale(b['img'].cpu(), b['gt_bboxes'], b['gt_labels'], ig0)
        _, _, _, _, metas = append_half_scale(b['img'][:, :, :8, :8].cpu(), b['gt_bboxes'], b['gt_labels'], ig0, b['img_metas'])
        kw = dict(loss_weight=3.0, soft_weight=1.0, soft_scale=1.0)
    l32, g32, _ = O.train_step(sd, img, gtb, gtl, ig, emulate_bf16=False, **kw)
    lem, gem, _ = O.train_step(sd, img, gtb, gtl, ig, emulate_bf16=True, **kw)
    _FULL_ORACLE[kind] = dict(img=img, gtb=gtb, gtl=gtl, ig=ig, metas=metas, l32=l32, g32=g32, lem=lem, gem=gem)
    return _FULL_ORACLE[kind]


_FULL_LEGS = [('sup', {}), ('sup', dict(side='0')), ('dsl', dict(tower_slots='128'))]


@pytest.mark.parametrize('kind,knobs', _FULL_LEGS, ids=[k + ''.join(f'-{a}={v}' for a, v in kn.items()) for k, kn in _FULL_LEGS])
def test_full_size_gradients_vs_oracle(monkeypatch, kind, knobs):
    """VERDICT round 5, "parity first" item 1: every whole-network gradient check ran at <= 256 x 320, where the launch planner picks
    other tiles, split factors and weight-gradient schedules than at the benchmark's 2 x 3 x 800 x 1344.  Here the schedule that SHIPS
    - FPN multi launch, layer3 / layer4 persistent `wgrad_pipe_multi_sched` tables, the x8 tower group on its 72-workgroup budget,
    image-split chains, the fused layer2 blocks, the pipelined prefix off (one step) - is compared parameter by parameter with the
    oracle's autograd backward (mmdet/models/detectors/base.py:210-243 + loss.backward()) under the golden tests' noise model:
    the HIP gradient may be no farther from the fp32 gradient than 1.6 x the bf16-emulating oracle's own distance + 5e-3.  Legs:
    the default schedule, everything on the caller's stream (side=0: the schedule in which two ordering bugs hid in rounds 4 and 5),
    and the N = 3 semi-supervised batch (ignore boxes, loss_weight 3, sisoft) with the towers' group on 128 workgroups."""
    from dsl_amd import tuning
    tuning.tune('side')
    for k_, v_ in knobs.items():
        monkeypatch.setitem(tuning._values, k_, v_)
    o = _full_size_oracle(kind)
    head = dict(loss_weight=3.0, soft_weight=1.0, soft_warm_up=0) if kind == 'dsl' else {}
    model = build(**head)
    if kind == 'dsl':
        model.bbox_head.cur_iter = 1
    losses = model.forward_train(o['img'].cuda(), o['metas'], o['gtb'], o['gtl'], o['ig'])
    sum(losses.values()).backward()
    torch.cuda.synchronize()
    plan = [p for p in model._engine.plans.values() if p.training][0]
    assert (plan.N, plan.H, plan.W) == (3 if kind == 'dsl' else 2, 800, 1344)
    assert bool(plan.BR) == (knobs.get('side', '1') == '1')
    for k, v in losses.items():
        tol = 3e-3 if k == 'loss_sisoft' else 1e-3          # (sisoft: DESIGN section 4, bf16 storage adds 2 s^2 to every squared difference)
        assert float(v.detach()) == pytest.approx(o['l32'][k], rel=tol), (k, float(v.detach()), o['l32'][k])
    named = dict(model.named_parameters())
    bad, worst, n = [], [], 0
    for k, gref in o['g32'].items():
        if k not in named or named[k].grad is None or float(gref.norm()) == 0:
            continue
        n += 1
        e_hip, e_emu = rel_l2(named[k].grad.cpu(), gref), rel_l2(o['gem'][k], gref)
        worst.append((e_hip / (1.6 * e_emu + 5e-3), k, e_hip, e_emu))
        if e_hip > 1.6 * e_emu + 5e-3:
            bad.append((k, e_hip, e_emu))
    worst.sort(reverse=True)
    print(kind, knobs, 'parameters compared:', n, 'worst err_hip / bound:', [(round(r, 3), k, round(a, 4), round(b, 4)) for r, k, a, b in worst[:6]])
    assert n >= 100, n            # every trainable tensor: 42 backbone convolutions (layer2-4), 16 FPN, 38 head, 5 scales = 101 (one may have a zero reference)
    assert not bad, bad[:10]
    # the whole gradient vector (what clipping and the all-reduce see): norm within the emulated run's own deviation + 2 %
    keys = [k for k in o['g32'] if k in named and named[k].grad is not None]
    nh = float(torch.sqrt(sum(named[k].grad.double().pow(2).sum() for k in keys)).cpu())
    n32 = float(torch.sqrt(sum(o['g32'][k].double().pow(2).sum() for k in keys)))
    nem = float(torch.sqrt(sum(o['gem'][k].double().pow(2).sum() for k in keys)))
    assert abs(nh - n32) <= 1.6 * abs(nem - n32) + 2e-2 * n32, (nh, n32, nem)


def test_plan_cache_is_bounded_and_reuses_shapes():
    """Multi-scale training visits many padded shapes: plans are cached per shape, least recently used evicted."""
    model = build()
    eng = model._get_engine()
    eng.MAX_PLANS = 3
    shapes = [(64, 96), (96, 128), (64, 128), (128, 128), (64, 96)]
    gtb = [torch.tensor([[4., 6., 40., 50.]])]
    gtl = [torch.tensor([7])]
    vals = {}
    for h, w in shapes:
        img = torch.full((1, 3, h, w), 0.5).cuda()
        losses = model.forward_train(img, [dict(img_shape=(h, w, 3))], gtb, gtl)
        v = float(sum(losses.values()))
        assert np.isfinite(v)
        assert vals.setdefault((h, w), v) == pytest.approx(v, rel=1e-4)      # the re-built (64, 96) plan gives the same loss
        assert len(eng.plans) <= 3
    shapes_cached = {k[-4:] for k in eng.plans}      # key = (store id, store generation, N, H, W, training)
    assert (1, 64, 96, True) in shapes_cached and (1, 96, 128, True) not in shapes_cached


def test_training_step_is_bit_reproducible(golden):
    """Same weights, same batch, twice (and on a second model instance): losses, every gradient element and the clipped SGD
    update are bit-identical - GroupNorm, loss, weight- and bias-gradient and gradient-norm sums all run in a fixed order."""
    from dsl_amd.optim import FlatSGD
    d = golden('net_small_dsl.npz')
    B = int(d['B'])
    img = T(d['img']).cuda()
    gtb, gtl, ig = [T(d[f'gt{i}']) for i in range(B)], [T(d[f'gl{i}']) for i in range(B)], [T(d[f'ig{i}']) for i in range(B)]
    metas = [dict(img_shape=tuple(img.shape[2:]) + (3,), pad_shape=tuple(img.shape[2:]) + (3,), scale_factor=1.0)] * B
    runs = []
    for inst in range(2):
        model = build(loss_weight=3.0, soft_weight=1.0, soft_warm_up=0)
        model.bbox_head.cur_iter = 1
        opt = FlatSGD(model, lr=0.01, momentum=0.9, weight_decay=1e-4, paramwise_cfg=dict(bias_lr_mult=2., bias_decay_mult=0.),
                      grad_clip=dict(max_norm=1.0, norm_type=2))            # small enough that the clip is active
        for rep in range(2 if inst == 0 else 1):
            losses = model.forward_train(img, metas, gtb, gtl, ig)
            sum(losses.values()).backward()
            torch.cuda.synchronize()
            runs.append(({k: float(v.detach()) for k, v in losses.items()}, model.store.grad.clone()))
        opt.step()
        torch.cuda.synchronize()
        runs[-1] += (model.store.train.clone(), float(opt.gnorm_sq))
    (l0, g0), (l1, g1, w1, n1), (l2, g2, w2, n2) = runs
    assert l0 == l1 == l2
    assert torch.equal(g0, g1) and torch.equal(g1, g2)
    assert n1 == n2 and n1 > 1.0 and torch.equal(w1, w2)


def test_pipelined_prefix_equals_inline_forward(golden):
    """FCOS.pipeline_prefix (frozen stem + layer1 of step i+1 on its own stream under the tail of step i's backward, layer1's
    output double-buffered) trains to bit-identical weights as the single-stream order, also when the images change per step."""
    from dsl_amd.optim import FlatSGD
    d = golden('net_tiny.npz')
    B = int(d['B'])
    gtb, gtl = [T(d[f'gt{i}']) for i in range(B)], [T(d[f'gl{i}']) for i in range(B)]
    g = torch.Generator().manual_seed(3)
    imgs = [(T(d['img']) + 0.5 * k * torch.randn(T(d['img']).shape, generator=g)).cuda() for k in range(4)]
    metas = [dict(img_shape=tuple(imgs[0].shape[2:]) + (3,), pad_shape=tuple(imgs[0].shape[2:]) + (3,), scale_factor=1.0)] * B
    finals = []
    for pipe in (False, True):
        model = build()
        model.eager_backward, model.pipeline_prefix, model.lazy_log = True, pipe, True
        opt = FlatSGD(model, lr=0.01, momentum=0.9, weight_decay=1e-4, paramwise_cfg=dict(bias_lr_mult=2., bias_decay_mult=0.))
        losses = []
        for img in imgs:
            out = model.train_step(dict(img=img, img_metas=metas, gt_bboxes=gtb, gt_labels=gtl), opt)
            out['loss'].backward()
            opt.step()
            losses.append(out['loss'].detach().clone())
        torch.cuda.synchronize()
        plan = [p for p in model._engine.plans.values() if p.training][0]
        assert (plan.prefix is not None) and (plan._parity == (0 if not pipe else len(imgs) % 2))
        finals.append((torch.stack(losses).cpu(), model.store.train.clone().cpu()))
    assert torch.equal(finals[0][0], finals[1][0]) and torch.equal(finals[0][1], finals[1][1])


def test_fused_bottleneck_stages_train_to_the_same_bits(monkeypatch):
    """Tuning key bneck_fwd: layer2 / layer3's bottlenecks as ONE launch each (dsl_bottleneck_fwd, csrc/bneck.hip) instead of three launches
    per block and image-split chain.  Same K order per MFMA chain and the same rounding points => at the benchmark's own size (2 x 3 x 800 x
    1344: none of these stages' launches is split-K there, which would change the fp32 summation order) the step's loss and every one of the
    32 M gradient elements are bit-identical whichever stages are fused; a stage's first block runs the stride-2 conv1 + separate-identity
    form of the kernel."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from dsl_amd import tuning
    from dsl_amd.optim import FlatSGD
    b = bench.synth_batch(0, 2)
    tuning.tune('side')                                    # (DSL_TUNE parsed before the overrides below)
    finals = []
    for fused in ('', '2', '23'):
        monkeypatch.setitem(tuning._values, 'bneck_fwd', fused)
        model = build()
        opt = FlatSGD(model, lr=0.01, momentum=0.9, weight_decay=1e-4, paramwise_cfg=dict(bias_lr_mult=2., bias_decay_mult=0.))
        out = model.train_step(b, opt)
        out['loss'].backward()
        torch.cuda.synchronize()
        plan = [p for p in model._engine.plans.values() if p.training][0]
        n_fused = sum(1 for o in plan.fwd.items if o.kind == L_OP_BNECK())
        assert n_fused == {'': 0, '2': 4, '23': 10}[fused], n_fused
        finals.append((out['loss'].detach().clone().cpu(), model.store.grad.clone().cpu()))
        del model, opt, out, plan
    for other in finals[1:]:
        assert torch.equal(finals[0][0], other[0]), (finals[0][0], other[0])
        assert torch.equal(finals[0][1], other[1]), float((finals[0][1] - other[1]).abs().max())


def L_OP_BNECK():
    from dsl_amd import _lib as L
    return L.OP_BNECK


def test_deferred_head_update_trains_to_the_same_bits(golden, monkeypatch):
    """Deferred head update (engine.Plan.defer, FlatSGD without clipping): the towers' weight gradients and the head + FPN bucket's
    optimizer step run under the NEXT step's backbone forward, which waits for them in front of the FPN (SLOT_HEADW).  Same kernels
    and summation order (engine.DEFER_SLOTS = the inline budget) => bit-identical losses and weights over steps with changing images;
    a state_dict() read right behind opt.step() already sees the finished update (ParamStore.wait_pending)."""
    from dsl_amd.optim import FlatSGD
    d = golden('net_tiny.npz')
    B = int(d['B'])
    gtb, gtl = [T(d[f'gt{i}']) for i in range(B)], [T(d[f'gl{i}']) for i in range(B)]
    g = torch.Generator().manual_seed(5)
    imgs = [(T(d['img']) + 0.5 * k * torch.randn(T(d['img']).shape, generator=g)).cuda() for k in range(5)]
    metas = [dict(img_shape=tuple(imgs[0].shape[2:]) + (3,), pad_shape=tuple(imgs[0].shape[2:]) + (3,), scale_factor=1.0)] * B
    from dsl_amd import tuning
    tuning.tune('side')                                    # (DSL_TUNE parsed before the overrides below)
    from dsl_amd import engine
    monkeypatch.setattr(engine, 'DEFER_SLOTS', 72)
    monkeypatch.setitem(tuning._values, 'tower_slots', '72')
    finals = []
    for defer in ('0', '1'):
        model = build()
        opt = FlatSGD(model, lr=0.01, momentum=0.9, weight_decay=1e-4, paramwise_cfg=dict(bias_lr_mult=2., bias_decay_mult=0.),
                      defer_head_update=defer == '1')
        assert bool(getattr(model.store, 'defer_head', False)) == (defer == '1')
        losses = []
        for img in imgs:
            out = model.train_step(dict(img=img, img_metas=metas, gt_bboxes=gtb, gt_labels=gtl), opt)
            out['loss'].backward()
            opt.step()
            losses.append(out['loss'].detach().clone())
        early = {k: v.clone() for k, v in model.state_dict().items()}          # no device-wide sync in front of this read
        torch.cuda.synchronize()
        late = model.state_dict()
        assert all(torch.equal(early[k], late[k]) for k in late)
        plan = [p for p in model._engine.plans.values() if p.training][0]
        assert plan.defer == (defer == '1')
        finals.append((torch.stack(losses).cpu(), model.store.train.clone().cpu(), model.store.train16.clone().cpu()))
    assert torch.equal(finals[0][0], finals[1][0])
    assert torch.equal(finals[0][1], finals[1][1]) and torch.equal(finals[0][2], finals[1][2])


def test_deferred_plan_with_a_clipping_optimizer_still_sees_every_gradient(golden, monkeypatch):
    """mmcv's OptimizerHook hands grad_clip to the optimizer AFTER the first backward pass: a list built for the deferred update
    (its last weight gradients are not joined into the caller's stream) then meets the whole-buffer clipped step, which must wait
    for them; from the next step on the lists are built without the deferral."""
    from dsl_amd.optim import FlatSGD
    d = golden('net_tiny.npz')
    B = int(d['B'])
    gtb, gtl = [T(d[f'gt{i}']) for i in range(B)], [T(d[f'gl{i}']) for i in range(B)]
    img = T(d['img']).cuda()
    metas = [dict(img_shape=tuple(img.shape[2:]) + (3,), pad_shape=tuple(img.shape[2:]) + (3,), scale_factor=1.0)] * B
    from dsl_amd import tuning
    tuning.tune('side')
    from dsl_amd import engine
    monkeypatch.setattr(engine, 'DEFER_SLOTS', 72)
    res = []
    for late_clip in (False, True):
        model = build()
        opt = FlatSGD(model, lr=0.01, momentum=0.9, weight_decay=1e-4, paramwise_cfg=dict(bias_lr_mult=2., bias_decay_mult=0.),
                      grad_clip=None if late_clip else dict(max_norm=1.0, norm_type=2), defer_head_update=True)
        for it in range(2):
            out = model.train_step(dict(img=img, img_metas=metas, gt_bboxes=gtb, gt_labels=gtl), opt)
            if late_clip and it == 0:
                assert model.store.defer_head
                opt.max_norm = 1.0                 # what OptimizerHook.after_train_iter does in front of its first step
            out['loss'].backward()
            opt.step()
        assert not model.store.defer_head
        torch.cuda.synchronize()
        res.append(model.store.train.clone().cpu())
    assert torch.equal(res[0], res[1])


def test_first_optimizer_step_keeps_every_buckets_momentum():
    """Round 4: the momentum buffer is created (zero-filled on the caller's stream) inside the first `step()`, while the per-bucket
    updates run on other streams as soon as their bucket's weight gradients are done - a bucket updated BEFORE the fill landed had its
    momentum wiped and started its second step from zero.  After one step without clipping, every bucket's momentum must be the
    first-step rule's `grad + wd * param` (torch.optim.SGD: buf = d_p), in particular non-zero wherever the gradient is."""
    from dsl_amd.optim import FlatSGD
    from oracle import fcos_oracle as O
    rng = np.random.RandomState(11)
    g = torch.Generator().manual_seed(11)
    H, W, B = 256, 320, 2
    img = (torch.randn(B, 3, H, W, generator=g) * 40).cuda()
    gtb = [T(O.synth_boxes(rng, 4, H=H, W=W, lo=8, hi=200)) for _ in range(B)]
    gtl = [T(rng.randint(0, 80, len(b)).astype('int64')) for b in gtb]
    metas = [dict(img_shape=(H, W, 3), pad_shape=(H, W, 3), scale_factor=1.0)] * B
    for rep in range(3):                      # (a race: a few fresh optimizers, the GPU kept busy in front of each first step)
        model = build()
        opt = FlatSGD(model, lr=0.01, momentum=0.9, weight_decay=1e-4, paramwise_cfg=dict(bias_lr_mult=2., bias_decay_mult=0.))
        p0 = model.store.train.clone()
        for _ in range(3):                    # queue work ahead of the step under test so that the GPU lags behind the host
            model.forward_train(img, metas, gtb, gtl)
        out = model.train_step(dict(img=img, img_metas=metas, gt_bboxes=gtb, gt_labels=gtl), opt)
        out['loss'].backward()
        opt.step()
        torch.cuda.synchronize()
        st = model.store
        grad, mom = st.grad, opt.momentum_buf
        wd = torch.where(st.group.bool(), torch.zeros_like(p0), torch.full_like(p0, 1e-4))
        want = grad + wd * p0
        err = (mom - want).abs().max()
        assert float(err) <= 1e-6 * float(want.abs().max()) + 1e-12, (rep, float(err))
        for lo, hi in st.grad_buckets():
            assert float(mom[lo:hi].abs().sum()) > 0, (rep, lo, hi)
