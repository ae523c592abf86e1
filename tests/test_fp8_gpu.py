"""fp8 forward path (BASELINE.json configs[4], first slice): quantisation kernels and the MX-scaled fp8 convolution against
torch's own float8_e4m3fn arithmetic, then the FCOS step with fp8 tower convolutions against the bf16 step.  The reference has no
such path (it trains in fp32): the bars are this build's own - stated here and in DESIGN.md."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from util import fcos_model_cfg

pytestmark = pytest.mark.gpu
T = torch.from_numpy
F8 = torch.float8_e4m3fn


@pytest.fixture(scope='module')
def K():
    from dsl_amd import _lib as L
    from dsl_amd import ops
    return L, ops


def test_quant_fp8_matches_torch_cast(K):
    L, _ = K
    g = torch.Generator().manual_seed(1)
    rows, c, ld = 1000, 256, 384
    x = (torch.randn(rows, ld, generator=g) * 3).bfloat16()
    x[0, :8] = torch.tensor([1000., -1000., 448., -448., 0., 1e-4, 0.0625, 300.]).bfloat16()     # saturation, zero, subnormal range
    xd = x.cuda()
    y = torch.zeros(rows, c, dtype=torch.uint8, device='cuda')
    L.check(L.lib.dsl_quant_fp8(L.ptr(xd), L.ptr(y), rows, c, ld, 16.0, L.stream_ptr()))
    ref = (x[:, :c].float() * 16.0).clamp(-448, 448).to(F8).view(torch.uint8)
    assert torch.equal(y.cpu(), ref)


def test_quant_fp8_weights(K):
    L, _ = K
    g = torch.Generator().manual_seed(2)
    co, cop, k = 80, 128, 2304
    w = torch.randn(co, k, generator=g) * 0.05
    w[3] = 0.0
    bn = torch.rand(co, generator=g) + 0.5
    wd, bnd = w.cuda(), bn.cuda()
    w8 = torch.full((cop, k), 7, dtype=torch.uint8, device='cuda')
    comb = torch.full((cop,), float('nan'), device='cuda')
    L.check(L.lib.dsl_quant_fp8_weights(L.ptr(wd), L.ptr(w8), L.ptr(comb), L.ptr(bnd), co, cop, k, 1.0 / 16.0, L.stream_ptr()))
    amax = w.abs().amax(1)
    s = torch.where(amax > 0, 448.0 / amax, torch.ones_like(amax))
    ref = (w * s[:, None]).clamp(-448, 448).to(F8).view(torch.uint8)
    assert torch.equal(w8[:co].cpu(), ref) and int(w8[co:].max()) == 0
    assert torch.allclose(comb[:co].cpu(), (1.0 / 16.0) / s * bn, rtol=1e-6) and float(comb[co:].abs().max()) == 0.0


@pytest.mark.parametrize('shape', [(2, 256, 256, 3, [(20, 28), (10, 14)]), (1, 128, 128, 1, [(17, 23)]), (2, 256, 80, 3, [(9, 11)])])
def test_conv_fp8_vs_torch_on_the_same_quantised_operands(K, shape):
    """The kernel multiplies exactly the e4m3 values torch dequantises: products are exact in fp32, sums differ by order only."""
    L, ops = K
    n, ci, co, k, lv = shape
    cop = (co + 127) // 128 * 128
    g = torch.Generator().manual_seed(5)
    P = sum(h * w for h, w in lv) * n
    x = torch.relu(torch.randn(P, ci, generator=g)).bfloat16()
    w = torch.randn(co, k, k, ci, generator=g) * 0.05
    sx = 16.0
    xd, wd = x.cuda(), w.reshape(co, -1).contiguous().cuda()
    x8 = torch.zeros(P, ci, dtype=torch.uint8, device='cuda')
    w8 = torch.zeros(cop, k * k * ci, dtype=torch.uint8, device='cuda')
    comb = torch.zeros(cop, device='cuda')
    bias = torch.randn(co, generator=g).cuda()
    L.check(L.lib.dsl_quant_fp8(L.ptr(xd), L.ptr(x8), P, ci, ci, sx, L.stream_ptr()))
    L.check(L.lib.dsl_quant_fp8_weights(L.ptr(wd), L.ptr(w8), L.ptr(comb), None, co, cop, k * k * ci, 1.0 / sx, L.stream_ptr()))
    y = torch.zeros(P, co, device='cuda')
    d = ops.conv_desc(x8, w8, y, n=n, grid=lv, src_hw=lv, dst_hw=lv, cs=ci, cd=co, cd_pad=cop, ldd=co, kh=k, kw=k, stride=1, pad=k // 2,
                      flags=L.CONV_FP8 | L.CONV_OUT_F32, scale=comb, bias=bias)
    L.check(L.lib.dsl_conv2d(C.byref(d), L.stream_ptr()), 'dsl_conv2d fp8')
    torch.cuda.synchronize()
    xq = x8.cpu().view(F8).float()
    wq = w8[:co].cpu().view(F8).float().reshape(co, k, k, ci)
    off, outs = 0, []
    for h, wd_ in lv:
        xs = xq[off:off + n * h * wd_].reshape(n, h, wd_, ci).permute(0, 3, 1, 2)
        r = F.conv2d(xs.double(), wq.permute(0, 3, 1, 2).double(), None, 1, k // 2)
        outs.append(r.permute(0, 2, 3, 1).reshape(-1, co))
        off += n * h * wd_
    ref = (torch.cat(outs) * comb[:co].cpu().double() + bias.cpu().double()).float()
    got = y.cpu()
    assert torch.allclose(got, ref, rtol=2e-5, atol=2e-5 * float(ref.abs().max())), float((got - ref).abs().max())


def test_dynamic_scale_quantisation(K):
    """dsl_absmax + dsl_quant_fp8_dyn + dsl_fp8_comb: the tensor's own maximum maps to 448, whatever its range."""
    L, _ = K
    g = torch.Generator().manual_seed(7)
    rows, c = 777, 256
    for mag in (1e-3, 1.0, 4608.0):
        x = (torch.randn(rows, c, generator=g) * mag).bfloat16()
        xd = x.cuda()
        part = torch.full((64,), float('nan'), device='cuda')
        y = torch.zeros(rows, c, dtype=torch.uint8, device='cuda')
        winv, comb = torch.rand(256, generator=g).cuda(), torch.zeros(256, device='cuda')
        L.check(L.lib.dsl_absmax(L.ptr(xd), rows, c, c, L.ptr(part), 64, L.stream_ptr()))
        L.check(L.lib.dsl_quant_fp8_dyn(L.ptr(xd), L.ptr(y), rows, c, c, L.ptr(part), 64, L.stream_ptr()))
        L.check(L.lib.dsl_fp8_comb(L.ptr(winv), L.ptr(comb), 256, L.ptr(part), 64, L.stream_ptr()))
        amax = float(x.float().abs().max())
        assert float(part.max()) == amax
        scale = np.float32(448.0) / np.float32(amax)
        ref = (x.float() * float(scale)).clamp(-448, 448).to(F8).view(torch.uint8)
        assert torch.equal(y.cpu(), ref)
        assert int((y.cpu().view(F8).float().abs() == 448).sum()) >= 1
        assert torch.allclose(comb.cpu(), winv.cpu() * (amax / 448.0), rtol=1e-6)


def test_delayed_scaling_prep_and_quantiser(K):
    """Round 6, delayed scaling: dsl_fp8_prep (weights + epilogue scales + input scales of several convolutions in one launch, from the
    block maxima the previous step left) and dsl_quant_fp8_delayed (quantise with that scale, leave this step's maxima) against torch."""
    L, _ = K
    g = torch.Generator().manual_seed(7)
    co, cop, k, n_items, margin = 250, 256, 2304, 3, 1.25
    ws = [(torch.randn(co, k, generator=g) * 0.05 * (i + 1)).cuda() for i in range(n_items)]
    ws[1][5] = 0.0
    amaxs = [torch.rand(40 + i, generator=g).cuda() * 9.0 for i in range(n_items)]
    amaxs[2].zero_()                                          # nothing recorded yet: scale 1
    w8 = [torch.full((cop, k), 7, dtype=torch.uint8, device='cuda') for _ in range(n_items)]
    comb = [torch.full((cop,), float('nan'), device='cuda') for _ in range(n_items)]
    scales = torch.full((n_items,), float('nan'), device='cuda')
    items = (L.Fp8PrepItem * n_items)()
    for i in range(n_items):
        it = items[i]
        it.w, it.w8, it.comb, it.amax, it.n_amax, it.cout = ws[i].data_ptr(), w8[i].data_ptr(), comb[i].data_ptr(), amaxs[i].data_ptr(), amaxs[i].numel(), co
        it.scale = scales.data_ptr() + 4 * i
    tab = torch.frombuffer(bytearray(bytes(items)), dtype=torch.uint8).cuda()
    L.check(L.lib.dsl_fp8_prep(L.ptr(tab), n_items, cop, k, margin, L.stream_ptr()), 'dsl_fp8_prep')
    torch.cuda.synchronize()
    for i in range(n_items):
        w = ws[i].cpu()
        wmax = w.abs().amax(1)
        s = torch.where(wmax > 0, 448.0 / wmax, torch.ones_like(wmax))
        assert torch.equal(w8[i][:co].cpu(), (w * s[:, None]).clamp(-448, 448).to(F8).view(torch.uint8)) and int(w8[i][co:].max()) == 0
        a = np.float32(float(amaxs[i].max())) * np.float32(margin)
        want_scale = float(np.float32(448.0) / a) if a > 0 else 1.0
        assert float(scales[i]) == want_scale, (i, float(scales[i]), want_scale)
        inv = float(a / np.float32(448.0)) if a > 0 else 1.0
        assert torch.allclose(comb[i][:co].cpu(), inv / s, rtol=1e-6) and float(comb[i][co:].abs().max()) == 0.0
    rows, c, ld = 777, 256, 320
    x = (torch.randn(rows, ld, generator=g) * 5).bfloat16()
    x[5, 3] = 3000.0                                          # beyond last step's maximum: saturates at 448
    xd = x.cuda()
    y = torch.zeros(rows, c, dtype=torch.uint8, device='cuda')
    part = torch.full((48,), float('nan'), device='cuda')
    L.check(L.lib.dsl_quant_fp8_delayed(L.ptr(xd), L.ptr(y), rows, c, ld, scales.data_ptr(), L.ptr(part), 48, L.stream_ptr()), 'dsl_quant_fp8_delayed')
    ref = (x[:, :c].float() * float(scales[0])).clamp(-448, 448).to(F8).view(torch.uint8)
    assert torch.equal(y.cpu(), ref)
    assert float(part.max()) == float(x[:, :c].float().abs().max()) and bool(torch.isfinite(part).all())


def test_groupnorm_apply_writes_the_fp8_copy(K):
    """dsl_gn_desc.y8: GroupNorm's apply pass writes the e4m3 copy of its (rounded) output with the given scale and leaves the block
    maxima - equal to quantising y in a pass of its own, bit for bit; y itself is unchanged by the option."""
    L, ops = K
    g = torch.Generator().manual_seed(9)
    N, sizes, Cc = 2, [(25, 36), (13, 18), (7, 9), (4, 5), (2, 3)], 256
    P = sum(h * w for h, w in sizes) * N
    x = (torch.randn(P, Cc, generator=g) * 2.0).bfloat16().cuda()
    ga, be = (1 + 0.2 * torch.randn(Cc, generator=g)).cuda(), (0.3 * torch.randn(Cc, generator=g)).cuda()
    ys = []
    nblk = (max(h * w for h, w in sizes) + 127) // 128
    for with8 in (False, True):
        y = torch.empty(P, Cc, dtype=torch.bfloat16, device='cuda')
        stats = torch.empty(5 * N * 32, 2, device='cuda')
        y8 = torch.full((P, Cc), 9, dtype=torch.uint8, device='cuda')
        sc = torch.tensor([150.0], device='cuda')
        am = torch.zeros(5 * N * nblk, device='cuda')
        gd = ops.gn_desc(x, y, ga, be, stats, n=N, hw=sizes, **(dict(y8=y8, y8_scale=sc, y8_amax=am) if with8 else {}))
        L.check(L.lib.dsl_groupnorm_relu_fwd(C.byref(gd), L.stream_ptr()), 'gn')
        torch.cuda.synchronize()
        ys.append(y.clone())
    assert torch.equal(ys[0], ys[1])
    ref = (ys[1].float().cpu() * 150.0).clamp(-448, 448).to(F8).view(torch.uint8)
    assert torch.equal(y8.cpu(), ref)
    assert float(am.max()) == float(ys[1].float().max())
    assert int((ref.view(F8).float() == 448).sum()) > 0          # (the scale is large enough that some values saturate: the clamp is exercised)


def _build(**extra):
    from dsl_amd import detectors  # noqa: F401
    from dsl_amd.registry import build_detector
    from oracle import fcos_oracle as O
    cfg = fcos_model_cfg()
    cfg.update(extra)
    model = build_detector(cfg)
    model.load_state_dict(O.synth_state_dict(0))
    return model.cuda()


def test_fcos_step_with_fp8_towers_vs_bf16_step():
    """FCOS(fp8=dict(layers='towers')) at 256x320: the three losses within 1e-2 of the all-bf16 step (the bar of this build for
    the fp8 slice: e4m3 carries 3 mantissa bits, the towers are 8 of the ~70 convolutions on the path to the losses), identical
    assignment (it does not depend on the network), finite gradients of the same scale, and the optimizer step runs."""
    from dsl_amd.optim import FlatSGD
    from oracle import fcos_oracle as O
    rng = np.random.RandomState(3)
    g = torch.Generator().manual_seed(4)
    H, W, B = 256, 320, 2
    img = (torch.randn(B, 3, H, W, generator=g) * 40).bfloat16().float().cuda()
    gtb = [T(O.synth_boxes(rng, 5, H=H, W=W, lo=8, hi=200)) for _ in range(B)]
    gtl = [T(rng.randint(0, 80, len(b)).astype('int64')) for b in gtb]
    res = {}
    for name, extra in (('bf16', {}), ('fp8', dict(fp8=dict(layers='towers')))):
        model = _build(**extra)
        opt = FlatSGD(model, lr=0.01, momentum=0.9, weight_decay=1e-4)
        losses = model.forward_train(img, [dict()] * B, gtb, gtl)
        sum(losses.values()).backward()
        torch.cuda.synchronize()
        plan = [p for p in model._engine.plans.values() if p.training][0]
        res[name] = ({k: float(v.detach()) for k, v in losses.items()}, plan.lossplan.assign_idx.clone().cpu(),
                     model.store.grad.detach().clone().cpu())
        assert ('feats.f8' in plan.bufs) == (name == 'fp8')
        opt.step()
        torch.cuda.synchronize()
        assert torch.isfinite(model.store.train).all()
    print('losses bf16', res['bf16'][0], 'fp8', res['fp8'][0])
    for k, v in res['bf16'][0].items():
        assert res['fp8'][0][k] == pytest.approx(v, rel=1e-2), (k, res['fp8'][0][k], v)
    assert torch.equal(res['bf16'][1], res['fp8'][1])
    g16, g8 = res['bf16'][2], res['fp8'][2]
    assert torch.isfinite(g8).all() and float(g8.norm()) == pytest.approx(float(g16.norm()), rel=0.1)
    cos = float((g16 * g8).sum() / (g16.norm() * g8.norm()))
    print('gradient norms', float(g16.norm()), float(g8.norm()), 'cosine', cos)
    assert cos > 0.9, cos          # (two bf16 runs of this net at different summation orders: ~0.97)


def test_full_size_fp8_towers_step_vs_fp32_oracle():
    """BASELINE.json configs[4]'s slice against the ORACLE, not against the bf16 step (round-4 review item 6): the benchmark's own batch
    (2 x 3 x 800 x 1344, 44 800 locations) through FCOS(fp8=dict(layers='towers')) - e4m3 tower forward convolutions, delayed
    per-tensor activation scales (margin 1.25 over the recorded maximum), per-channel weight scales - against the fp32 CPU restatement of the reference step on the same
    inputs.  Stated tolerance: every loss within 2e-3 relative of fp32 (5e-3 until round 5) (the all-bf16 step is held to 1e-3 in
    tests/test_step_gpu.py::test_full_size_losses_and_assignment_vs_oracle; e4m3 carries 3 mantissa bits on 8 of the ~70
    convolutions between the image and the losses; measured values are printed), and bit-identical target assignment."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from oracle import fcos_oracle as O
    model = _build(**dict(fp8=dict(layers='towers')))
    b = bench.synth_batch(0, 2)
    losses = model.forward_train(b['img'], b['img_metas'], b['gt_bboxes'], b['gt_labels'])
    torch.cuda.synchronize()
    plan = next(iter(model._engine.plans.values()))
    assert 'feats.f8' in plan.bufs                      # the fp8 path is the one that ran
    l32, _, aux = O.train_step(O.synth_state_dict(0), b['img'].cpu(), b['gt_bboxes'], b['gt_labels'], None, emulate_bf16=False, want_grads=False)
    got = {k: float(v.detach()) for k, v in losses.items()}
    print('fp8 towers', got, 'fp32 oracle', {k: l32[k] for k in got}, 'relative', {k: abs(got[k] - l32[k]) / abs(l32[k]) for k in got})
    for k, v in got.items():
        assert v == pytest.approx(l32[k], rel=2e-3), (k, v, l32[k])
    with torch.no_grad():
        _, raux = O.fcos_loss([t.detach() for t in aux['cls']], [t.detach() for t in aux['reg']], [t.detach() for t in aux['ctr']],
                              b['gt_bboxes'], b['gt_labels'], None, return_aux=True)
    assert torch.equal(plan.lossplan.assign_idx.cpu().long(), raux['assign_idx'])
    assert torch.equal(plan.lossplan.labels.cpu(), raux['labels'])


def test_delayed_scales_follow_the_previous_step():
    """Three optimizer steps with fp8 towers: from the second step on every activation scale is 448 / (1.25 x the maximum the step
    before recorded), the losses stay next to the all-bf16 run's (3e-3 on the first step, 5e-2 on the diverging trajectories after it), nothing saturates badly (the
    share of e4m3 codes at +-448 in the tower inputs stays under 1e-4)."""
    from dsl_amd.optim import FlatSGD
    from oracle import fcos_oracle as O
    rng = np.random.RandomState(5)
    g = torch.Generator().manual_seed(6)
    H, W, B = 192, 256, 2
    img = (torch.randn(B, 3, H, W, generator=g) * 40).bfloat16().float().cuda()
    gtb = [T(O.synth_boxes(rng, 4, H=H, W=W, lo=8, hi=150)) for _ in range(B)]
    gtl = [T(rng.randint(0, 80, len(b)).astype('int64')) for b in gtb]
    traj = {}
    for name, extra in (('bf16', {}), ('fp8', dict(fp8=dict(layers='towers')))):
        model = _build(**extra)
        opt = FlatSGD(model, lr=0.01, momentum=0.9, weight_decay=1e-4)
        vals = []
        for step in range(3):
            if name == 'fp8' and step > 0:
                plan = [p for p in model._engine.plans.values() if p.training][0]
                prev = {k: float(v.max()) for k, v in plan.bufs.items() if k.endswith('.amax')}
            losses = model.forward_train(img, [dict()] * B, gtb, gtl)
            sum(losses.values()).backward()
            torch.cuda.synchronize()
            vals.append({k: float(v.detach()) for k, v in losses.items()})
            if name == 'fp8':
                plan = [p for p in model._engine.plans.values() if p.training][0]
                sc = plan.bufs['fp8.scales'].cpu().numpy()
                if step > 0:
                    for t_, tower in enumerate(('cls_convs', 'reg_convs')):
                        for i in range(4):
                            a = prev['feats.f8.amax'] if i == 0 else prev[f'{tower}.{i - 1}.act8.amax']
                            assert sc[t_ * 4 + i] == np.float32(448.0) / (np.float32(a) * np.float32(1.25)), (step, tower, i)
                for k8 in ['feats.f8'] + [f'{tw}.{i}.act8' for tw in ('cls_convs', 'reg_convs') for i in range(3)]:
                    sat = float((plan.bufs[k8].view(F8).float().abs() == 448).float().mean())
                    assert sat < 1e-4, (step, k8, sat)
            opt.step()
        traj[name] = vals
    print('bf16', traj['bf16'], 'fp8', traj['fp8'])
    for step, (a, b) in enumerate(zip(traj['bf16'], traj['fp8'])):
        for k in a:      # the first step sees the same weights (measured: 1e-3); afterwards the two runs are different trajectories of a
            #              randomly initialised net at lr 0.01 (measured: up to 2 % on loss_bbox) - they must stay neighbours, no more
            assert b[k] == pytest.approx(a[k], rel=3e-3 if step == 0 else 5e-2), (step, k, a[k], b[k])
