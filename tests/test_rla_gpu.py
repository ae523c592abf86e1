"""RLA_ResNet on the GPU: the new memory-bound kernels against torch, the strided-source /
shared-weight forms of the conv kernels, and the whole FCOS + RLA_ResNet training step against the CPU oracle
(oracle/rla_oracle.py, pinned to the reference's module by tests/golden/rla_tiny.npz)."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from util import fcos_model_cfg, rel_l2

pytestmark = pytest.mark.gpu
T = torch.from_numpy
BF = torch.bfloat16


def rla_model_cfg(**head):
    cfg = fcos_model_cfg(**head)
    cfg['backbone'] = dict(type='RLA_ResNet', layers=[3, 4, 6, 3], frozen_stages=1, norm_eval=True, style='pytorch')
    return cfg


def test_rla_elementwise_kernels():
    from dsl_amd import _lib as L
    g = torch.Generator().manual_seed(3)
    N, H, W, Cc = 2, 10, 14, 32
    sp = L.stream_ptr
    # average pool of the h slice of wider rows, and its backward
    xh = torch.randn(N, H, W, 96, generator=g).to(BF).cuda()
    y = torch.zeros(N, H // 2, W // 2, 32, dtype=BF, device='cuda')
    L.check(L.lib.dsl_avgpool2x2(xh.data_ptr() + 64 * 2, 96, L.ptr(y), 32, N, H, W, Cc, sp()))
    ref = F.avg_pool2d(xh[..., 64:].float().permute(0, 3, 1, 2), 2, 2).permute(0, 2, 3, 1)
    assert torch.allclose(y.float(), ref, rtol=1e-2, atol=1e-2)
    gy = torch.randn(N, H // 2, W // 2, 64, generator=g).to(BF).cuda()
    gx = torch.full((N, H, W, 64), 7.0, dtype=BF, device='cuda')
    L.check(L.lib.dsl_avgpool2x2_bwd(L.ptr(gy), 64, L.ptr(gx), 64, N, H, W, Cc, sp()))
    refg = F.interpolate(gy[..., :32].float().permute(0, 3, 1, 2), scale_factor=2, mode='nearest').permute(0, 2, 3, 1) * 0.25
    assert torch.allclose(gx[..., :32].float(), refg, rtol=1e-2, atol=1e-3) and float((gx[..., 32:].float() - 7).abs().max()) == 0
    # t = tanh(bn(u)) and backward with (dgamma, dbeta)
    rows = N * H * W
    u = (torch.randn(rows, Cc, generator=g) * 1.5).to(BF)
    gam, bet = 1 + 0.2 * torch.randn(Cc, generator=g), 0.1 * torch.randn(Cc, generator=g)
    mu, var = 0.2 * torch.randn(Cc, generator=g), 0.5 + torch.rand(Cc, generator=g)
    u32 = u.float().requires_grad_()
    gam_r, bet_r = gam.clone().requires_grad_(), bet.clone().requires_grad_()
    t_ref = torch.tanh(F.batch_norm(u32, mu, var, gam_r, bet_r, False, 0.0, 1e-5))
    gt = torch.randn(rows, Cc, generator=g).to(BF)
    (t_ref * gt.float()).sum().backward()
    sc, bi = torch.empty(Cc, device='cuda'), torch.empty(Cc, device='cuda')
    gam_d, bet_d, mu_d, var_d, gt_d = gam.cuda(), bet.cuda(), mu.cuda(), var.cuda(), gt.cuda()      # kept alive: the library gets raw pointers
    L.check(L.lib.dsl_bn_fold(L.ptr(gam_d), L.ptr(bet_d), L.ptr(mu_d), L.ptr(var_d), 1e-5, L.ptr(sc), L.ptr(bi), Cc, sp()))
    assert torch.allclose(sc.cpu(), gam / torch.sqrt(var + 1e-5), rtol=1e-5) and torch.allclose(bi.cpu(), bet - mu * gam / torch.sqrt(var + 1e-5), rtol=1e-4, atol=1e-6)
    t_d = torch.zeros(rows, 128, dtype=BF, device='cuda')
    u_d = u.cuda()
    L.check(L.lib.dsl_bn_tanh_fwd(L.ptr(u_d), Cc, L.ptr(sc), L.ptr(bi), L.ptr(t_d), 128, rows, Cc, sp()))
    assert torch.allclose(t_d[:, :Cc].float().cpu(), t_ref.detach(), rtol=1e-2, atol=1e-2) and float(t_d[:, Cc:].float().abs().max()) == 0
    gu = torch.zeros(rows, 64, dtype=BF, device='cuda')
    dg, db = torch.full((Cc,), float('nan'), device='cuda'), torch.full((Cc,), float('nan'), device='cuda')
    ws = torch.empty(L.lib.dsl_bn_tanh_bwd_workspace_bytes(rows, Cc), dtype=torch.uint8, device='cuda')
    for _ in range(2):
        L.check(L.lib.dsl_bn_tanh_bwd(L.ptr(gt_d), Cc, L.ptr(t_d), 128, L.ptr(u_d), Cc, L.ptr(sc), L.ptr(mu_d), L.ptr(var_d), 1e-5,
                                      L.ptr(gu), 64, L.ptr(dg), L.ptr(db), L.ptr(ws), rows, Cc, sp()))
    torch.cuda.synchronize()
    assert torch.allclose(gu[:, :Cc].float().cpu(), u32.grad, rtol=3e-2, atol=3e-2 * float(u32.grad.abs().max()))
    assert torch.allclose(dg.cpu(), gam_r.grad, rtol=2e-2, atol=2e-2 * float(gam_r.grad.abs().max()))
    assert torch.allclose(db.cpu(), bet_r.grad, rtol=2e-2, atol=2e-2 * float(bet_r.grad.abs().max()))


def test_bn_gradients_from_unscaled_weight_gradient():
    """conv -> BN(eval, trainable affine): dgamma = (<W, dWu> - mean S) / sqrt(var + eps), dbeta = S, dW = dWu * gamma /
    sqrt(var + eps) (dsl_bn_wgrad_post) == autograd of F.batch_norm(F.conv2d(x, W))."""
    from dsl_amd import _lib as L
    g = torch.Generator().manual_seed(5)
    Co, Ci, k = 64, 128, 3
    x = torch.randn(2, Ci, 9, 11, generator=g)
    w = (torch.randn(Co, Ci, k, k, generator=g) * 0.05).requires_grad_()
    gam, bet = (1 + 0.2 * torch.randn(Co, generator=g)).requires_grad_(), (0.1 * torch.randn(Co, generator=g)).requires_grad_()
    gam.data[3] = 0.0                                  # zero_init_last_bn: no division by gamma anywhere
    mu, var = 0.2 * torch.randn(Co, generator=g), 0.5 + torch.rand(Co, generator=g)
    yv = F.batch_norm(F.conv2d(x, w, None, 1, 1), mu, var, gam, bet, False, 0.0, 1e-5)
    gy = torch.randn(yv.shape, generator=g)
    (yv * gy).sum().backward()
    # the unscaled weight gradient w.r.t. the raw conv output and the column sum of g_y, as the wgrad launch leaves them
    conv = F.conv2d(x, w.detach().clone().requires_grad_(), None, 1, 1)
    wu = torch.autograd.grad(conv, [p for p in [conv.grad_fn.next_functions[1][0].variable]], gy)[0] if False else None
    w2 = w.detach().clone().requires_grad_()
    (F.conv2d(x, w2, None, 1, 1) * gy).sum().backward()
    dWu = w2.grad.permute(0, 2, 3, 1).contiguous().cuda()                     # KRSC, as the flat gradient buffer holds it
    W = w.detach().permute(0, 2, 3, 1).contiguous().cuda()
    S = gy.sum((0, 2, 3)).cuda()
    dgam = torch.full((Co,), float('nan'), device='cuda')
    it = L.BnPostItem()
    keep = (gam.detach().cuda(), mu.cuda(), var.cuda())
    it.w, it.dw, it.dgamma, it.dbeta, it.gamma, it.mean, it.var = (t.data_ptr() for t in (W, dWu, dgam, S) + keep)
    it.rows, it.k, it.row_start = Co, k * k * Ci, 0
    tab = torch.frombuffer(bytearray(bytes((L.BnPostItem * 1)(it))), dtype=torch.uint8).clone().cuda()
    L.check(L.lib.dsl_bn_wgrad_post(L.ptr(tab), 1, Co, 1e-5, L.stream_ptr()))
    torch.cuda.synchronize()
    assert torch.allclose(dgam.cpu(), gam.grad, rtol=1e-3, atol=1e-3 * float(gam.grad.abs().max()))
    assert torch.allclose(S.cpu(), bet.grad, rtol=1e-4, atol=1e-4)
    assert torch.allclose(dWu.cpu(), w.grad.permute(0, 2, 3, 1), rtol=1e-3, atol=1e-4 * float(w.grad.abs().max()))


def build(**head):
    from dsl_amd import detectors  # noqa: F401
    from dsl_amd.registry import build_detector
    from oracle import rla_oracle as RO
    model = build_detector(rla_model_cfg(**head))
    model.load_state_dict(RO.synth_state_dict(0))
    return model.cuda()


@pytest.mark.parametrize('side', ['1', '0'])
def test_rla_train_step_vs_oracle(monkeypatch, side):
    """(side = '0': every launch on the caller's stream, tuning key side=0.)
    FCOS + RLA_ResNet, forward + loss + hand-written backward on the GPU vs the oracle: losses within 3e-3 of the
    bf16-emulating oracle and 1e-3 of fp32, identical assignment, and every parameter gradient (trainable eval-mode BN
    affine terms, shared recurrent / conv_out weights, zero-padded conv1 rows included) no farther from the fp32 gradient than
    1.6 x the bf16-emulating oracle's own distance."""
    from oracle import fcos_oracle as O
    from oracle import rla_oracle as RO
    from dsl_amd import tuning
    tuning.tune('side')                                    # (DSL_TUNE parsed before the override below)
    monkeypatch.setitem(tuning._values, 'side', side)
    model = build()
    assert len(model.state_dict()) == 465
    rng = np.random.RandomState(1)
    g = torch.Generator().manual_seed(3)
    H, W, B = 128, 192, 2
    img = (torch.randn(B, 3, H, W, generator=g) * 40).bfloat16().float()
    gtb = [T(O.synth_boxes(rng, 4, H=H, W=W, lo=8, hi=min(H, W))) for _ in range(B)]
    gtl = [T(rng.randint(0, 80, len(b)).astype('int64')) for b in gtb]
    losses = model.forward_train(img.cuda(), [dict()] * B, gtb, gtl)
    sum(losses.values()).backward()
    torch.cuda.synchronize()
    got = {k: float(v.detach()) for k, v in losses.items()}
    sd = RO.synth_state_dict(0)
    l32, g32, aux = RO.train_step(sd, img, gtb, gtl, None, emulate_bf16=False)
    lem, gem, _ = RO.train_step(sd, img, gtb, gtl, None, emulate_bf16=True)
    print('losses hip', got, 'oracle-bf16', lem, 'fp32', l32)
    for k in got:
        assert got[k] == pytest.approx(lem[k], rel=3e-3), (k, got[k], lem[k])
        assert got[k] == pytest.approx(l32[k], rel=1e-3), (k, got[k], l32[k])
    plan = next(iter(model._engine.plans.values()))
    _, raux = O.fcos_loss([t.detach() for t in aux['cls']], [t.detach() for t in aux['reg']], [t.detach() for t in aux['ctr']],
                          gtb, gtl, None, return_aux=True)
    assert torch.equal(plan.lossplan.assign_idx.cpu().long(), raux['assign_idx'])
    named = dict(model.named_parameters())
    tk = RO.trainable_keys(sd)
    assert sorted(k for k, p in named.items() if p.requires_grad) == sorted(tk)
    bad, worst, noisy = [], [], []
    for k in tk:
        gh = named[k].grad.detach().cpu()
        e_hip, e_emu = rel_l2(gh, g32[k]), rel_l2(gem[k], g32[k])
        worst.append((round(e_hip, 4), round(e_emu, 4), k))
        if e_emu > 0.25:          # below the bf16 noise floor (deep recurrent-path parameters behind saturated tanh's: fp32
            noisy.append(k)       # gradient norms of 1e-7, and the first trainable stage): the emulation itself is off by > 25 %
            continue
        if float(g32[k].norm()) > 0 and e_hip > 1.6 * e_emu + 5e-3:
            bad.append((k, e_hip, e_emu))
    print('largest hip errors vs fp32 (hip, emulated, key):', sorted(worst, reverse=True)[:8])
    print('BAD', [(k, round(a, 3), round(b, 3), float(g32[k].norm())) for k, a, b in bad])
    print(len(noisy), 'parameters below the noise floor:', noisy)
    assert not bad, bad[:10]
    assert len(noisy) <= 60, noisy          # of 215
    # the zero columns of the padded conv1 rows stay zero: their gradient is exactly zero
    st = model.store
    gw = st.tview('backbone.stages.1.1.conv1.weight', st.grad)
    assert float(gw[..., 512 + 32:].abs().max()) == 0.0 and float(gw[..., :512 + 32].abs().max()) > 0


def test_rla_image_split_backward_chains_give_the_same_gradients(monkeypatch):
    """Tuning key rla_split_bwd=123 (the stage's data-gradient chain as two part-batch chains on two streams, off by default): the
    same gradients as the one-chain backward.  Not bit for bit - a part-batch launch may choose another split-K factor, i.e.
    another fp32 summation order in front of a bf16 rounding, and the recurrent path's BatchNorm records are cut at the
    image boundary - but within bf16 rounding noise of each other (an image offset gone wrong would be an O(1) error).
    (Deleted in round 5 while the branches stayed: ADVICE round 5 - the key and the test are back together.)"""
    from dsl_amd import tuning
    from oracle import fcos_oracle as O
    tuning.tune('side')
    rng = np.random.RandomState(5)
    g = torch.Generator().manual_seed(7)
    H, W, B = 128, 192, 3
    img = (torch.randn(B, 3, H, W, generator=g) * 40).bfloat16().float().cuda()
    gtb = [T(O.synth_boxes(rng, 4, H=H, W=W, lo=8, hi=min(H, W))) for _ in range(B)]
    gtl = [T(rng.randint(0, 80, len(b)).astype('int64')) for b in gtb]
    grads = []
    for split in ('', '123'):
        monkeypatch.setitem(tuning._values, 'rla_split_bwd', split)
        model = build()
        losses = model.forward_train(img, [dict()] * B, gtb, gtl)
        sum(losses.values()).backward()
        torch.cuda.synchronize()
        plan = next(iter(model._engine.plans.values()))
        assert any(n.startswith('rla.recsum.') for n in plan.bufs) == bool(split)
        grads.append({k: p.grad.detach().clone() for k, p in model.named_parameters() if p.requires_grad})
    a, b = grads
    assert a.keys() == b.keys()
    errs = sorted(((rel_l2(b[k], a[k]), k) for k in a if float(a[k].norm()) > 0), reverse=True)
    print('largest differences:', [(round(e, 5), k) for e, k in errs[:6]])
    same = sum(torch.equal(a[k], b[k]) for k in a)
    assert same >= 10                                   # head, FPN and stage 3 down to its first split launch: untouched
    assert errs[0][0] < 3e-2, errs[:5]


def test_rla_tail_kernel_vs_three_launches():
    """dsl_rla_tail_fwd (round 6: a block's recurrent path u = h' + conv_out(out), t = tanh(bn(u)), h = recurrent_conv(t) -
    resnet_rla.py:125-136 - as one launch) against dsl_conv2d -> dsl_bn_tanh_fwd -> dsl_conv2d on the same tensors: shapes with ragged
    14 x 14 tiles, a one-tile image, both stored K widths of the 3x3.  u may differ from the conv path by its fp32 summation order only
    when that launch splits K (not at these sizes): everything is compared bit for bit."""
    import ctypes as C
    from dsl_amd import _lib as L
    from dsl_amd import ops
    g = torch.Generator().manual_seed(21)
    for (n, h, w, c4, tw) in [(2, 25, 42, 256, 128), (1, 14, 14, 512, 64), (3, 9, 31, 1024, 128), (1, 50, 84, 2048, 128)]:
        P = n * h * w
        ldx = c4 + 128
        xh = (torch.randn(P, ldx, generator=g) * 0.7).bfloat16().cuda()
        hin = (torch.randn(P, 32, generator=g) * 0.5).bfloat16().cuda()
        wco = torch.zeros(64, c4)
        wco[:32] = torch.randn(32, c4, generator=g) * (1.0 / c4 ** 0.5)
        wco = wco.bfloat16().cuda()
        wrc = torch.zeros(64, 3, 3, tw)
        wrc[:32, :, :, :32] = torch.randn(32, 3, 3, 32, generator=g) * 0.08
        wrc = wrc.bfloat16().cuda()
        sc, bi = (1 + 0.3 * torch.randn(32, generator=g)).cuda(), (0.2 * torch.randn(32, generator=g)).cuda()
        outs = []
        for fused in (False, True):
            u = torch.full((P, 32), 7.0, dtype=torch.bfloat16, device='cuda')
            t = torch.zeros(P, tw, dtype=torch.bfloat16, device='cuda')
            nxt = torch.zeros(P, ldx, dtype=torch.bfloat16, device='cuda')
            hout = nxt.data_ptr() + c4 * 2
            if fused:
                L.lib.dsl_set_option(b'rla_tail_form', 1)         # the 14 x 14 form: conv_out's K in one chain, as the three launches sum it
                L.check(L.lib.dsl_rla_tail_fwd(L.ptr(xh), ldx, L.ptr(hin), 32, L.ptr(wco), c4, L.ptr(sc), L.ptr(bi), L.ptr(wrc), tw, L.ptr(u),
                                               L.ptr(t), L.ptr(hout), ldx, n, h, w, L.stream_ptr()), 'dsl_rla_tail_fwd')
                L.lib.dsl_set_option(b'rla_tail_form', 0)
            else:
                d1 = ops.conv_desc(xh, wco, u, n=n, grid=[(h, w)], src_hw=[(h, w)], dst_hw=[(h, w)], cs=c4, cd=32, cd_pad=64, ldd=32, kh=1, kw=1,
                                   stride=1, pad=0, lds=ldx, addend=hin, lda=32)
                L.check(L.lib.dsl_conv2d(C.byref(d1), L.stream_ptr()), 'conv_out')
                L.check(L.lib.dsl_bn_tanh_fwd(L.ptr(u), 32, L.ptr(sc), L.ptr(bi), L.ptr(t), tw, P, 32, L.stream_ptr()), 'bn_tanh')
                d2 = ops.conv_desc(t, wrc, hout, n=n, grid=[(h, w)], src_hw=[(h, w)], dst_hw=[(h, w)], cs=tw, cd=32, cd_pad=64, ldd=ldx, kh=3,
                                   kw=3, stride=1, pad=1)
                L.check(L.lib.dsl_conv2d(C.byref(d2), L.stream_ptr()), 'recurrent_conv')
            torch.cuda.synchronize()
            outs.append((u.clone(), t.clone(), nxt.clone()))
        for name, a, b in zip(('u', 't', 'h'), outs[0], outs[1]):
            assert torch.equal(a, b), (n, h, w, c4, tw, name, float((a.float() - b.float()).abs().max()))
        # the K-split tile form (few tiles, long K: conv_out's sum arrives as four partial sums added in K order - another fp32
        # summation order in front of the bf16 rounding of u): within one bf16 step of the 14 x 14 form, and bit-reproducible
        if c4 % 64 == 0:
            L.lib.dsl_set_option(b'rla_tail_form', 2)
            runs = []
            for rep in range(2):
                u2 = torch.full((P, 32), 7.0, dtype=torch.bfloat16, device='cuda')
                t2 = torch.zeros(P, tw, dtype=torch.bfloat16, device='cuda')
                nxt2 = torch.zeros(P, ldx, dtype=torch.bfloat16, device='cuda')
                L.check(L.lib.dsl_rla_tail_fwd(L.ptr(xh), ldx, L.ptr(hin), 32, L.ptr(wco), c4, L.ptr(sc), L.ptr(bi), L.ptr(wrc), tw, L.ptr(u2),
                                               L.ptr(t2), C.c_void_p(nxt2.data_ptr() + c4 * 2), ldx, n, h, w, L.stream_ptr()), 'dsl_rla_tail_fwd')
                torch.cuda.synchronize()
                runs.append((u2, t2, nxt2))
            L.lib.dsl_set_option(b'rla_tail_form', 0)
            for a, b in zip(runs[0], runs[1]):
                assert torch.equal(a, b)
            for name, a, b in zip(('u', 't', 'h'), outs[1], runs[0]):
                d = (a.float() - b.float()).abs()
                assert float(d.max()) <= 2.0 ** -6 * max(1.0, float(a.float().abs().max())), (c4, name, float(d.max()))
                assert float((d > 0).float().mean()) < 0.05, (c4, name)
        assert float(outs[1][2][:, c4:c4 + 32].abs().max()) > 0 and float(outs[1][2][:, :c4].abs().max()) == 0      # only the h columns are written


def test_rla_tail_backward_kernel_vs_two_launches():
    """dsl_rla_tail_bwd (round 6: the recurrent convolution's data gradient + the BN / tanh backward of a block's recurrent path as one
    launch) against dsl_conv2d (mode 1) -> dsl_bn_tanh_bwd on the same tensors: g_u bit for bit, (dgamma, dbeta) within their fp32
    summation order (tile records instead of 256-row records), bit-reproducible."""
    import ctypes as C
    from dsl_amd import _lib as L
    from dsl_amd import ops
    g = torch.Generator().manual_seed(31)
    for (n, h, w, tw) in [(2, 25, 42, 128), (1, 14, 14, 64), (3, 9, 31, 128), (1, 50, 84, 128)]:
        P = n * h * w
        gh = torch.zeros(P, 64)
        gh[:, :32] = torch.randn(P, 32, generator=g) * 0.3
        gh = gh.bfloat16().cuda()
        wT = torch.zeros(tw, 3, 3, 64)                       # the data-gradient pack [ci][r][s][co]
        wT[:32, :, :, :32] = torch.randn(32, 3, 3, 32, generator=g) * 0.08
        wT = wT.bfloat16().cuda()
        u = (torch.randn(P, 32, generator=g) * 1.2).bfloat16().cuda()
        gam, mu, var = 1 + 0.2 * torch.randn(32, generator=g), 0.2 * torch.randn(32, generator=g), 0.5 + torch.rand(32, generator=g)
        sc = (gam / torch.sqrt(var + 1e-5)).cuda()
        bi = (0.1 * torch.randn(32, generator=g)).cuda()
        mu_d, var_d = mu.cuda(), var.cuda()
        t = torch.zeros(P, tw, dtype=torch.bfloat16, device='cuda')
        L.check(L.lib.dsl_bn_tanh_fwd(L.ptr(u), 32, L.ptr(sc), L.ptr(bi), L.ptr(t), tw, P, 32, L.stream_ptr()), 'bn_tanh')
        outs = []
        for mode in ('two', 'fused', 'fused'):
            gu = torch.zeros(P, 64, dtype=torch.bfloat16, device='cuda')
            dgam, dbet = torch.full((32,), float('nan'), device='cuda'), torch.full((32,), float('nan'), device='cuda')
            if mode == 'two':
                gt = torch.empty(P, 32, dtype=torch.bfloat16, device='cuda')
                d1 = ops.conv_desc(gh, wT, gt, n=n, grid=[(h, w)], src_hw=[(h, w)], dst_hw=[(h, w)], cs=64, cd=32, cd_pad=tw, ldd=32, kh=3, kw=3,
                                   stride=1, pad=1, mode=1)
                L.check(L.lib.dsl_conv2d(C.byref(d1), L.stream_ptr()), 'recurrent dgrad')
                ws = torch.empty(L.lib.dsl_bn_tanh_bwd_workspace_bytes(P, 32), dtype=torch.uint8, device='cuda')
                L.check(L.lib.dsl_bn_tanh_bwd(L.ptr(gt), 32, L.ptr(t), tw, L.ptr(u), 32, L.ptr(sc), L.ptr(mu_d), L.ptr(var_d), 1e-5, L.ptr(gu), 64,
                                              L.ptr(dgam), L.ptr(dbet), L.ptr(ws), P, 32, L.stream_ptr()), 'bn_tanh_bwd')
            else:
                ws = torch.full((int(L.lib.dsl_rla_tail_bwd_workspace_bytes(n, h, w)),), 0xff, dtype=torch.uint8, device='cuda')
                L.check(L.lib.dsl_rla_tail_bwd(L.ptr(gh), 64, L.ptr(wT), 64, L.ptr(t), tw, L.ptr(u), L.ptr(sc), L.ptr(mu_d), L.ptr(var_d), 1e-5,
                                               L.ptr(gu), 64, L.ptr(dgam), L.ptr(dbet), L.ptr(ws), n, h, w, L.stream_ptr()), 'dsl_rla_tail_bwd')
            torch.cuda.synchronize()
            outs.append((gu.clone(), dgam.clone(), dbet.clone()))
        assert torch.equal(outs[0][0], outs[1][0]), (n, h, w, float((outs[0][0].float() - outs[1][0].float()).abs().max()))
        assert float(outs[1][0][:, 32:].float().abs().max()) == 0 and float(outs[1][0][:, :32].float().abs().max()) > 0
        for a, b in zip(outs[0][1:], outs[1][1:]):
            assert torch.allclose(a, b, rtol=1e-4, atol=1e-4 * float(a.abs().max())), (n, h, w, float((a - b).abs().max()))
        for a, b in zip(outs[1], outs[2]):
            assert torch.equal(a, b)


def test_rla_tail_fused_step_equals_the_three_launch_step(monkeypatch):
    """Tuning key rla_tail (default 1): the whole training step with the fused recurrent path against the step with the three launches
    per block - the same losses, bit for bit, and the same gradients."""
    from dsl_amd import tuning
    from oracle import fcos_oracle as O
    tuning.tune('side')
    rng = np.random.RandomState(15)
    g = torch.Generator().manual_seed(17)
    H, W, B = 128, 192, 3
    img = (torch.randn(B, 3, H, W, generator=g) * 40).bfloat16().float().cuda()
    gtb = [T(O.synth_boxes(rng, 4, H=H, W=W, lo=8, hi=min(H, W))) for _ in range(B)]
    gtl = [T(rng.randint(0, 80, len(b)).astype('int64')) for b in gtb]
    res = []
    from dsl_amd import _lib as L
    for tail in ('0', '1'):
        monkeypatch.setitem(tuning._values, 'rla_tail', tail)
        model = build()
        # the 14 x 14 tile form sums conv_out's K in one chain, as the three launches do: the two steps then agree bit for bit in the
        # forward pass.  (The K-split form this image size would get differs by a bf16 step in a few per cent of u's elements - the
        # kernel test bounds that, the oracle tests run with it - and the near-zero gradients of the last recurrent BatchNorms are all
        # rounding noise: 100 % apart between ANY two summation orders, no use as a yardstick here.)
        L.lib.dsl_set_option(b'rla_tail_form', 1)
        try:
            losses = model.forward_train(img, [dict()] * B, gtb, gtl)
            sum(losses.values()).backward()
            torch.cuda.synchronize()
        finally:
            L.lib.dsl_set_option(b'rla_tail_form', 0)
        plan = next(iter(model._engine.plans.values()))
        n_tail = sum(1 for o in plan.fwd.items if o.kind == L_OP_RLA() and C_kind(o) == 9)
        assert (n_tail > 0) == (tail == '1'), n_tail
        res.append(({k: float(v.detach()) for k, v in losses.items()},
                    {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.requires_grad}))
    (la, a), (lb, b) = res
    for k in la:
        assert lb[k] == la[k], (k, la[k], lb[k])
    errs = sorted(((rel_l2(b[k], a[k]), k) for k in a if float(a[k].norm()) > 0), reverse=True)
    print('losses', la, lb, 'largest gradient differences:', [(round(e, 5), k) for e, k in errs[:6]])
    assert errs[0][0] < 3e-2, errs[:5]


def L_OP_RLA():
    from dsl_amd import _lib as L
    return L.OP_RLA


def C_kind(o):
    import ctypes as C
    from dsl_amd import _lib as L
    return C.cast(o.desc, C.POINTER(L.RlaDesc)).contents.kind


def test_full_size_dsl_iteration_rla_vs_oracle():
    """BASELINE.json configs[2] with the DSL config's own backbone at its real size: RLA_ResNet, the semi-supervised batch
    3 x (3, 800, 1344) (labeled image, unlabeled image with ignore boxes, its half-scale copy), loss_weight 3, sisoft at full
    weight: the four losses within 1e-3 relative of the fp32 CPU oracle (north_star's bar) and bit-identical assignment
    indices, labels and classification weights on all 67 200 locations."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from dsl_amd.runner import append_half_scale
    from oracle import fcos_oracle as O
    from oracle import rla_oracle as RO
    kw = dict(loss_weight=3.0, soft_weight=1.0)
    model = build(soft_warm_up=0, **kw)
    model.bbox_head.cur_iter = 1                      # past the warm-up window
    b = bench.synth_batch(0, 2)
    rng = np.random.RandomState(77)
    ig = [torch.zeros(0, 4), T(bench.synth_boxes(rng, 3))]
    oimg, ogb, ogl, ogi = O.append_half_scale(b['img'].cpu(), b['gt_bboxes'], b['gt_labels'], ig)
    _, _, _, _, metas = append_half_scale(b['img'][:, :, :8, :8].cpu(), b['gt_bboxes'], b['gt_labels'], ig, b['img_metas'])
    losses = model.forward_train(oimg.cuda(), metas, ogb, ogl, ogi)
    torch.cuda.synchronize()
    assert set(losses) == {'loss_cls', 'loss_bbox', 'loss_centerness', 'loss_sisoft'}
    l32, _, aux = RO.train_step(RO.synth_state_dict(0), oimg, ogb, ogl, ogi, emulate_bf16=False, want_grads=False,
                                soft_scale=1.0, **kw)
    got = {k: float(v.detach()) for k, v in losses.items()}
    lem, _, _ = RO.train_step(RO.synth_state_dict(0), oimg, ogb, ogl, ogi, emulate_bf16=True, want_grads=False, soft_scale=1.0, **kw)
    print('losses hip', got, 'fp32 oracle', l32, 'bf16-emulating oracle', lem)
    for k in ('loss_cls', 'loss_bbox', 'loss_centerness'):
        assert got[k] == pytest.approx(l32[k], rel=1e-3), (k, got[k], l32[k])
    # loss_sisoft = sum over level pairs of mean((logit_full - logit_half)^2) (fcos_head.py:312-328): independent rounding noise
    # of variance s^2 on the two logits adds 2 s^2 to every squared difference - a systematic POSITIVE offset of bf16 storage
    # (measured here: +2.2e-3 relative on RLA_ResNet, +0.9e-3 on ResNet-50), which the bf16-emulating oracle reproduces.  The
    # bar for this one term is therefore 1e-3 against the emulating oracle and 3e-3 against fp32.
    assert got['loss_sisoft'] == pytest.approx(lem['loss_sisoft'], rel=1e-3), (got['loss_sisoft'], lem['loss_sisoft'])
    assert got['loss_sisoft'] == pytest.approx(l32['loss_sisoft'], rel=3e-3), (got['loss_sisoft'], l32['loss_sisoft'])
    plan = [p for p in model._engine.plans.values() if p.N == 3][0]
    with torch.no_grad():
        _, raux = O.fcos_loss([t.detach() for t in aux['cls']], [t.detach() for t in aux['reg']],
                              [t.detach() for t in aux['ctr']], ogb, ogl, ogi, return_aux=True, soft_scale=1.0, **kw)
    assert plan.lossplan.assign_idx.numel() == 3 * 22400
    assert torch.equal(plan.lossplan.assign_idx.cpu().long(), raux['assign_idx'])
    assert torch.equal(plan.lossplan.labels.cpu(), raux['labels'])
    if 'cls_weight' in raux:
        assert torch.equal(plan.lossplan.cls_weight.cpu(), raux['cls_weight'].float())


def test_dsl_config_trains_through_train_detector(tmp_path):
    """configs/fcos_semi/RLA_*.py (its model / optimizer / hook sections restated here - the GPU box has no reference tree;
    tests/test_boundary_cpu.py builds the real file) through dsl_amd.apis.train_detector on the synthetic semi-supervised
    loader: RLA_ResNet student + EMA teacher, SemiEpochBasedRunner, lr / optimizer (clip 35) / EMAOWNHook /
    checkpoint / logger / NumClassCheckHook / self-scheduled UnlabelPredHook, two short epochs."""
    from dsl_amd import detectors  # noqa: F401
    from dsl_amd.apis import train_detector
    from dsl_amd.data import SyntheticSemiLoader
    from dsl_amd.pseudo import PseudoLabelBank
    from dsl_amd.registry import Config, build_detector
    model_cfg = rla_model_cfg(loss_weight=3.0, soft_weight=1.0, soft_warm_up=5000)
    model_cfg['test_cfg'] = dict(nms_pre=1000, min_bbox_size=0, score_thr=0.05, nms=dict(type='nms', iou_threshold=0.6), max_per_img=100)
    cfg = Config(dict(
        model=model_cfg,
        data=dict(samples_per_gpu=2, workers_per_gpu=2, batch_config=dict(ratio=[[1, 1]]),
                  unlabel_train=dict(thres='adathres.json'),
                  unlabel_pred=dict(type='SemiCOCODataset', num_gpus=1, infer_score_thre=0.1, first_score_thre=0.1, use_ema=True,
                                    eval_flip=False, fuse_history=False, first_fuse=False, eval_config={'iou': [0.6]},
                                    eval_checkpoint_config=dict(interval=1, mode='iteration'), preload=6, start_point=1)),
        optimizer=dict(type='SGD', lr=0.01, momentum=0.9, weight_decay=0.0001, paramwise_cfg=dict(bias_lr_mult=2., bias_decay_mult=0.)),
        optimizer_config=dict(grad_clip=dict(max_norm=35, norm_type=2)),
        lr_config=dict(policy='step', warmup='linear', warmup_iters=500, warmup_ratio=1.0 / 3, step=[20, 26]),
        runner=dict(type='SemiEpochBasedRunner', max_epochs=2), checkpoint_config=dict(interval=1),
        ema_config=dict(interval=1, mode='iteration', ratio=0.99, start_point=1), scale_invariant=True,
        log_config=dict(interval=2, hooks=[dict(type='TextLoggerHook')]), custom_hooks=[dict(type='NumClassCheckHook')],
        log_level='WARNING', load_from=None, resume_from=None, workflow=[('train', 1)], work_dir=str(tmp_path)))
    student, teacher = build_detector(cfg.model), build_detector(cfg.model)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        student.init_weights()
        teacher.init_weights()
    bank = PseudoLabelBank(num_classes=80, thres='adathres.json')
    loader = SyntheticSemiLoader(bank, n_labeled=3, n_unlabeled=3, iters_per_epoch=3, H=128, W=192, W_img=190, img_std=30.0)
    from dsl_amd.data import SyntheticValLoader
    cfg.val_dataloader = SyntheticValLoader(n_images=2, H=128, W=192, W_img=190)
    cfg.evaluation = dict(interval=1, metric='bbox')
    w0 = student.store.train.clone()
    runner = train_detector(student, [loader], cfg, distributed=False, validate=True, ema_model=teacher)
    torch.cuda.synchronize()
    assert runner.iter == 6 and runner.epoch == 2 and runner.ema_flag
    kinds = [type(h).__name__ for h in runner._hooks]
    for k in ('StepLrUpdaterHook', 'OptimizerHook', 'EMAOWNHook', 'CheckpointHook', 'TextLoggerHook', 'NumClassCheckHook', 'UnlabelPredHook'):
        assert k in kinds, kinds
    assert torch.isfinite(student.store.train).all() and float((student.store.train.cpu() - w0).abs().max()) > 0
    assert runner.current_lr()[0] == pytest.approx(0.01 * (1 - (1 - 5 / 500) * (1 - 1 / 3)))      # linear warm-up at iteration 5
    import os
    assert os.path.exists(os.path.join(str(tmp_path), 'epoch_2.pth')) and os.path.exists(os.path.join(str(tmp_path), 'epoch_2.pth_ema'))
    ck = torch.load(os.path.join(str(tmp_path), 'latest.pth'), map_location='cpu')
    assert set(ck) == {'meta', 'state_dict', 'optimizer'} and len(ck['state_dict']) == 465 and ck['meta']['iter'] == 6
    ev = [h for h in runner._hooks if type(h).__name__ == 'EvalHook'][0]
    assert [e for e, _ in ev.history] == [1, 2] and os.path.exists(os.path.join(str(tmp_path), 'eval_epoch_2.bbox.json'))
    hook = [h for h in runner._hooks if type(h).__name__ == 'UnlabelPredHook'][0]
    assert hook.bank is bank and hook.n_refreshed == 3 + 1 and bank.thres is not None
