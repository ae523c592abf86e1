"""GPU parity tests of the individual HIP kernels (through the C ABI) against plain fp32 torch-CPU
restatements of the same op on identical (bf16-representable) inputs."""
import ctypes as C
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def K():
    from dsl_amd import _lib as L
    from dsl_amd import ops
    assert torch.cuda.is_available()
    return L, ops


def bf(t):
    return t.bfloat16().float()


def nhwc(t, dev='cuda'):     # NCHW fp32 -> NHWC bf16 on device
    return t.permute(0, 2, 3, 1).contiguous().bfloat16().to(dev)


def from_nhwc(t):            # NHWC (bf16/fp32) device -> NCHW fp32 cpu
    return t.float().cpu().permute(0, 3, 1, 2).contiguous()


def pack_w(w, cd_pad):       # OIHW fp32 -> [cd_pad][kh][kw][cin] bf16
    co, ci, kh, kw = w.shape
    out = torch.zeros(cd_pad, kh, kw, ci)
    out[:co] = w.permute(0, 2, 3, 1)
    return out.bfloat16().cuda()


def pack_w_dgrad(w, cy):     # OIHW fp32 -> [cin][kh][kw][cy] bf16 (cout padded to cy)
    co, ci, kh, kw = w.shape
    out = torch.zeros(ci, kh, kw, cy)
    out[..., :co] = w.permute(1, 2, 3, 0)
    return out.bfloat16().cuda()


def rnd(*shape, g=None, scale=1.0):
    return bf(torch.randn(*shape, generator=g) * scale)


def sync():
    torch.cuda.synchronize()


@pytest.mark.gpu
def test_probe_xcc_local_atomics(K):
    """Workgroups are dealt round-robin to the 8 XCDs - blocks with equal b % 8 share an XCD (what the XCD-aware tile
    mappings of the conv / wgrad kernels assume) - and workgroup-scope float atomics into per-XCD buffers are complete
    after the kernel."""
    L, _ = K
    nb = 4096
    xcc = torch.full((nb,), -1, dtype=torch.int32, device='cuda')
    acc = torch.zeros(8, 256, dtype=torch.float32, device='cuda')
    L.check(L.lib.dsl_probe_xcc(L.ptr(xcc), L.ptr(acc), nb, L.stream_ptr()))
    torch.cuda.synchronize()
    x = xcc.cpu()
    assert int(x.min()) >= 0 and int(x.max()) <= 7
    counts = torch.bincount(x.long(), minlength=8).float()
    assert (counts > 0).all(), counts
    # round-robin dispatch: block b runs on XCD (b + c) % 8, c = where the previous launch stopped - so all blocks
    # with the same b % 8 share an XCD, which is what the tile mappings need
    off = (x - torch.arange(nb, dtype=torch.int32)) % 8
    assert int(off.min()) == int(off.max()), off[:32].tolist()
    assert torch.equal(acc.cpu(), counts[:, None].expand(8, 256)), (acc[:, 0].cpu(), counts)
    print('blocks per XCD', counts.tolist(), 'block->xcc head', x[:16].tolist())


@pytest.mark.gpu
def test_probe_cu_mask(K):
    """hipExtStreamCreateWithCUMask on MI355X: bit i of the 256-bit mask = CU i // 8 of XCD i % 8 (what the side_cus
    library option of api.hip relies on): the first 96 bits give 12 CUs in every XCD."""
    import ctypes as C
    L, _ = K
    nb = 1024
    out = torch.zeros(nb, 2, dtype=torch.int32, device='cuda')
    mask = (C.c_uint32 * 8)()
    for b in range(96):
        mask[b >> 5] |= 1 << (b & 31)
    L.check(L.lib.dsl_probe_cu_mask(mask, 8, L.ptr(out), nb))
    o = out.cpu()
    xcc, hw = o[:, 0], o[:, 1]
    ids = {(int(x), int(h) & 0x7f00) for x, h in zip(xcc.tolist(), hw.tolist())}      # (XCD, SE | SH | CU fields of HW_ID)
    per = torch.bincount(torch.tensor([i[0] for i in ids]), minlength=8)
    assert len(ids) == 96 and per.tolist() == [12] * 8, (len(ids), per.tolist())


# ------------------------------------------------------------------------------------------------
def test_probe_tr16(K):
    """Documents the ds_read_b64_tr_b16 semantics the weight-gradient kernel relies on:
    within each 16-lane group, out[lane i][j] = in[lane 4j + (i>>2)][i & 3]."""
    L, _ = K
    img = torch.arange(4096, dtype=torch.int16, device='cuda')
    off = (torch.arange(64, dtype=torch.int32) * 4).cuda()       # lane l reads u16[4l .. 4l+3]
    out = torch.zeros(64, 4, dtype=torch.int16, device='cuda')
    L.check(L.lib.dsl_probe_tr16(L.ptr(img), L.ptr(off), L.ptr(out), L.stream_ptr()))
    sync()
    got = out.cpu().numpy()
    exp = np.zeros((64, 4), dtype=np.int16)
    for l in range(64):
        g, i = l // 16, l % 16
        for j in range(4):
            src_lane = g * 16 + 4 * j + (i >> 2)
            exp[l, j] = src_lane * 4 + (i & 3)
    print('tr16 lanes 0..3, 16..17:', got[:4].tolist(), got[16:18].tolist())
    assert (got == exp).all(), got[:20].tolist()


CONV_CASES = [
    # name, N, Cin, Cout, H, W, k, stride, pad, flags-ish
    ('3x3_s1', 2, 64, 128, 13, 21, 3, 1, 1),
    ('1x1_s1', 2, 128, 64, 9, 11, 1, 1, 0),
    ('1x1_s2', 1, 64, 256, 14, 18, 1, 2, 0),
    ('3x3_s2', 2, 128, 128, 13, 21, 3, 2, 1),
    ('3x3_big', 1, 256, 256, 25, 42, 3, 1, 1),
]


@pytest.mark.parametrize('force', [0, 1, 2, 3, 4, 5, 6, 7, 8, 15, 1 + (2 << 4), 4 + (3 << 4), 5 + (2 << 4), 6 + (5 << 4), 7 + (2 << 4), 0 + (16 << 4)])
@pytest.mark.parametrize('case', CONV_CASES, ids=[c[0] for c in CONV_CASES])
def test_conv_forward(K, case, force):
    """force: 0 = library's own tile choice, 1..5 = v2 (DMA-to-LDS) tile configs, 15 = v1 kernel."""
    L, ops = K
    _, N, Ci, Co, H, W, k, s, p = case
    ws = torch.empty(64 << 20, dtype=torch.uint8, device='cuda') if force >> 4 else None
    force = (force & 15) | ((force >> 4) & 15) << 4         # bits 8-11 tile config, bits 12-15 forced split-K (16 -> auto)
    bco = {1: 256, 2: 256, 3: 128, 4: 128, 5: 64, 6: 128, 7: 64, 8: 64}.get(force & 15)
    if bco and ((Co + 63) // 64 * 64) % bco:
        pytest.skip('tile does not divide Cout')
    g = torch.Generator().manual_seed(hash(case[0]) % 1000)
    x, w = rnd(N, Ci, H, W, g=g), rnd(Co, Ci, k, k, g=g, scale=1 / math.sqrt(Ci * k * k))
    scale, bias = torch.rand(Co, generator=g) + 0.5, torch.randn(Co, generator=g)
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    res = rnd(N, Co, Ho, Wo, g=g)
    ref = F.relu(F.conv2d(x, w, None, s, p) * scale[None, :, None, None] + bias[None, :, None, None] + res)
    cd_pad = (Co + 63) // 64 * 64
    y = torch.empty(N, Ho, Wo, Co, dtype=torch.bfloat16, device='cuda')
    ops.conv2d(nhwc(x), pack_w(w, cd_pad), y, n=N, grid=[(Ho, Wo)], src_hw=[(H, W)], dst_hw=[(Ho, Wo)],
               cs=Ci, cd=Co, cd_pad=cd_pad, ldd=Co, kh=k, kw=k, stride=s, pad=p,
               flags=L.CONV_RELU_OUT | (force << 8), scale=scale.cuda(), bias=bias.cuda(), addend=nhwc(res), lda=Co,
               workspace=ws)
    sync()
    got = from_nhwc(y)
    assert torch.allclose(got, ref, rtol=1e-2, atol=1e-2), (got - ref).abs().max()
    assert (got - bf(ref)).abs().max() <= 2 ** -7 * ref.abs().max()


def test_conv_stem_small_c(K):
    L, ops = K
    g = torch.Generator().manual_seed(1)
    N, H, W = 2, 37, 45
    x, w = rnd(N, 3, H, W, g=g), rnd(64, 3, 7, 7, g=g, scale=0.1)
    ref = F.relu(F.conv2d(x, w, None, 2, 3))
    Ho, Wo = ref.shape[2:]
    x8 = torch.empty(N, H, W, 8, dtype=torch.bfloat16, device='cuda')
    x_d = x.cuda()
    L.check(L.lib.dsl_pack_image(L.ptr(x_d), L.ptr(x8), N, H, W, L.stream_ptr()))
    sync()
    assert torch.equal(x8.float().cpu()[..., :3], x.permute(0, 2, 3, 1)) and float(x8[..., 3:].abs().max()) == 0
    wp = torch.zeros(64, 7 * 64)      # K = 49 taps * 8 channels = 392, padded to 448
    wp[:, :392] = torch.cat([w.permute(0, 2, 3, 1), torch.zeros(64, 7, 7, 5)], -1).reshape(64, 392)
    y = torch.empty(N, Ho, Wo, 64, dtype=torch.bfloat16, device='cuda')
    ops.conv2d(x8, wp.bfloat16().cuda(), y, n=N, grid=[(Ho, Wo)], src_hw=[(H, W)], dst_hw=[(Ho, Wo)], cs=8, cd=64,
               cd_pad=64, ldd=64, kh=7, kw=7, stride=2, pad=3, flags=L.CONV_RELU_OUT | L.CONV_SMALL_C)
    sync()
    assert torch.allclose(from_nhwc(y), ref, rtol=1e-2, atol=1e-2)


@pytest.mark.gpu
@pytest.mark.parametrize('shape', [(2, 64, 96), (3, 70, 132), (2, 800, 1344)])
def test_stem_pool_reads_the_half_scale_copy_out_of_the_last_image(K, shape):
    """dsl_stem_pool_half: the batch's last image is SemiEpochBasedRunner's scale-invariant copy (semi_epoch_based_runner.py:186-204:
    F.interpolate(img[-1:], (H / 2, W / 2), mode='bilinear') in the top-left corner of a zero canvas), sampled from its source inside
    the stem kernel - bit for bit the output of the same kernel on the batch the framework ops build."""
    L, ops = K
    from dsl_amd.runner import append_half_scale
    B, H, W = shape
    g = torch.Generator().manual_seed(B * 100 + H)
    x = (torch.randn(B, 3, H, W, generator=g) * 50).cuda()
    w = rnd(64, 3, 7, 7, g=g, scale=0.05)
    wg = torch.zeros(64, 7, 24)
    wg[:, :, :21] = w.permute(0, 2, 3, 1).reshape(64, 7, 21)
    wg = torch.cat([wg.reshape(64, 21, 8).permute(1, 0, 2), torch.zeros(1, 64, 8)], 0).bfloat16().cuda().contiguous()
    sc_d, bi_d = (torch.rand(64, generator=g) + 0.5).cuda(), torch.randn(64, generator=g).cuda()
    boxes = [torch.zeros(0, 4, device='cuda')] * B
    labels = [torch.zeros(0, dtype=torch.long, device='cuda')] * B
    metas = [dict(img_shape=(H, W, 3), pad_shape=(H, W, 3))] * B
    full = append_half_scale(x, boxes, labels, None, metas)[0].contiguous()
    assert full.shape[0] == B + 1
    same = append_half_scale(x, boxes, labels, None, metas, materialize=False)
    assert same[0] is x and len(same[1]) == B + 1 and len(same[4]) == B + 1
    PH, PW = (H + 1) // 2, (W + 1) // 2
    PH, PW = (PH + 1) // 2, (PW + 1) // 2
    a = torch.empty(B + 1, PH, PW, 64, dtype=torch.bfloat16, device='cuda')
    b = torch.empty_like(a)
    L.check(L.lib.dsl_stem_pool(L.ptr(full), L.ptr(wg), L.ptr(sc_d), L.ptr(bi_d), L.ptr(a), 64, B + 1, H, W, L.stream_ptr()))
    L.check(L.lib.dsl_stem_pool_half(L.ptr(x), L.ptr(wg), L.ptr(sc_d), L.ptr(bi_d), L.ptr(b), 64, B + 1, H, W, 1, L.stream_ptr()))
    sync()
    assert torch.equal(a, b)
    assert float(b[B].float().abs().max()) > 0
    assert L.lib.dsl_stem_pool_half(L.ptr(x), L.ptr(wg), L.ptr(sc_d), L.ptr(bi_d), L.ptr(b), 64, B + 1, H + 1, W, 1, L.stream_ptr()) != 0     # odd size: refused


@pytest.mark.gpu
@pytest.mark.parametrize('shape', [(2, 37, 45), (1, 64, 96), (2, 128, 192), (1, 800, 1344), (3, 70, 131)])
def test_stem_pool_fused_kernel(K, shape):
    """dsl_stem_pool (image layout + conv1 7x7 / 2 + BN + ReLU + max pool 3x3 / 2 in one kernel) against (a) torch fp32 on the
    bf16-rounded operands with the stem output rounded to bf16 before the pooling - what the three launches it replaces compute -
    and (b) those three launches themselves (dsl_pack_image, dsl_conv2d, dsl_maxpool3x3s2): same values up to the summation
    order of the 147 products in front of one bf16 rounding; odd sizes exercise the tile edges and both paddings; the output
    row stride is honoured (the RLA engine pools into [x | h | zeros] rows)."""
    L, ops = K
    N, H, W = shape
    g = torch.Generator().manual_seed(N * 1000 + H)
    x = torch.randn(N, 3, H, W, generator=g) * 50            # fp32 image, NOT pre-rounded: the kernel rounds it as pack_image does
    w = rnd(64, 3, 7, 7, g=g, scale=0.05)
    scale, bias = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g)
    conv = F.conv2d(bf(x), w, None, 2, 3)
    stem = bf(F.relu(conv * scale.view(1, -1, 1, 1) + bias.view(1, -1, 1, 1)))
    ref = F.max_pool2d(stem, 3, 2, 1)
    PH, PW = ref.shape[2:]
    wg = torch.zeros(64, 7, 24)
    wg[:, :, :21] = w.permute(0, 2, 3, 1).reshape(64, 7, 21)
    wg = torch.cat([wg.reshape(64, 21, 8).permute(1, 0, 2), torch.zeros(1, 64, 8)], 0).bfloat16().cuda().contiguous()
    x_d, sc_d, bi_d = x.cuda(), scale.cuda(), bias.cuda()
    for ld in (64, 192):
        out = torch.full((N, PH, PW, ld), 7.0, dtype=torch.bfloat16, device='cuda')
        L.check(L.lib.dsl_stem_pool(L.ptr(x_d), L.ptr(wg), L.ptr(sc_d), L.ptr(bi_d), L.ptr(out), ld, N, H, W, L.stream_ptr()))
        sync()
        got = from_nhwc(out[..., :64])
        assert float(out[..., 64:].float().sub(7.0).abs().max() if ld > 64 else 0.0) == 0.0        # columns beyond 64 untouched
        err = (got - ref).abs()
        tol = 1e-2 * ref.abs() + 2e-2
        assert bool((err <= tol).all()), (float(err.max()), float(ref.abs().max()))
        assert float((got != ref).float().mean()) < 0.02           # ... and almost everywhere the very same bf16 value
    # (b) the three-launch path on the same operands
    Ho, Wo = conv.shape[2:]
    x8 = torch.empty(N, H, W, 8, dtype=torch.bfloat16, device='cuda')
    L.check(L.lib.dsl_pack_image(L.ptr(x_d), L.ptr(x8), N, H, W, L.stream_ptr()))
    wp = torch.zeros(64, 7 * 64)
    wp[:, :392] = torch.cat([w.permute(0, 2, 3, 1), torch.zeros(64, 7, 7, 5)], -1).reshape(64, 392)
    y = torch.empty(N, Ho, Wo, 64, dtype=torch.bfloat16, device='cuda')
    ops.conv2d(x8, wp.bfloat16().cuda(), y, n=N, grid=[(Ho, Wo)], src_hw=[(H, W)], dst_hw=[(Ho, Wo)], cs=8, cd=64, cd_pad=64,
               ldd=64, kh=7, kw=7, stride=2, pad=3, flags=L.CONV_RELU_OUT | L.CONV_SMALL_C, scale=sc_d, bias=bi_d)
    z = torch.empty(N, PH, PW, 64, dtype=torch.bfloat16, device='cuda')
    L.check(L.lib.dsl_maxpool3x3s2(L.ptr(y), L.ptr(z), N, Ho, Wo, 64, L.stream_ptr()))
    sync()
    three = from_nhwc(z)
    assert float((three != got).float().mean()) < 0.02
    assert bool(((three - got).abs() <= 1e-2 * got.abs() + 2e-2).all())


def _multiseg(tensors):      # list of NCHW fp32 -> level-major flat NHWC bf16 on device
    return torch.cat([t.permute(0, 2, 3, 1).reshape(-1, t.shape[1]) for t in tensors]).bfloat16().cuda()


def test_conv_multilevel_fp32_out_ragged_channels(K):
    """Shared-weight head predictor over 5 level segments, fp32 output with 5 real channels (ldd 8)
    and 80 channels (ldd 80), RELU_IN on the source."""
    L, ops = K
    g = torch.Generator().manual_seed(2)
    N, sizes = 2, [(12, 16), (6, 8), (3, 4), (2, 2), (1, 1)]
    xs = [rnd(N, 256, h, w, g=g) for h, w in sizes]
    for Co, ldd, cd_pad in ((5, 8, 64), (80, 80, 128)):
        w = rnd(Co, 256, 3, 3, g=g, scale=0.02)
        b = torch.randn(Co, generator=g)
        P = sum(h * w_ for h, w_ in sizes) * N
        y = torch.full((P, ldd), -7.0, dtype=torch.float32, device='cuda')
        ops.conv2d(_multiseg(xs), pack_w(w, cd_pad), y, n=N, grid=sizes, src_hw=sizes, dst_hw=sizes, cs=256, cd=Co,
                   cd_pad=cd_pad, ldd=ldd, kh=3, kw=3, stride=1, pad=1, flags=L.CONV_OUT_F32 | L.CONV_RELU_IN,
                   bias=b.cuda())
        sync()
        ref = torch.cat([F.conv2d(F.relu(x), w, b, 1, 1).permute(0, 2, 3, 1).reshape(-1, Co) for x in xs])
        got = y.cpu()
        assert torch.allclose(got[:, :Co], ref, rtol=2e-3, atol=2e-3), (got[:, :Co] - ref).abs().max()
        if ldd > Co:
            assert float((got[:, Co:] + 7.0).abs().max()) == 0.0     # padding columns untouched


def test_conv_fpn_lateral_upsample_add(K):
    L, ops = K
    g = torch.Generator().manual_seed(3)
    N = 2
    x, top = rnd(N, 128, 10, 14, g=g), rnd(N, 64, 5, 7, g=g)
    w, b = rnd(64, 128, 1, 1, g=g, scale=0.1), torch.randn(64, generator=g)
    ref = F.conv2d(x, w, b) + F.interpolate(top, size=(10, 14), mode='nearest')
    y = torch.empty(N, 10, 14, 64, dtype=torch.bfloat16, device='cuda')
    ops.conv2d(nhwc(x), pack_w(w, 64), y, n=N, grid=[(10, 14)], src_hw=[(10, 14)], dst_hw=[(10, 14)], cs=128, cd=64,
               cd_pad=64, ldd=64, kh=1, kw=1, bias=b.cuda(), addend=nhwc(top), lda=64, add_hw=[(5, 7)],
               flags=L.CONV_ADD_UPSAMPLE)
    sync()
    assert torch.allclose(from_nhwc(y), ref, rtol=1e-2, atol=1e-2)


DGRAD_CASES = [('3x3_s1', 2, 128, 64, 11, 13, 3, 1, 1), ('3x3_s2', 1, 128, 128, 13, 21, 3, 2, 1),
               ('1x1_s1', 2, 256, 128, 7, 9, 1, 1, 0), ('3x3_s1_pad80', 1, 256, 80, 9, 9, 3, 1, 1)]


@pytest.mark.parametrize('force', [0, 1, 2, 3, 4, 5, 6, 7, 8, 15])
@pytest.mark.parametrize('case', DGRAD_CASES, ids=[c[0] for c in DGRAD_CASES])
def test_conv_dgrad_transposed(K, case, force):
    """mode 1 gather == autograd input-gradient; epilogue (acc + addend) * (mask > 0)."""
    L, ops = K
    _, N, Ci, Co, H, W, k, s, p = case
    bco = {1: 256, 2: 256, 3: 128, 4: 128, 5: 64, 6: 128, 7: 64, 8: 64}.get(force)
    if bco and Ci % bco:
        pytest.skip('tile does not divide Cin')
    g = torch.Generator().manual_seed(len(case[0]))
    x = rnd(N, Ci, H, W, g=g).requires_grad_()
    w = rnd(Co, Ci, k, k, g=g, scale=1 / math.sqrt(Co * k * k))
    yref = F.conv2d(x, w, None, s, p)
    Ho, Wo = yref.shape[2:]
    dy = rnd(N, Co, Ho, Wo, g=g)
    yref.backward(dy)
    add, msk = rnd(N, Ci, H, W, g=g), rnd(N, Ci, H, W, g=g)
    ref = (x.grad + add) * (msk > 0)
    cy = (Co + 63) // 64 * 64
    dyp = torch.zeros(N, Ho, Wo, cy)
    dyp[..., :Co] = dy.permute(0, 2, 3, 1)
    dx = torch.empty(N, H, W, Ci, dtype=torch.bfloat16, device='cuda')
    ops.conv2d(dyp.bfloat16().cuda(), pack_w_dgrad(w, cy), dx, n=N, grid=[(H, W)], src_hw=[(Ho, Wo)], dst_hw=[(H, W)],
               cs=cy, cd=Ci, cd_pad=Ci, ldd=Ci, kh=k, kw=k, stride=s, pad=p, mode=1, addend=nhwc(add), lda=Ci,
               mask=nhwc(msk), ldm=Ci, flags=L.CONV_MASK_LAST | (force << 8))
    sync()
    got = from_nhwc(dx)
    assert torch.allclose(got, ref, rtol=1e-2, atol=2e-2), (got - ref).abs().max()


def test_conv_dgrad_1x1_s2_scatter(K):
    """1x1 stride-2 data gradient as a 1x1 conv on the dY grid scattered with os=2 into a zeroed dX."""
    L, ops = K
    g = torch.Generator().manual_seed(9)
    N, Ci, Co, H, W = 2, 128, 64, 13, 18
    x = rnd(N, Ci, H, W, g=g).requires_grad_()
    w = rnd(Co, Ci, 1, 1, g=g, scale=0.1)
    y = F.conv2d(x, w, None, 2, 0)
    Ho, Wo = y.shape[2:]
    dy = rnd(N, Co, Ho, Wo, g=g)
    y.backward(dy)
    msk = rnd(N, Ci, H, W, g=g)
    ref = x.grad * (msk > 0)
    dx = torch.zeros(N, H, W, Ci, dtype=torch.bfloat16, device='cuda')
    ops.conv2d(nhwc(dy), pack_w_dgrad(w, 64), dx, n=N, grid=[(Ho, Wo)], src_hw=[(Ho, Wo)], dst_hw=[(H, W)], cs=64,
               cd=Ci, cd_pad=Ci, ldd=Ci, kh=1, kw=1, stride=1, pad=0, mode=1, os=2, mask=nhwc(msk), ldm=Ci,
               flags=L.CONV_MASK_FIRST)
    sync()
    assert torch.allclose(from_nhwc(dx), ref, rtol=1e-2, atol=1e-2)


WG_CASES = [('3x3_s1', 2, 128, 128, 12, 17, 3, 1, 1), ('1x1_s2', 2, 256, 128, 14, 18, 1, 2, 0), ('3x3_256', 2, 256, 256, 9, 13, 3, 1, 1),
            ('3x3_s2', 1, 128, 256, 13, 21, 3, 2, 1), ('1x1_s1_co64', 2, 128, 64, 9, 10, 1, 1, 0)]


@pytest.mark.parametrize('cfg', [None, 0, 1, 2, 3, 4])
@pytest.mark.parametrize('case', WG_CASES, ids=[c[0] for c in WG_CASES])
def test_wgrad(K, case, cfg):
    """cfg: None = library's choice, 0 = v1 kernel, 1..4 = v2 (DMA-to-LDS) tiles 256x256, 256x128, 128x256, 128x128."""
    L, ops = K
    _, N, Ci, Co, H, W, k, s, p = case
    need = {1: (256, 256), 2: (256, 128), 3: (128, 256), 4: (128, 128)}.get(cfg)
    if need and ((Co % need[0] and not (need[0] == 128 and Co % 64 == 0)) or Ci % need[1]):   # 128-cout tiles take cy = 64 (mod 128): zero upper half
        pytest.skip('tile does not divide the channels')
    g = torch.Generator().manual_seed(7 + len(case[0]))
    x = rnd(N, Ci, H, W, g=g)
    w = rnd(Co, Ci, k, k, g=g).requires_grad_()
    y = F.conv2d(x, w, None, s, p)
    Ho, Wo = y.shape[2:]
    dy = rnd(N, Co, Ho, Wo, g=g)
    y.backward(dy)
    scale = torch.rand(Co, generator=g) + 0.5
    ref = (w.grad * scale[:, None, None, None]).permute(0, 2, 3, 1)      # KRSC
    dw = torch.empty(Co, k, k, Ci, dtype=torch.float32, device='cuda')
    db = torch.empty(Co, dtype=torch.float32, device='cuda')
    ops.conv2d_wgrad(nhwc(dy), nhwc(x), dw, n=N, grid=[(Ho, Wo)], src_hw=[(H, W)], cs=Ci, cy=Co, cd=Co, kh=k, kw=k,
                     stride=s, pad=p, scale=scale.cuda(), db=db, force_cfg=cfg)
    sync()
    got = dw.cpu()
    tol = 2e-2 * float(ref.abs().max())
    assert torch.allclose(got, ref, rtol=1e-2, atol=tol), (got - ref).abs().max()
    assert torch.allclose(db.cpu(), dy.sum((0, 2, 3)), rtol=1e-3, atol=1e-2)


@pytest.mark.parametrize('count', [2, 3, 8])
@pytest.mark.parametrize('case', WG_CASES, ids=[c[0] for c in WG_CASES])
def test_wgrad_group(K, case, count):
    """dsl_conv2d_wgrad_group: `count` convolutions of one geometry in one launch == the same convolutions one by one."""
    L, ops = K
    _, N, Ci, Co, H, W, k, s, p = case
    g = torch.Generator().manual_seed(11 + count)
    descs, refs, outs = [], [], []
    for m in range(count):
        x = rnd(N, Ci, H, W, g=g)
        w = rnd(Co, Ci, k, k, g=g).requires_grad_()
        y = F.conv2d(x, w, None, s, p)
        Ho, Wo = y.shape[2:]
        dy = rnd(N, Co, Ho, Wo, g=g)
        y.backward(dy)
        scale = (torch.rand(Co, generator=g) + 0.5) if m % 2 == 0 else None
        ref = w.grad if scale is None else w.grad * scale[:, None, None, None]
        dw = torch.full((Co, k, k, Ci), float('nan'), dtype=torch.float32, device='cuda')
        db = torch.empty(Co, dtype=torch.float32, device='cuda') if m != 1 else None
        descs.append(ops.wgrad_desc(nhwc(dy), nhwc(x), dw, n=N, grid=[(Ho, Wo)], src_hw=[(H, W)], cs=Ci, cy=Co, cd=Co,
                                    kh=k, kw=k, stride=s, pad=p, scale=None if scale is None else scale.cuda(), db=db))
        refs.append((ref.permute(0, 2, 3, 1), dy.sum((0, 2, 3))))
        outs.append((dw, db))
    ops.conv2d_wgrad_group(descs)
    sync()
    for (ref, rdb), (dw, db) in zip(refs, outs):
        got = dw.cpu()
        assert torch.allclose(got, ref, rtol=1e-2, atol=2e-2 * float(ref.abs().max())), (got - ref).abs().max()
        if db is not None:
            assert torch.allclose(db.cpu(), rdb, rtol=1e-3, atol=1e-2)


MULTI_SETS = {
    # every sub-launch: (members, N, Ci, Co, H, W, k, stride, pad); one tile configuration per set
    'cfg1_256x256': [(3, 2, 256, 256, 40, 52, 3, 1, 1), (1, 2, 512, 256, 20, 26, 1, 1, 0), (2, 2, 256, 512, 5, 7, 1, 1, 0),
                     (1, 1, 512, 256, 13, 21, 1, 2, 0), (1, 2, 256, 256, 2, 3, 3, 2, 1)],
    'cfg3_128x256': [(1, 2, 256, 80, 30, 40, 3, 1, 1), (1, 2, 256, 5, 30, 40, 3, 1, 1), (2, 2, 512, 128, 9, 11, 1, 1, 0)],
    'cfg4_128x128': [(2, 2, 128, 128, 12, 17, 3, 1, 1), (1, 2, 128, 64, 30, 31, 1, 1, 0)],
    'cfg2_256x128': [(2, 2, 128, 512, 12, 17, 1, 1, 0), (1, 2, 128, 256, 40, 41, 3, 1, 1)],
}


@pytest.mark.parametrize('name', list(MULTI_SETS))
def test_wgrad_multi(K, name):
    """dsl_conv2d_wgrad_multi: sub-launches of different geometries as one grid + one reduce grid == autograd, and
    bit-identical on repeat (no atomics in the weight sums); direct (one split) and reduced sub-launches both occur."""
    L, ops = K
    import ctypes as C
    g = torch.Generator().manual_seed(3 + len(name))
    subs, checks = [], []
    for (cnt, N, Ci, Co, H, W, k, s, p) in MULTI_SETS[name]:
        cy = (Co + 63) // 64 * 64
        if cy == 64 and name.startswith('cfg3'):
            cy = 64
        grp = []
        for m in range(cnt):
            x = rnd(N, Ci, H, W, g=g)
            w = rnd(Co, Ci, k, k, g=g).requires_grad_()
            y = F.conv2d(x, w, None, s, p)
            Ho, Wo = y.shape[2:]
            dy = rnd(N, Co, Ho, Wo, g=g)
            y.backward(dy)
            scale = (torch.rand(Co, generator=g) + 0.5) if m % 2 == 0 else None
            ref = w.grad if scale is None else w.grad * scale[:, None, None, None]
            dyp = torch.cat([dy.permute(0, 2, 3, 1), torch.zeros(N, Ho, Wo, cy - Co)], -1).reshape(-1, cy).bfloat16().cuda()
            dw = torch.full((Co, k, k, Ci), float('nan'), dtype=torch.float32, device='cuda')
            db = torch.full((Co,), float('nan'), dtype=torch.float32, device='cuda') if m != 1 else None
            grp.append(ops.wgrad_desc(dyp, nhwc(x), dw, n=N, grid=[(Ho, Wo)], src_hw=[(H, W)], cs=Ci, cy=cy, cd=Co, kh=k, kw=k,
                                      stride=s, pad=p, scale=None if scale is None else scale.cuda(), db=db))
            checks.append((ref.permute(0, 2, 3, 1), dy.sum((0, 2, 3)), dw, db))
        subs.append(grp)
    cfgs = {L.lib.dsl_wgrad_multi_config(C.byref(d)) for grp in subs for d in grp}
    assert cfgs == {int(name[3])}, cfgs
    plan = ops.WgradMulti(subs)
    plan.run()
    sync()
    first = []
    for ref, rdb, dw, db in checks:
        got = dw.cpu()
        assert torch.allclose(got, ref, rtol=1e-2, atol=2e-2 * float(ref.abs().max())), (name, (got - ref).abs().max())
        if db is not None:
            assert torch.allclose(db.cpu(), rdb, rtol=1e-3, atol=1e-2)
        first.append((got.clone(), None if db is None else db.cpu().clone()))
        dw.fill_(float('nan'))
        if db is not None:
            db.fill_(float('nan'))
    plan.run()
    sync()
    for (ref, rdb, dw, db), (f, fdb) in zip(checks, first):
        assert torch.equal(dw.cpu(), f)
        if db is not None:                 # the bias gradients are summed inside the launch in a fixed order, too
            assert torch.equal(db.cpu(), fdb)


def test_wgrad_multi_rejects_mixed_tile_configurations(K):
    L, ops = K
    z = lambda *s: torch.zeros(*s, device='cuda', dtype=torch.bfloat16)
    a = ops.wgrad_desc(z(2 * 8 * 8, 256), z(2 * 8 * 8, 256), torch.zeros(256, 1, 1, 256, device='cuda'), n=2, grid=[(8, 8)],
                       src_hw=[(8, 8)], cs=256, cy=256, cd=256, kh=1, kw=1)
    b = ops.wgrad_desc(z(2 * 8 * 8, 128), z(2 * 8 * 8, 128), torch.zeros(128, 1, 1, 128, device='cuda'), n=2, grid=[(8, 8)],
                       src_hw=[(8, 8)], cs=128, cy=128, cd=128, kh=1, kw=1)
    with pytest.raises(RuntimeError, match='tile configuration'):
        ops.WgradMulti([[a], [b]])


def test_wgrad_group_rejects_mixed_geometry(K):
    L, ops = K
    a = ops.wgrad_desc(torch.zeros(2 * 8 * 8, 128, device='cuda', dtype=torch.bfloat16), torch.zeros(2 * 8 * 8, 128, device='cuda', dtype=torch.bfloat16),
                       torch.zeros(128, 1, 1, 128, device='cuda'), n=2, grid=[(8, 8)], src_hw=[(8, 8)], cs=128, cy=128, cd=128, kh=1, kw=1)
    b = ops.wgrad_desc(torch.zeros(2 * 8 * 8, 128, device='cuda', dtype=torch.bfloat16), torch.zeros(2 * 8 * 8, 256, device='cuda', dtype=torch.bfloat16),
                       torch.zeros(128, 1, 1, 256, device='cuda'), n=2, grid=[(8, 8)], src_hw=[(8, 8)], cs=256, cy=128, cd=128, kh=1, kw=1)
    with pytest.raises(RuntimeError, match='different geometry'):
        ops.conv2d_wgrad_group([a, b], workspace=torch.empty(64 << 20, dtype=torch.uint8, device='cuda'))


def test_wgrad_multilevel_padded_cout(K):
    """Head predictor weight gradient: 5 level segments, dY rows padded 80 -> 128 channels."""
    L, ops = K
    g = torch.Generator().manual_seed(21)
    N, sizes, Co, cy = 2, [(12, 16), (6, 8), (3, 4), (2, 2), (1, 1)], 80, 128
    xs = [rnd(N, 256, h, w, g=g) for h, w in sizes]
    dys = [rnd(N, Co, h, w, g=g) for h, w in sizes]
    w = torch.zeros(Co, 256, 3, 3, requires_grad=True)
    sum((F.conv2d(x, w, None, 1, 1) * dy).sum() for x, dy in zip(xs, dys)).backward()
    ref = w.grad.permute(0, 2, 3, 1)
    dyp = torch.cat([torch.cat([d.permute(0, 2, 3, 1), torch.zeros(N, d.shape[2], d.shape[3], cy - Co)], -1)
                     .reshape(-1, cy) for d in dys]).bfloat16().cuda()
    dw = torch.empty(Co, 3, 3, 256, dtype=torch.float32, device='cuda')
    ops.conv2d_wgrad(dyp, _multiseg(xs), dw, n=N, grid=sizes, src_hw=sizes, cs=256, cy=cy, cd=Co, kh=3, kw=3, stride=1,
                     pad=1)
    sync()
    assert torch.allclose(dw.cpu(), ref, rtol=1e-2, atol=2e-2 * float(ref.abs().max()))


@pytest.mark.parametrize('co', [5, 8])
def test_wgrad_small_cout(K, co):
    """conv_reg + conv_centerness weight gradient: 5 (8) real output channels in 64-channel dY rows over 5 level
    segments, with the bias gradient (cleared by the reduce kernel, accumulated by the column-sum kernel)."""
    L, ops = K
    g = torch.Generator().manual_seed(23)
    N, sizes, cy = 2, [(12, 16), (6, 8), (3, 4), (2, 2), (1, 1)], 64
    xs = [rnd(N, 256, h, w, g=g) for h, w in sizes]
    dys = [rnd(N, co, h, w, g=g) for h, w in sizes]
    w = torch.zeros(co, 256, 3, 3, requires_grad=True)
    sum((F.conv2d(x, w, None, 1, 1) * dy).sum() for x, dy in zip(xs, dys)).backward()
    ref = w.grad.permute(0, 2, 3, 1)
    dyp = torch.cat([torch.cat([d.permute(0, 2, 3, 1), torch.zeros(N, d.shape[2], d.shape[3], cy - co)], -1)
                     .reshape(-1, cy) for d in dys]).bfloat16().cuda()
    dw = torch.full((co, 3, 3, 256), float('nan'), dtype=torch.float32, device='cuda')
    db = torch.full((co,), float('nan'), dtype=torch.float32, device='cuda')
    ops.conv2d_wgrad(dyp, _multiseg(xs), dw, n=N, grid=sizes, src_hw=sizes, cs=256, cy=cy, cd=co, kh=3, kw=3, stride=1,
                     pad=1, db=db)
    sync()
    assert torch.allclose(dw.cpu(), ref, rtol=1e-2, atol=2e-2 * float(ref.abs().max())), (dw.cpu() - ref).abs().max()
    rdb = sum(d.sum((0, 2, 3)) for d in dys)
    assert torch.allclose(db.cpu(), rdb, rtol=1e-3, atol=1e-2)


def test_groupnorm_relu_fwd_bwd(K):
    L, ops = K
    g = torch.Generator().manual_seed(4)
    N, sizes, Cc = 2, [(12, 20), (6, 10), (3, 5), (2, 3), (1, 2)], 256
    xs = [rnd(N, Cc, h, w, g=g, scale=2.0).requires_grad_() for h, w in sizes]
    gamma = (1 + 0.2 * torch.randn(Cc, generator=g)).requires_grad_()
    beta = (0.3 * torch.randn(Cc, generator=g)).requires_grad_()
    ys = [F.relu(F.group_norm(x, 32, gamma, beta, 1e-5)) for x in xs]
    dys = [rnd(N, Cc, h, w, g=g) for h, w in sizes]
    sum((y * d).sum() for y, d in zip(ys, dys)).backward()
    P = sum(h * w for h, w in sizes) * N
    x_d = _multiseg([x.detach() for x in xs])
    y_d = torch.empty(P, Cc, dtype=torch.bfloat16, device='cuda')
    stats = torch.empty(5 * N * 32, 2, device='cuda')
    ga, be = gamma.detach().cuda(), beta.detach().cuda()
    d = ops.gn_desc(x_d, y_d, ga, be, stats, n=N, hw=sizes)
    L.check(L.lib.dsl_groupnorm_relu_fwd(C.byref(d), L.stream_ptr()))
    sync()
    ref_y = torch.cat([y.detach().permute(0, 2, 3, 1).reshape(-1, Cc) for y in ys])
    assert torch.allclose(y_d.float().cpu(), ref_y, rtol=1e-2, atol=1e-2)
    dy_d = _multiseg(dys)
    dx_d = torch.empty_like(y_d)
    dgam, dbet, dbias = (torch.full((Cc,), float('nan'), device='cuda') for _ in range(3))      # overwritten, never accumulated
    d = ops.gn_desc(x_d, y_d, ga, be, stats, n=N, hw=sizes, dy=dy_d, dx=dx_d, dgamma=dgam, dbeta=dbet, dbias=dbias)
    L.check(L.lib.dsl_groupnorm_relu_bwd(C.byref(d), L.stream_ptr()))
    sync()
    first = [t.clone() for t in (dx_d, dgam, dbet, dbias, y_d)]
    L.check(L.lib.dsl_groupnorm_relu_fwd(C.byref(d), L.stream_ptr()))
    L.check(L.lib.dsl_groupnorm_relu_bwd(C.byref(d), L.stream_ptr()))
    sync()
    for a, b in zip(first, (dx_d, dgam, dbet, dbias, y_d)):          # fixed-order reductions: bit-identical reruns
        assert torch.equal(a, b)
    ref_dx = torch.cat([x.grad.permute(0, 2, 3, 1).reshape(-1, Cc) for x in xs])
    got = dx_d.float().cpu()
    assert torch.allclose(got, ref_dx, rtol=2e-2, atol=2e-2 * float(ref_dx.abs().max())), (got - ref_dx).abs().max()
    assert torch.allclose(dgam.cpu(), gamma.grad, rtol=1e-2, atol=1e-2 * float(gamma.grad.abs().max()))
    assert torch.allclose(dbet.cpu(), beta.grad, rtol=1e-2, atol=1e-2 * float(beta.grad.abs().max()))
    # bias gradient of the conv in front of the norm = column sum of dx (here of the fp32 reference dx)
    ref_db = ref_dx.sum(0)
    assert torch.allclose(dbias.cpu(), ref_db, rtol=1e-2, atol=1e-2 * float(ref_db.abs().max()) + 1e-3), (dbias.cpu() - ref_db).abs().max()


@pytest.mark.parametrize('N,sizes,force', [
    (2, [(12, 20), (6, 10), (3, 5), (2, 3), (1, 2)], 0),          # every (level, image) row inside one or two pixel tiles
    (3, [(40, 52), (20, 26), (10, 13), (5, 7), (3, 4)], 0),       # rows that span many tiles, ragged ends
    (2, [(40, 52), (20, 26), (10, 13), (5, 7), (3, 4)], 4),       # 128 x 128 tile (two cout tiles write one record)
    (2, [(40, 52), (20, 26), (10, 13), (5, 7), (3, 4)], 7),       # 64 x 64 tile
    (2, [(40, 52), (20, 26), (10, 13), (5, 7), (3, 4)], 3),       # 128 x 256 tile
    (2, [(100, 168), (50, 84), (25, 42), (13, 21), (7, 11)], 0),  # the benchmark's head: 88 tiles per row of the first level
])
def test_groupnorm_statistics_from_the_conv_epilogue(K, N, sizes, force):
    """conv -> GN -> ReLU (ConvModule, anchor_free_head.py:104-133) with the statistics left by the convolution's epilogue
    (dsl_conv_desc.gn_ws) against the two-pass GroupNorm on the same convolution output."""
    L, ops = K
    g = torch.Generator().manual_seed(41)
    Cc = 256
    P = sum(h * w for h, w in sizes) * N
    x = _multiseg([rnd(N, Cc, h, w, g=g) for h, w in sizes])
    w = (torch.randn(Cc, 9 * Cc, generator=g) * 0.03).bfloat16().cuda()
    bias = (0.2 * torch.randn(Cc, generator=g)).cuda()
    ga, be = (1 + 0.2 * torch.randn(Cc, generator=g)).cuda(), (0.3 * torch.randn(Cc, generator=g)).cuda()
    outs = []
    for fused in (0, 1):
        pre = torch.zeros(P, Cc, dtype=torch.bfloat16, device='cuda')
        y = torch.empty_like(pre)
        stats = torch.empty(5 * N * 32, 2, device='cuda')
        gd = ops.gn_desc(pre, y, ga, be, stats, n=N, hw=sizes)
        cd = ops.conv_desc(x, w, pre, n=N, grid=sizes, src_hw=sizes, dst_hw=sizes, cs=Cc, cd=Cc, cd_pad=Cc, ldd=Cc, kh=3, kw=3,
                           stride=1, pad=1, flags=force << 8, bias=bias)
        if fused:
            assert L.lib.dsl_conv2d_gn_fusable(C.byref(cd)) == 1
            cd.gn_ws = gd.workspace
            gd.conv_stats = 1
            gd._keep[5].fill_(0xff)                   # (nothing relies on a zeroed workspace: an unwritten record would read NaN)
        L.check(L.lib.dsl_conv2d(C.byref(cd), L.stream_ptr()), 'dsl_conv2d')
        L.check(L.lib.dsl_groupnorm_relu_fwd(C.byref(gd), L.stream_ptr()), 'gn')
        sync()
        outs.append((pre.clone(), y.clone(), stats.clone()))
        if fused:                                    # fixed-order reductions: a second run gives the same bits
            L.check(L.lib.dsl_conv2d(C.byref(cd), L.stream_ptr()), 'dsl_conv2d')
            L.check(L.lib.dsl_groupnorm_relu_fwd(C.byref(gd), L.stream_ptr()), 'gn')
            sync()
            assert torch.equal(y, outs[-1][1]) and torch.equal(stats, outs[-1][2])
    (p0, y0, s0), (p1, y1, s1) = outs
    assert torch.equal(p0, p1)                       # the convolution's own output does not change
    assert torch.allclose(s0, s1, rtol=2e-5, atol=2e-6), (s0 - s1).abs().max()
    d = (y0.float() - y1.float()).abs()
    assert float(d.max()) <= 2 ** -7 * float(y0.float().abs().max()), float(d.max())
    assert float((d > 0).float().mean()) < 0.01      # the same statistics up to fp32 summation order: rare 1-ulp flips only
    # a split-K launch, an fp32 output or an addend cannot leave records: the query says so and the launch refuses
    cd2 = ops.conv_desc(x, w, torch.empty(P, Cc, device='cuda'), n=N, grid=sizes, src_hw=sizes, dst_hw=sizes, cs=Cc, cd=Cc, cd_pad=Cc,
                        ldd=Cc, kh=3, kw=3, stride=1, pad=1, flags=L.CONV_OUT_F32)
    assert L.lib.dsl_conv2d_gn_fusable(C.byref(cd2)) == 0
    cd2.gn_ws = gd.workspace
    assert L.lib.dsl_conv2d(C.byref(cd2), L.stream_ptr()) != 0


@pytest.mark.parametrize('N,sizes,force', [
    (2, [(12, 20), (6, 10), (3, 5), (2, 3), (1, 2)], 0),
    (3, [(40, 52), (20, 26), (10, 13), (5, 7), (3, 4)], 1),       # 256 x 192: the reduction scratch takes two rounds
    (2, [(40, 52), (20, 26), (10, 13), (5, 7), (3, 4)], 4),       # 128 x 128: not instantiated, the launch refuses
    (2, [(40, 52), (20, 26), (10, 13), (5, 7), (3, 4)], 2),       # 256 x 128
    (2, [(100, 168), (50, 84), (25, 42), (13, 21), (7, 11)], 0),  # the benchmark's head
])
def test_groupnorm_backward_records_from_the_data_gradient_epilogue(K, N, sizes, force):
    """dY of a tower layer's GroupNorm + ReLU comes out of the next layer's data gradient: with dsl_conv_desc.gn_x that launch also
    leaves the norm's backward block records, and dsl_groupnorm_relu_bwd (conv_stats = 1) is one pass - against the two-pass
    backward on the same dY (autograd of ConvModule conv -> GN -> ReLU, anchor_free_head.py:104-133)."""
    L, ops = K
    g = torch.Generator().manual_seed(43)
    Cc = 256
    P = sum(h * w for h, w in sizes) * N
    x = _multiseg([rnd(N, Cc, h, w, g=g, scale=2.0) for h, w in sizes])            # the norm's input of the forward pass
    gnext = _multiseg([rnd(N, Cc, h, w, g=g) for h, w in sizes])                   # gradient at the next conv's output
    w = torch.randn(Cc, Cc, 3, 3, generator=g) * 0.03
    ga, be = (1 + 0.2 * torch.randn(Cc, generator=g)).cuda(), (0.3 * torch.randn(Cc, generator=g)).cuda()
    y = torch.empty(P, Cc, dtype=torch.bfloat16, device='cuda')
    stats = torch.empty(5 * N * 32, 2, device='cuda')
    L.check(L.lib.dsl_groupnorm_relu_fwd(C.byref(ops.gn_desc(x, y, ga, be, stats, n=N, hw=sizes)), L.stream_ptr()))
    wT = pack_w_dgrad(w, Cc)
    outs = []
    for fused in (0, 1):
        dy = torch.zeros(P, Cc, dtype=torch.bfloat16, device='cuda')
        dx = torch.empty_like(dy)
        dgam, dbet, dbias = (torch.full((Cc,), float('nan'), device='cuda') for _ in range(3))
        gd = ops.gn_desc(x, y, ga, be, stats, n=N, hw=sizes, dy=dy, dx=dx, dgamma=dgam, dbeta=dbet, dbias=dbias)
        cd = ops.conv_desc(gnext, wT, dy, n=N, grid=sizes, src_hw=sizes, dst_hw=sizes, cs=Cc, cd=Cc, cd_pad=Cc, ldd=Cc, kh=3, kw=3,
                           stride=1, pad=1, mode=1, flags=force << 8)
        if fused:
            cd.gn_x = L.ptr(x)
            ok = L.lib.dsl_conv2d_gn_fusable(C.byref(cd))
            assert force == 0 or ok == (1 if force in (1, 2) else 0)        # instantiated for the two 256-cout tiles
            if not ok:
                cd.gn_ws = gd.workspace
                assert L.lib.dsl_conv2d(C.byref(cd), L.stream_ptr()) != 0          # refused, not silently skipped
                return
            cd.gn_ws, cd.gn_gamma, cd.gn_beta, cd.gn_stats = gd.workspace, L.ptr(ga), L.ptr(be), L.ptr(stats)
            gd.conv_stats = 1
            gd._keep[5].fill_(0xff)
        for rep in range(2):
            L.check(L.lib.dsl_conv2d(C.byref(cd), L.stream_ptr()), 'dsl_conv2d')
            L.check(L.lib.dsl_groupnorm_relu_bwd(C.byref(gd), L.stream_ptr()), 'gn bwd')
            sync()
            if rep == 0:
                outs.append([t.clone() for t in (dy, dx, dgam, dbet, dbias)])
            else:                                    # fixed-order reductions: the same bits again
                for a, b in zip(outs[-1], (dy, dx, dgam, dbet, dbias)):
                    assert torch.equal(a, b)
    a, b = outs
    assert torch.equal(a[0], b[0])                   # the data gradient itself does not change
    tol = 2 ** -7 * float(a[1].float().abs().max())
    assert float((a[1].float() - b[1].float()).abs().max()) <= tol
    assert float(((a[1] != b[1]).float()).mean()) < 0.01
    for u, v in zip(a[2:], b[2:]):                   # parameter gradients: the same sums in another (fixed) order
        assert torch.allclose(u, v, rtol=1e-4, atol=1e-4 * float(u.abs().max())), (u - v).abs().max()


def test_maxpool_sum2x2_colsum(K):
    L, _ = K
    g = torch.Generator().manual_seed(5)
    x = rnd(2, 64, 37, 45, g=g)
    ref = F.max_pool2d(x, 3, 2, 1)
    y = torch.empty(2, ref.shape[2], ref.shape[3], 64, dtype=torch.bfloat16, device='cuda')
    L.check(L.lib.dsl_maxpool3x3s2(L.ptr(nhwc(x)), L.ptr(y), 2, 37, 45, 64, L.stream_ptr()))
    gch = rnd(2, 64, 10, 14, g=g)
    top = torch.zeros(2, 64, 5, 7, requires_grad=True)
    (F.interpolate(top, size=(10, 14), mode='nearest') * gch).sum().backward()
    out = torch.empty(2, 5, 7, 64, dtype=torch.bfloat16, device='cuda')
    L.check(L.lib.dsl_sum2x2(L.ptr(nhwc(gch)), L.ptr(out), 2, 5, 7, 10, 14, 64, L.stream_ptr()))
    m = rnd(1000, 80, g=g)
    mp = torch.cat([m, torch.ones(1000, 48)], 1).bfloat16().cuda()
    cs = torch.empty(80, device='cuda')
    L.check(L.lib.dsl_colsum(L.ptr(mp), L.ptr(cs), 1000, 80, 128, L.stream_ptr()))
    sync()
    assert torch.equal(from_nhwc(y), ref)
    assert torch.allclose(from_nhwc(out), top.grad, rtol=1e-2, atol=1e-2)
    assert torch.allclose(cs.cpu(), m.sum(0), rtol=1e-4, atol=1e-3)


def test_sgd_ema_cast_packdgrad(K):
    L, _ = K
    from oracle import fcos_oracle as O
    g = torch.Generator().manual_seed(6)
    n = 4096
    p0, m0 = torch.randn(n, generator=g), torch.zeros(n)
    grp = (torch.rand(n, generator=g) < 0.1).to(torch.uint8)
    p, m = p0.clone().cuda(), m0.clone().cuda()
    p16 = torch.empty(n, dtype=torch.bfloat16, device='cuda')
    pr, mr = p0.clone(), m0.clone()
    for step in range(3):
        gr = torch.randn(n, generator=g) * 30
        gn = torch.zeros(1, device='cuda')
        gr_d, grp_d = gr.cuda(), grp.cuda()      # keep the device tensors alive across the launches
        L.check(L.lib.dsl_sumsq(L.ptr(gr_d), n, L.ptr(gn), L.stream_ptr()))
        L.check(L.lib.dsl_sgd_step(L.ptr(p), L.ptr(gr_d), L.ptr(m), L.ptr(p16), L.ptr(grp_d), n, 0.01, 0.9, 1e-4,
                                   2.0, 0.0, L.ptr(gn), 35.0, int(step == 0), L.stream_ptr()))
        coef = min(35.0 / (float(gr.double().norm()) + 1e-6), 1.0)
        lr = torch.where(grp.bool(), torch.tensor(0.02), torch.tensor(0.01))
        wd = torch.where(grp.bool(), torch.tensor(0.0), torch.tensor(1e-4))
        d = gr * coef + wd * pr
        mr = d if step == 0 else 0.9 * mr + d
        pr = pr - lr * mr
    sync()
    assert torch.allclose(p.cpu(), pr, rtol=1e-5, atol=1e-6)
    assert torch.equal(p16.cpu(), p.cpu().bfloat16())
    # the optimizer's form of the squared norm: fixed summation order (bit-identical on repeat), overwrites its output
    big = torch.randn(3_000_001, generator=g).cuda()
    ws, o1, o2 = torch.zeros(1024, device='cuda'), torch.full((1,), 7.0, device='cuda'), torch.full((1,), -3.0, device='cuda')
    L.check(L.lib.dsl_sumsq_det(L.ptr(big), big.numel(), L.ptr(o1), L.ptr(ws), L.stream_ptr()))
    L.check(L.lib.dsl_sumsq_det(L.ptr(big), big.numel(), L.ptr(o2), L.ptr(ws), L.stream_ptr()))
    sync()
    assert float(o1) == float(o2) and float(o1) == pytest.approx(float(big.double().pow(2).sum()), rel=1e-5)
    t, s = torch.randn(n, generator=g), torch.randn(n, generator=g)
    td, sd_ = t.clone().cuda(), s.cuda()
    L.check(L.lib.dsl_ema_lerp(L.ptr(td), L.ptr(sd_), n, 0.99, L.stream_ptr()))
    sync()
    assert torch.allclose(td.cpu(), O.ema_update({'w': t}, {'w': s}, 0.99)['w'], rtol=1e-6, atol=1e-7)
    td2, t16 = t.clone().cuda(), torch.zeros(n, dtype=torch.bfloat16, device='cuda')      # the same with the bf16 forward copy in the same pass
    L.check(L.lib.dsl_ema_lerp_bf16(L.ptr(td2), L.ptr(sd_), L.ptr(t16), n, 0.99, L.stream_ptr()))
    sync()
    assert torch.equal(td2, td) and torch.equal(t16, td.bfloat16())
    w = torch.randn(80, 3, 3, 256, generator=g)
    sc = torch.rand(80, generator=g) + 0.5
    out = torch.full((256, 3, 3, 128), 5.0).bfloat16().cuda()
    w_d, sc_d = w.cuda(), sc.cuda()
    L.check(L.lib.dsl_pack_dgrad(L.ptr(w_d), L.ptr(sc_d), L.ptr(out), 80, 128, 9, 256, L.stream_ptr()))
    sync()
    ref = torch.zeros(256, 3, 3, 128)
    ref[..., :80] = (w * sc[:, None, None, None]).permute(3, 1, 2, 0)
    assert torch.equal(out.cpu(), ref.bfloat16())


def test_bf16_rounding_bit_patterns(K):
    """fp32 -> bf16 of every kernel's epilogue (common.hpp pack2bf: v_cvt_pk_bf16_f32 on gfx950) against torch's .bfloat16() on the
    cases that separate rounding rules: exact ties up and down (round to nearest EVEN), the carry into the exponent, the largest
    finite value (rounds to inf), denormals, signed zeros, infinities, NaNs (stay NaN), and two million random bit patterns."""
    L, _ = K
    g = torch.Generator().manual_seed(5)
    bits = torch.randint(-2 ** 31, 2 ** 31 - 1, (1 << 21,), generator=g, dtype=torch.int64).to(torch.int32)
    special = torch.tensor([0x3f808000, 0x3f818000, 0x3f807fff, 0x3f808001, 0x3f7f8000, 0x7f7fffff, 0x7f7f8000, 0x00000001, 0x00008000,
                            0x00018000, 0x807fffff, 0x00000000, -0x80000000, 0x7f800000, -0x00800000, 0x7fc00000, 0x7f800001, -0x00000001,
                            0x477fe000, 0x477ff000, 0x3effffff, 0x3f800000, 0x3f80ffff, 0x00808000], dtype=torch.int64).to(torch.int32)
    x = torch.cat([special, bits]).view(torch.float32)
    y = torch.empty(x.numel(), dtype=torch.bfloat16, device='cuda')
    L.check(L.lib.dsl_cast_bf16(L.ptr(x.cuda()), L.ptr(y), x.numel(), L.stream_ptr()), 'dsl_cast_bf16')
    got, want = y.cpu().view(torch.int16), x.bfloat16().view(torch.int16)
    nan = torch.isnan(x)
    assert torch.equal(got[~nan], want[~nan])
    assert torch.isnan(y.cpu()[nan].float()).all()


@pytest.mark.parametrize('force', [0, 1, 2, 3, 4, 5, 6, 7, 8])
@pytest.mark.parametrize('flavour', ['bn_relu', 'bias', 'plain', 'mask_last'])
def test_conv_in_register_epilogue_equals_staged_epilogue(K, force, flavour):
    """Launches without an addend take conv_tile_epilogue's in-register path (scale / bias / ReLU / rounding in the accumulator
    registers, one bf16 staging round, a last-applied ReLU mask as an AND on the staged words).  The same launch with the test hook
    CONV_EPI_STAGED takes the general staged fp32 path.  Both must produce the same BITS (the sign of an exact zero aside), and
    both must equal the fp32 reference rounded to bf16 to within one bf16 step."""
    L, ops = K
    N, Ci, Co, H, W, k = 2, 128, 256, 24, 40, 3
    bco = {1: 256, 2: 256, 3: 128, 4: 128, 5: 64, 6: 128, 7: 64, 8: 64}.get(force)
    if bco and Co % bco:
        pytest.skip('tile does not divide Cout')
    g = torch.Generator().manual_seed(11 + force)
    x, w = rnd(N, Ci, H, W, g=g), rnd(Co, Ci, k, k, g=g, scale=1 / math.sqrt(Ci * k * k))
    scale, bias = torch.rand(Co, generator=g) + 0.5, torch.randn(Co, generator=g)
    msk = rnd(N, Co, H, W, g=g)
    ref = F.conv2d(x, w, None, 1, 1)
    kw = dict(flags=force << 8)
    if flavour == 'bn_relu':
        ref = F.relu(ref * scale[None, :, None, None] + bias[None, :, None, None])
        kw = dict(flags=L.CONV_RELU_OUT | (force << 8), scale=scale.cuda(), bias=bias.cuda())
    elif flavour == 'bias':
        ref = ref + bias[None, :, None, None]
        kw = dict(flags=force << 8, bias=bias.cuda())
    elif flavour == 'mask_last':
        ref = ref * (msk > 0)
        kw = dict(flags=L.CONV_MASK_LAST | (force << 8), mask=nhwc(msk), ldm=Co)
    outs = []
    for staged in (False, True):
        y = torch.empty(N, H, W, Co, dtype=torch.bfloat16, device='cuda')
        kw2 = dict(kw, flags=kw['flags'] | (L.CONV_EPI_STAGED if staged else 0))
        ops.conv2d(nhwc(x), pack_w(w, Co), y, n=N, grid=[(H, W)], src_hw=[(H, W)], dst_hw=[(H, W)], cs=Ci, cd=Co, cd_pad=Co, ldd=Co,
                   kh=k, kw=k, stride=1, pad=1, **kw2)
        sync()
        outs.append(y.cpu())
    a, b = outs[0].view(torch.int16), outs[1].view(torch.int16)
    same = (a == b) | (((a & 0x7fff) == 0) & ((b & 0x7fff) == 0))          # +0 / -0 aside
    assert bool(same.all()), int((~same).sum())
    got = from_nhwc(outs[0].cuda())
    assert (got - bf(ref)).abs().max() <= 2 ** -7 * ref.abs().max()


@pytest.mark.parametrize('force', [0, 1, 2, 3, 4, 5, 6, 7, 8])
@pytest.mark.parametrize('flavour', ['add', 'bn_add_relu', 'add_mask_last', 'add_wide_rows', 'add_upsample'])
def test_conv_addend_epilogue_equals_staged_epilogue(K, force, flavour):
    """Launches WITH an addend on the destination's own pixel grid (the bottlenecks' residual adds forward and backward, the FPN's and
    the head's gradient sums) take conv_tile_epilogue's addend path: the addend tile is DMA'd into LDS (swizzled dense bf16 rows), added
    in the accumulator registers (scale, bias, + addend, ReLU, one rounding) and the rounded tile leaves as 16-byte stores; a ReLU
    mask applied last is an AND on the rounded words.  Same operations in the same order as the staged fp32 path (test hook
    CONV_EPI_STAGED) => the same BITS, the sign of an exact zero aside; ragged pixel tiles, a padded channel tail (80 of 128 weight
    rows) and row strides wider than the channel count included."""
    L, ops = K
    N, Ci, H, W, k = 2, 128, 23, 37, 3                     # 1 702 pixels: ragged for every pixel tile
    if flavour == 'add_upsample':                          # the FPN lateral's epilogue: the addend lives on the half-size grid (fpn.py:163-172)
        H, W = 24, 38
    Co, cd_pad = (80, 128) if flavour == 'add_wide_rows' else (256, 256)
    bco = {1: 256, 2: 256, 3: 128, 4: 128, 5: 64, 6: 128, 7: 64, 8: 64}.get(force)
    if bco and cd_pad % bco:
        pytest.skip('tile does not divide Cout')
    ldd, lda, ldm = (Co, Co, Co) if flavour != 'add_wide_rows' else (96, 128, 88)
    g = torch.Generator().manual_seed(23 + force)
    x, w = rnd(N, Ci, H, W, g=g), rnd(Co, Ci, k, k, g=g, scale=1 / math.sqrt(Ci * k * k))
    scale, bias = torch.rand(Co, generator=g) + 0.5, torch.randn(Co, generator=g)
    msk, add = rnd(N, Co, H, W, g=g), rnd(N, Co, H, W, g=g)

    def rows(t, ld):          # NCHW fp32 -> [N][H][W][ld] bf16 on the device, the columns beyond the channels poisoned
        o = torch.full((N, H, W, ld), 777.0, dtype=torch.bfloat16, device='cuda')
        o[..., :t.shape[1]] = nhwc(t)
        return o
    ref = F.conv2d(x, w, None, 1, 1)
    kw = dict(flags=force << 8, addend=rows(add, lda), lda=lda)
    if flavour == 'add_upsample':
        small = rnd(N, Co, H // 2, W // 2, g=g)
        ref = ref + bias[None, :, None, None] + F.interpolate(small, size=(H, W), mode='nearest')
        kw = dict(flags=L.CONV_ADD_UPSAMPLE | (force << 8), addend=nhwc(small), lda=Co, add_hw=[(H // 2, W // 2)], bias=bias.cuda())
    elif flavour == 'bn_add_relu':
        ref = F.relu(ref * scale[None, :, None, None] + bias[None, :, None, None] + add)
        kw.update(flags=L.CONV_RELU_OUT | (force << 8), scale=scale.cuda(), bias=bias.cuda())
    elif flavour in ('add_mask_last', 'add_wide_rows'):
        ref = (ref + add) * (msk > 0)
        kw.update(flags=L.CONV_MASK_LAST | (force << 8), mask=rows(msk, ldm), ldm=ldm)
    else:
        ref = ref + add
    outs = []
    for staged in (False, True):
        y = torch.full((N, H, W, ldd), 512.0, dtype=torch.bfloat16, device='cuda')
        kw2 = dict(kw, flags=kw['flags'] | (L.CONV_EPI_STAGED if staged else 0))
        ops.conv2d(nhwc(x), pack_w(w, cd_pad), y, n=N, grid=[(H, W)], src_hw=[(H, W)], dst_hw=[(H, W)], cs=Ci, cd=Co, cd_pad=cd_pad,
                   ldd=ldd, kh=k, kw=k, stride=1, pad=1, **kw2)
        sync()
        assert ldd == Co or float(y[..., Co:].float().sub(512.0).abs().max()) == 0.0         # columns beyond the channels untouched
        outs.append(y[..., :Co].contiguous().cpu())
    a, b = outs[0].view(torch.int16), outs[1].view(torch.int16)
    same = (a == b) | (((a & 0x7fff) == 0) & ((b & 0x7fff) == 0))          # +0 / -0 aside
    assert bool(same.all()), int((~same).sum())
    got = from_nhwc(outs[0].cuda())
    assert (got - bf(ref)).abs().max() <= 2 ** -7 * ref.abs().max()


@pytest.mark.parametrize('planes,stride,shape', [(128, 1, (2, 13, 19)), (128, 1, (1, 25, 42)), (128, 2, (2, 14, 18)), (256, 1, (2, 7, 15)),
                                                 (256, 1, (1, 11, 25)), (256, 2, (2, 6, 13))])
def test_bottleneck_forward_fused(K, planes, stride, shape):
    """dsl_bottleneck_fwd (csrc/bneck.hip): a trained bottleneck - conv1 1x1 [/ stride] -> BN -> ReLU -> conv2 3x3 -> BN -> ReLU -> conv3 1x1
    -> BN -> + identity -> ReLU (resnet.py:262-301) - as ONE launch, against (a) the three dsl_conv2d launches it replaces: a1, a2 and out
    expected bit for bit (same K order per MFMA chain, same rounding points), and (b) fp32 torch with the intermediates rounded to bf16
    where the launches store them.  Ragged tiles (sizes that are no multiple of the 12 x 16 / 5 x 12 pixel tiles), both identity kinds
    (x itself; a separate tensor beside a stride-2 conv1 - a stage's first block)."""
    L, ops = K
    N, H, W = shape
    P, C4 = planes, 4 * planes
    cin = C4 if stride == 1 else 2 * planes
    hin, win = (H, W) if stride == 1 else (2 * H, 2 * W - 1)
    g = torch.Generator().manual_seed(3 * planes + stride + H)
    x = rnd(N, cin, hin, win, g=g)
    w1, w2, w3 = rnd(P, cin, 1, 1, g=g, scale=1 / math.sqrt(cin)), rnd(P, P, 3, 3, g=g, scale=1 / math.sqrt(9 * P)), rnd(C4, P, 1, 1, g=g, scale=1 / math.sqrt(P))
    bns = [(torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g) * 0.3) for c in (P, P, C4)]
    idt = x if stride == 1 else rnd(N, C4, H, W, g=g)
    xd, idd = nhwc(x), (None if stride == 1 else nhwc(idt))
    if stride == 1:
        idd = xd
    p1, p2, p3 = pack_w(w1, P), pack_w(w2, P), pack_w(w3, C4)
    dev = [(s_.cuda(), b_.cuda()) for s_, b_ in bns]
    # (a) the three launches
    a1 = torch.empty(N, H, W, P, dtype=torch.bfloat16, device='cuda')
    a2, out = torch.empty_like(a1), torch.empty(N, H, W, C4, dtype=torch.bfloat16, device='cuda')
    ops.conv2d(xd, p1, a1, n=N, grid=[(H, W)], src_hw=[(hin, win)], dst_hw=[(H, W)], cs=cin, cd=P, cd_pad=P, ldd=P, kh=1, kw=1, stride=stride, pad=0,
               flags=L.CONV_RELU_OUT, scale=dev[0][0], bias=dev[0][1])
    ops.conv2d(a1, p2, a2, n=N, grid=[(H, W)], src_hw=[(H, W)], dst_hw=[(H, W)], cs=P, cd=P, cd_pad=P, ldd=P, kh=3, kw=3, stride=1, pad=1,
               flags=L.CONV_RELU_OUT, scale=dev[1][0], bias=dev[1][1])
    ops.conv2d(a2, p3, out, n=N, grid=[(H, W)], src_hw=[(H, W)], dst_hw=[(H, W)], cs=P, cd=C4, cd_pad=C4, ldd=C4, kh=1, kw=1, stride=1, pad=0,
               flags=L.CONV_RELU_OUT, scale=dev[2][0], bias=dev[2][1], addend=idd, lda=C4)
    # (b) one launch (poisoned outputs: every element must be written)
    f1 = torch.full_like(a1, 7.0)
    f2, fo = torch.full_like(a2, 7.0), torch.full_like(out, 7.0)
    ops.bottleneck_fwd(xd, p1, p2, p3, idd, (dev[0][0].data_ptr(), dev[0][1].data_ptr()), (dev[1][0].data_ptr(), dev[1][1].data_ptr()),
                       (dev[2][0].data_ptr(), dev[2][1].data_ptr()), f1, f2, fo, n=N, hin=hin, win=win, h=H, w=W, planes=P, cin=cin, stride=stride)
    sync()
    for name, got, want in (('a1', f1, a1), ('a2', f2, a2), ('out', fo, out)):
        ga, wa = got.cpu().view(torch.int16), want.cpu().view(torch.int16)
        same = (ga == wa) | (((ga & 0x7fff) == 0) & ((wa & 0x7fff) == 0))
        assert bool(same.all()), (name, int((~same).sum()), float((got.float() - want.float()).abs().max()))
    # fp32 reference, intermediates rounded where they are stored
    bnf = lambda t, k_: t * bns[k_][0][None, :, None, None] + bns[k_][1][None, :, None, None]
    r1 = bf(F.relu(bnf(F.conv2d(x, w1, None, stride, 0), 0)))
    r2 = bf(F.relu(bnf(F.conv2d(r1, w2, None, 1, 1), 1)))
    ro = F.relu(bnf(F.conv2d(r2, w3, None, 1, 0), 2) + idt)
    go = from_nhwc(fo)
    assert (go - bf(ro)).abs().max() <= 2 ** -6 * ro.abs().max()


# ------------------------------------------------------------------------------------------------
# Kernel-level checks at the benchmark's OWN sizes (round-3 review item 10: the weight- and data-gradient kernel tests above
# stop at 25 x 42 pixels, the whole-net gradient tests are relative to a bf16 noise model).  Inputs are bf16-representable, the
# reference is fp32 torch on the CPU, the bar is the one test_conv_forward uses: |err| <= 2^-7 * max|ref| (bf16 outputs carry
# 2^-9 of their own magnitude, the fp32 sums of up to 44 800 x 256 exact products the rest).
FULL_LEVELS = [(100, 168), (50, 84), (25, 42), (13, 21), (7, 11)]        # the five FPN levels of an 800 x 1344 image


def test_full_size_tower_layer_wgrad_and_dgrad(K):
    """One FCOS tower layer (3x3 256 -> 256 over the five levels of two 800 x 1344 images = 44 800 locations): the weight gradient
    (direct 256 x 256 tiles on 72 workgroups, as the step launches the towers' group, and the library's own choice) and the
    data gradient against fp32 autograd."""
    L, ops = K
    g = torch.Generator().manual_seed(31)
    N = 2
    xs = [rnd(N, 256, h, w, g=g) for h, w in FULL_LEVELS]
    dys = [rnd(N, 256, h, w, g=g, scale=0.25) for h, w in FULL_LEVELS]
    w = rnd(256, 256, 3, 3, g=g, scale=1 / 48.0)
    # fp32 reference: torch's own convolution autograd in fp32 ON THE DEVICE (an independent implementation - MIOpen / rocBLAS, true
    # fp32 - of 317 GFLOP; the same sums on the host's cores take minutes)
    wd = w.cuda().requires_grad_()
    xg = [x.cuda().requires_grad_() for x in xs]
    sum((F.conv2d(x, wd, None, 1, 1) * dy.cuda()).sum() for x, dy in zip(xg, dys)).backward()
    ref_w = wd.grad.permute(0, 2, 3, 1).cpu()
    ref_x = torch.cat([x.grad.permute(0, 2, 3, 1).reshape(-1, 256) for x in xg]).cpu()
    X, DY = _multiseg(xs), _multiseg(dys)
    M = X.shape[0]
    assert M == 44800
    for slots in (72, 0):
        dw = torch.full((256, 3, 3, 256), float('nan'), dtype=torch.float32, device='cuda')
        d = ops.wgrad_desc(DY, X, dw, n=N, grid=FULL_LEVELS, src_hw=FULL_LEVELS, cs=256, cy=256, cd=256, kh=3, kw=3, stride=1, pad=1,
                           slots=slots)
        L.check(L.lib.dsl_conv2d_wgrad(C.byref(d), L.stream_ptr()), 'dsl_conv2d_wgrad')
        sync()
        err = float((dw.cpu() - ref_w).abs().max())
        assert err <= 2.0 ** -7 * float(ref_w.abs().max()), (slots, err, float(ref_w.abs().max()))
    dx = torch.empty(M, 256, dtype=torch.bfloat16, device='cuda')
    ops.conv2d(DY, pack_w_dgrad(w, 256), dx, n=N, grid=FULL_LEVELS, src_hw=FULL_LEVELS, dst_hw=FULL_LEVELS, cs=256, cd=256,
               cd_pad=256, ldd=256, kh=3, kw=3, stride=1, pad=1, mode=1)
    sync()
    err = float((dx.float().cpu() - ref_x).abs().max())
    assert err <= 2.0 ** -7 * float(ref_x.abs().max()), (err, float(ref_x.abs().max()))


def test_full_size_layer3_block_wgrad_and_dgrad(K):
    """The three convolutions of a layer3 bottleneck at the benchmark's size (2 x 50 x 84 = 8 400 pixels; 1x1 1024 -> 256, 3x3 256 ->
    256, 1x1 256 -> 1024): weight gradients through the multi launch the step uses (one grid, the planner's split factors and
    schedule table) and the three data gradients, against fp32 autograd."""
    L, ops = K
    g = torch.Generator().manual_seed(32)
    N, H, W = 2, 50, 84
    shapes = [(1024, 256, 1, 0), (256, 256, 3, 1), (256, 1024, 1, 0)]            # (cin, cout, k, pad)
    subs, refs, outs, dgr = [], [], [], []
    for ci, co, k, p in shapes:
        x = rnd(N, ci, H, W, g=g)
        w = rnd(co, ci, k, k, g=g, scale=1 / math.sqrt(ci * k * k))
        dy = rnd(N, co, H, W, g=g, scale=0.25)
        xd, wd = x.cuda().requires_grad_(), w.cuda().requires_grad_()          # fp32 reference on the device (see above)
        (F.conv2d(xd, wd, None, 1, p) * dy.cuda()).sum().backward()
        dw = torch.full((co, k, k, ci), float('nan'), dtype=torch.float32, device='cuda')
        subs.append([ops.wgrad_desc(nhwc(dy), nhwc(x), dw, n=N, grid=[(H, W)], src_hw=[(H, W)], cs=ci, cy=co, cd=co, kh=k,
                                    kw=k, stride=1, pad=p)])
        refs.append(wd.grad.permute(0, 2, 3, 1).cpu())
        outs.append(dw)
        dgr.append((dy, w, xd.grad.cpu(), ci, co, k, p))
    assert {L.lib.dsl_wgrad_multi_config(C.byref(s[0])) for s in subs} == {1}
    plan = ops.WgradMulti(subs)
    plan.run()
    sync()
    for ref, dw in zip(refs, outs):
        err = float((dw.cpu() - ref).abs().max())
        assert err <= 2.0 ** -7 * float(ref.abs().max()), (tuple(ref.shape), err, float(ref.abs().max()))
    for dy, w, ref, ci, co, k, p in dgr:
        dx = torch.empty(N, H, W, ci, dtype=torch.bfloat16, device='cuda')
        ops.conv2d(nhwc(dy), pack_w_dgrad(w, co), dx, n=N, grid=[(H, W)], src_hw=[(H, W)], dst_hw=[(H, W)], cs=co, cd=ci, cd_pad=ci,
                   ldd=ci, kh=k, kw=k, stride=1, pad=p, mode=1)
        sync()
        err = float((from_nhwc(dx) - ref).abs().max())
        assert err <= 2.0 ** -7 * float(ref.abs().max()), ((ci, co, k), err, float(ref.abs().max()))


@pytest.mark.parametrize('shape', [(2, 128, 128, 20, 28), (1, 64, 128, 13, 21), (3, 256, 256, 10, 14), (2, 128, 64, 7, 9)])
def test_dgrad_3x3_stride2_as_four_parity_class_convolutions(K, shape):
    """ops.dgrad_s2_descs: the data gradient of a 3x3 / 2 / pad 1 convolution as four stride-1 transposed convolutions over the dY
    grid (1x1 for the even-even pixels, 2x2 for the rest) scattered with output stride 2 == autograd's input gradient, with the ReLU
    mask of the consumer applied last; even and odd image sizes; and == the general strided gather it replaces."""
    L, ops = K
    N, Ci, Co, H, W = shape
    g = torch.Generator().manual_seed(H * W + Ci)
    x = rnd(N, Ci, H, W, g=g).requires_grad_()
    w = rnd(Co, Ci, 3, 3, g=g, scale=1 / math.sqrt(Co * 9))
    y = F.conv2d(x, w, None, 2, 1)
    Ho, Wo = y.shape[2:]
    dy = rnd(N, Co, Ho, Wo, g=g)
    y.backward(dy)
    msk = rnd(N, Ci, H, W, g=g)
    ref = x.grad * (msk > 0)
    cy = (Co + 63) // 64 * 64
    dyp = torch.zeros(N, Ho, Wo, cy)
    dyp[..., :Co] = dy.permute(0, 2, 3, 1)
    dyp = dyp.bfloat16().cuda()
    packs = {}
    for py in (0, 1):
        for px in (0, 1):
            k, pad, taps = ops.s2_class(py, px)
            pk = torch.zeros(Ci, k * k, cy)
            for t, src in enumerate(taps):
                if src >= 0:
                    pk[:, t, :Co] = w[:, :, src // 3, src % 3].t()
            packs[(py, px)] = pk.bfloat16().cuda()
    dx = torch.full((N, H, W, Ci), float('nan'), dtype=torch.bfloat16, device='cuda')
    m = nhwc(msk)
    descs = ops.dgrad_s2_descs(dyp, {k_: v.data_ptr() for k_, v in packs.items()}, dx, n=N, dy_hw=(Ho, Wo), dst_hw=(H, W), cs=cy, cd=Ci,
                               mask=m, ldm=Ci, flags=L.CONV_MASK_LAST)
    assert len(descs) == 4
    for d in descs:
        L.check(L.lib.dsl_conv2d(C.byref(d), L.stream_ptr()), 'dsl_conv2d')
    sync()
    got = from_nhwc(dx)
    assert torch.isfinite(got).all()                       # every pixel belongs to exactly one class
    assert torch.allclose(got, ref, rtol=1e-2, atol=2e-2), (got - ref).abs().max()
    # the strided gather it replaces (conv_glds): same result up to the bf16 rounding of differently ordered fp32 sums
    dx2 = torch.empty(N, H, W, Ci, dtype=torch.bfloat16, device='cuda')
    ops.conv2d(dyp, pack_w_dgrad(w, cy), dx2, n=N, grid=[(H, W)], src_hw=[(Ho, Wo)], dst_hw=[(H, W)], cs=cy, cd=Ci, cd_pad=Ci, ldd=Ci,
               kh=3, kw=3, stride=2, pad=1, mode=1, mask=m, ldm=Ci, flags=L.CONV_MASK_LAST)
    sync()
    assert torch.allclose(from_nhwc(dx2), got, rtol=1e-2, atol=2e-2)

