"""Row f3, the unlabeled stream: RandomAugmentBBox_Fast(aug_type='affine') and UBAug (configs/fcos_semi/RLA_*.py:93-94) rendered
by dsl_image_aug.  Every pass is compared with oracle/datapath_oracle.py bit for bit (the colour / grayscale / blur restatements
are themselves pinned to Pillow's outputs: tests/test_datapath_cpu.py); the whole pipeline is replayed on the oracle from the
parameters the host drew."""
import random

import numpy as np
import pytest
import torch

from oracle import datapath_oracle as DO

pytestmark = pytest.mark.gpu
SIZES = [(61, 83), (96, 128), (40, 37)]
HC, WC = 96, 128


def _canvases(seed):
    rng = np.random.RandomState(seed)
    can = rng.randint(0, 256, (len(SIZES), HC, WC, 3)).astype(np.uint8)      # junk outside the images: it must be copied through
    for i, (h, w) in enumerate(SIZES):
        # smooth content + noise, so that the blur and the bilinear warp see structure
        yy, xx = np.mgrid[0:h, 0:w]
        base = np.stack([127 + 100 * np.sin(xx / 7.0 + i), 127 + 100 * np.cos(yy / 5.0), (xx * 3 + yy * 2) % 256], -1)
        can[i, :h, :w] = np.clip(base + rng.randint(-30, 30, (h, w, 3)), 0, 255).astype(np.uint8)
    return can


def _run(can, items_fn):
    from dsl_amd import _lib as L
    n = can.shape[0]
    its = (L.AugItem * n)()
    need_mean = 0
    for i, (h, w) in enumerate(SIZES):
        its[i].h, its[i].w = h, w
        items_fn(i, its[i])
        need_mean |= int(its[i].kind == L.AUG_CONTRAST) | 2 * int(its[i].kind in (L.AUG_AUTOCONTRAST, L.AUG_EQUALIZE))
    src = torch.from_numpy(can).cuda()
    dst = torch.zeros_like(src)
    at = torch.frombuffer(bytearray(bytes(its)), dtype=torch.uint8).cuda()
    sums = torch.full((int(L.lib.dsl_image_aug_scratch_bytes(n)) // 8,), -1, dtype=torch.int64, device='cuda')      # junk: the call clears what it uses
    L.check(L.lib.dsl_image_aug(L.ptr(at), n, L.ptr(src), L.ptr(dst), HC, WC, L.ptr(sums), need_mean, L.stream_ptr()), 'dsl_image_aug')
    torch.cuda.synchronize()
    return dst.cpu().numpy()


def _check(can, got, fn):
    for i, (h, w) in enumerate(SIZES):
        want = can[i].copy()
        want[:h, :w] = fn(i, can[i, :h, :w])
        bad = np.argwhere(got[i] != want)
        assert bad.shape[0] == 0, (i, bad[:5], got[i][tuple(bad[0])], want[tuple(bad[0])])


@pytest.mark.parametrize('name', ['brightness', 'contrast', 'saturation', 'hue', 'gray'])
def test_colour_passes_are_pillow_arithmetic(name):
    from dsl_amd import _lib as L
    kind = dict(brightness=L.AUG_BRIGHTNESS, contrast=L.AUG_CONTRAST, saturation=L.AUG_SATURATION, hue=L.AUG_HUE, gray=L.AUG_GRAY)[name]
    fn = dict(brightness=DO.adjust_brightness, contrast=DO.adjust_contrast, saturation=DO.adjust_saturation, hue=DO.adjust_hue,
              gray=lambda a, f: DO.to_grayscale3(a))[name]
    for trial, facs in enumerate([(0.6, 1.4, 1.0), (0.873, 1.2291, 0.61234)] if name != 'hue' else [(-0.1, 0.1, 0.0), (0.0371, -0.0642, 0.0999)]):
        can = _canvases(trial)

        def item(i, it):
            it.kind = kind
            it.f[0] = float(int(facs[i] * 255)) if name == 'hue' else facs[i]
        got = _run(can, item)
        _check(can, got, lambda i, a: fn(a, facs[i]))


@pytest.mark.parametrize('name', ['autocontrast', 'equalize', 'solarize', 'posterize', 'sharpness'])
def test_randaug_ops_are_pillow_arithmetic(name):
    """The no-box colour branch's histogram / filter ops (autoaug_fast.py:219-224, 244-250, 371-372, 407), every level the policy draws,
    against the restatement that tests/golden/randaug_pil.npz pins to Pillow."""
    from dsl_amd import _lib as L
    kind = dict(autocontrast=L.AUG_AUTOCONTRAST, equalize=L.AUG_EQUALIZE, solarize=L.AUG_SOLARIZE, posterize=L.AUG_POSTERIZE,
                sharpness=L.AUG_SHARPNESS)[name]
    fn = dict(autocontrast=lambda a, v: DO.autocontrast(a), equalize=lambda a, v: DO.equalize(a), solarize=DO.solarize,
              posterize=DO.posterize, sharpness=DO.adjust_sharpness)[name]
    op = name.capitalize()
    for trial, levels in enumerate([(1, 5, 9), (2, 6, 8), (3, 4, 7)]):
        can = _canvases(20 + trial)
        if trial == 1:                                   # narrow / degenerate histograms: one band constant, one with two values
            for i, (h, w) in enumerate(SIZES):
                can[i, :h, :w, 0] = 40 + can[i, :h, :w, 0] // 3
                can[i, :h, :w, 1] = 77
                can[i, :h, :w, 2] = 200 + (can[i, :h, :w, 2] & 1)
        vals = [DO.randaug_level(op, lv) for lv in levels]

        def item(i, it):
            it.kind, it.f[0] = kind, float(vals[i])
        got = _run(can, item)
        _check(can, got, lambda i, a: fn(a, vals[i]))


def test_mixed_stat_passes_in_one_launch():
    """One launch whose images take a CONTRAST, an EQUALIZE and an AUTOCONTRAST pass: the luma sums and the histograms share the scratch."""
    from dsl_amd import _lib as L
    can = _canvases(31)
    kinds = (L.AUG_CONTRAST, L.AUG_EQUALIZE, L.AUG_AUTOCONTRAST)

    def item(i, it):
        it.kind, it.f[0] = kinds[i], 1.3
    got = _run(can, item)
    _check(can, got, lambda i, a: (lambda a: DO.adjust_contrast(a, 1.3), DO.equalize, DO.autocontrast)[i](a))


def test_hsv_round_trip_every_colour_class():
    """Hue shifts on a canvas that sweeps the six hue sectors, grays and saturated corners (Convert.c's branches)."""
    from dsl_amd import _lib as L
    rng = np.random.RandomState(3)
    can = _canvases(5)
    pal = np.array([[255, 0, 0], [0, 255, 0], [0, 0, 255], [255, 255, 0], [0, 255, 255], [255, 0, 255], [0, 0, 0], [255, 255, 255],
                    [128, 128, 128], [200, 100, 100], [100, 200, 100], [100, 100, 200], [1, 0, 0], [254, 255, 255]], np.uint8)
    for i, (h, w) in enumerate(SIZES):
        can[i, :h, :w] = pal[rng.randint(0, len(pal), (h, w))]
    shifts = (13, -25, 127)

    def item(i, it):
        it.kind, it.f[0] = L.AUG_HUE, float(shifts[i])
    got = _run(can, item)
    _check(can, got, lambda i, a: DO.adjust_hue(a, shifts[i] / 255.0 + (1e-6 if shifts[i] > 0 else -1e-6)))


@pytest.mark.parametrize('sigma', [0.1, 0.45, 1.0, 1.37, 2.0])
def test_gaussian_blur_is_pillow_box_blur(sigma):
    """ImageFilter.GaussianBlur(sigma) = 3 horizontal + 3 vertical extended-box passes (BoxBlur.c), six launches."""
    from dsl_amd import _lib as L
    from dsl_amd.datapath import blur_box_radius
    fr = blur_box_radius(sigma)
    assert fr == float(DO.gauss_box_radius(sigma))
    can = _canvases(7)
    cur = can
    for kind in [L.AUG_BLUR_H] * 3 + [L.AUG_BLUR_V] * 3:
        if fr == 0:
            break

        def item(i, it, kind=kind):
            it.kind, it.f[0] = kind, fr
        cur = _run(cur, item)
    _check(can, cur, lambda i, a: DO.gaussian_blur(a, sigma))


@pytest.mark.parametrize('order', [0, 1])
@pytest.mark.parametrize('kind,value', [('translate_x', 0.083), ('translate_y', -0.1), ('rotate', 27.5), ('rotate', -11.0), ('shear', 30.0),
                                        ('shear', -8.0)])
def test_affine_whole_image(kind, value, order):
    from dsl_amd import _lib as L
    from dsl_amd.datapath import _affine_forward
    can = _canvases(11)
    Ms = [_affine_forward(kind, value, w, h) for (h, w) in SIZES]
    for (h, w), M in zip(SIZES, Ms):
        assert np.allclose(M, DO.affine_matrix(kind, value, w, h))

    def item(i, it):
        h, w = SIZES[i]
        Mi = np.linalg.inv(Ms[i])
        it.kind, it.order, it.cval = L.AUG_AFFINE, order, 125
        for q in range(6):
            it.m[q] = float(Mi[q // 3, q % 3])
        it.roi[0], it.roi[1], it.roi[2], it.roi[3] = 0, 0, w, h
    got = _run(can, item)
    _check(can, got, lambda i, a: DO.warp_affine_u8(a, Ms[i], order, 125))


def test_affine_inside_one_box():
    """bbox_affine_transform (semi_aug.py:431-452): the weak transform acts on the crop of one box, the rest of the image stays."""
    from dsl_amd import _lib as L
    from dsl_amd.datapath import _affine_forward
    can = _canvases(13)
    boxes = [(10, 5, 50, 44), (0, 0, 128, 96), (20, 30, 37, 40)]
    Ms = [_affine_forward('rotate', 9.0, x1 - x0, y1 - y0) for (x0, y0, x1, y1) in boxes]

    def item(i, it):
        Mi = np.linalg.inv(Ms[i])
        it.kind, it.order, it.cval = L.AUG_AFFINE, 1, 125
        for q in range(6):
            it.m[q] = float(Mi[q // 3, q % 3])
        for q in range(4):
            it.roi[q] = boxes[i][q]
    got = _run(can, item)

    def want(i, a):
        x0, y0, x1, y1 = boxes[i]
        a = a.copy()
        a[y0:y1, x0:x1] = DO.warp_affine_u8(a[y0:y1, x0:x1], Ms[i], 1, 125)
        return a
    _check(can, got, want)


def test_random_erasing_rectangles():
    """RandomErasing(value='random'): inside the rectangles N(0, 1) noise through mul(255).byte() - truncation toward zero and the
    mod-256 wrap make every byte value nearly equally likely, and the three channels independent; outside nothing moves; the
    stream is a function of the seed."""
    from dsl_amd import _lib as L
    can = _canvases(17)
    rects = [[(5, 5, 40, 30), (30, 20, 70, 55)], [(0, 0, 128, 96)], [(3, 4, 20, 30), (0, 0, 0, 0), (10, 10, 30, 39)]]

    def item(i, it, seed=1234):
        it.kind, it.seed = L.AUG_ERASE, seed + i
        for q, rc in enumerate(rects[i]):
            for e in range(4):
                it.rect[q][e] = rc[e]
    got = _run(can, item)
    again = _run(can, item)
    other = _run(can, lambda i, it: item(i, it, 99))
    assert np.array_equal(got, again)
    for i, (h, w) in enumerate(SIZES):
        m = np.zeros((HC, WC), bool)
        for (x0, y0, x1, y1) in rects[i]:
            m[y0:y1, x0:x1] = True
        m[h:], m[:, w:] = False, False
        assert np.array_equal(got[i][~m], can[i][~m])
        v = got[i][m].astype(np.float64)
        assert (got[i][m] != other[i][m]).mean() > 0.98
        if v.size > 3000:
            # |z| * 255 mod 256 of a normal: flat to within a few percent over the byte range
            hist = np.bincount(got[i][m].ravel(), minlength=256) / v.size
            assert abs(v.mean() - 127.5) < 6 and abs(v.std() - 73.9) < 4 and hist.max() < 3.0 / 256
            c = np.corrcoef(got[i][m].reshape(-1, 3).T.astype(np.float64))
            assert abs(c[0, 1]) < 0.05 and abs(c[0, 2]) < 0.05
    # the value mapping itself: z = -0.5 -> trunc(-127.5) = -127 -> 129; z = 1.2 -> 306 -> 50 (the restated rule)
    assert DO.erase_value([-0.5, 1.2, 0.0]).tolist() == [129, 50, 0]


NORM = dict(mean=[102.9801, 115.9465, 122.7717], std=[1.0, 1.0, 1.0], to_rgb=False)
UNLABELED = [dict(type='LoadImageFromFile'), dict(type='LoadAnnotations', with_bbox=True),
             dict(type='Resize', img_scale=[(300, 160), (300, 200)], multiscale_mode='value', keep_ratio=True),
             dict(type='PatchShuffle', ratio=0.5, ranges=[0.0, 1.0], mode=['flip', 'flop']),
             dict(type='RandomFlip', flip_ratio=0.5), dict(type='RandomAugmentBBox_Fast', aug_type='affine'), dict(type='UBAug'),
             dict(type='Normalize', **NORM), dict(type='Pad', size_divisor=32), dict(type='DefaultFormatBundle'),
             dict(type='Collect', keys=['img', 'gt_bboxes', 'gt_labels', 'gt_bboxes_ignore'])]


def _replay(img, passes):
    from dsl_amd import _lib as L
    erased = np.zeros(img.shape[:2], bool)
    for p in passes:
        k = p['kind']
        if k == L.AUG_AFFINE:
            x0, y0, x1, y1 = p['roi']
            img = img.copy()
            img[y0:y1, x0:x1] = DO.warp_affine_u8(img[y0:y1, x0:x1], p['M'], p['order'], 125)
        elif k == L.AUG_BRIGHTNESS:
            img = DO.adjust_brightness(img, p['f'])
        elif k == L.AUG_CONTRAST:
            img = DO.adjust_contrast(img, p['f'])
        elif k == L.AUG_SATURATION:
            img = DO.adjust_saturation(img, p['f'])
        elif k == L.AUG_HUE:
            img = DO.adjust_hue(img, (p['f'] + (0.5 if p['f'] > 0 else -0.5 if p['f'] < 0 else 0)) / 255.0)
        elif k == L.AUG_GRAY:
            img = DO.to_grayscale3(img)
        elif k == L.AUG_AUTOCONTRAST:
            img = DO.autocontrast(img)
        elif k == L.AUG_EQUALIZE:
            img = DO.equalize(img)
        elif k == L.AUG_SOLARIZE:
            img = DO.solarize(img, int(p['f']))
        elif k == L.AUG_POSTERIZE:
            img = DO.posterize(img, int(p['f']))
        elif k == L.AUG_SHARPNESS:
            img = DO.adjust_sharpness(img, p['f'])
        elif k == L.AUG_BLUR_H:
            img = DO.box_blur_pass_h(img, np.float32(p['f']))
        elif k == L.AUG_BLUR_V:
            img = np.ascontiguousarray(DO.box_blur_pass_h(img.transpose(1, 0, 2), np.float32(p['f'])).transpose(1, 0, 2))
        elif k == L.AUG_ERASE:
            for (x0, y0, x1, y1) in p['rects']:
                erased[y0:y1, x0:x1] = True
    return img, erased


@pytest.mark.parametrize('seed', [0, 1, 2, 3, 4, 5])
def test_unlabeled_pipeline_replayed_on_the_oracle(seed):
    """The DSL config's unlabel_train_pipeline end to end: Resize -> PatchShuffle -> RandomFlip -> RandomAugmentBBox_Fast -> UBAug ->
    Normalize -> Pad.  The drawn passes are replayed on the CPU restatement; outside the erased rectangles the batch is equal
    bit for bit, and the boxes are the corner hulls of the drawn affine map."""
    from test_datapath_gpu import _samples
    from dsl_amd import _lib as L
    from dsl_amd.datapath import GpuBatchPipeline
    rng = np.random.RandomState(seed)
    samples = _samples(rng, [(120, 160), (150, 100), (97, 131), (64, 64)])
    samples[3]['gt_bboxes'], samples[3]['gt_labels'] = np.zeros((0, 4), np.float32), np.zeros((0,), np.int64)    # the colour branch
    np.random.seed(300 + seed)
    random.seed(400 + seed)
    pipe = GpuBatchPipeline(UNLABELED)
    batch = pipe(samples)
    torch.cuda.synchronize()
    got = batch['img'].cpu().numpy()
    n_pass = 0
    for i, (s, m) in enumerate(zip(samples, batch['img_metas'])):
        img = DO.resize_bilinear_u8(s['img'], DO.rescale_size((s['img'].shape[1], s['img'].shape[0]), (300, 160) if m['scale_idx'] == 0 else (300, 200)))
        b = DO.resize_bboxes(s['gt_bboxes'].copy(), m['scale_factor'], m['img_shape'][:2])
        lab = s['gt_labels'].astype(np.int64)
        h, w = img.shape[:2]
        if m['PS']:
            img = DO.patch_shuffle_image(img, m['PS_place'], m['PS_mode'])
            b, lab = DO.patch_shuffle_boxes(b, lab, h, w, m['PS_place'], m['PS_mode'])
        if m['flip']:
            img = img[:, ::-1]
            b = DO.bbox_flip_horizontal(b, w)
        passes = pipe.last_passes[i]
        n_pass += len(passes)
        img, erased = _replay(np.ascontiguousarray(img), passes)
        whole = [p for p in passes if p['kind'] == L.AUG_AFFINE and tuple(p['roi']) == (0, 0, w, h)]
        b = np.asarray(b, np.float32).reshape(-1, 4)
        if whole:
            b, keep = DO.affine_boxes(b, whole[0]['M'], w, h)
            lab = lab[keep]
        elif len(b):
            b, keep = DO.affine_boxes(b, np.eye(3), w, h)
            lab = lab[keep]
        assert np.array_equal(batch['gt_bboxes'][i].numpy(), b) and np.array_equal(batch['gt_labels'][i].numpy(), lab)
        want = DO.imnormalize(img, NORM['mean'], NORM['std'], NORM['to_rgb']).transpose(2, 0, 1)
        g = got[i, :, :h, :w]
        assert np.array_equal(g[:, ~erased], want[:, ~erased]), (i, [p['kind'] for p in passes])
        assert (got[i, :, h:] == 0).all() and (got[i, :, :, w:] == 0).all()
        if erased.sum() > 500:
            assert (g[:, erased] != want[:, erased]).mean() > 0.9
    assert n_pass > 0
    assert all(t.skipped_ops == 0 for t in pipe.transforms if hasattr(t, 'skipped_ops'))


def test_colour_branch_every_op_replayed_on_the_oracle():
    """Images WITHOUT boxes through the unlabeled pipeline (semi_aug.py:494-497: the RandAug colour branch): batches are drawn until
    every one of the nine RANDOM_COLOR_POLICY_OPS has been rendered at least once; each batch equals its replay on the Pillow-pinned
    restatement outside the erased rectangles, and no op is skipped."""
    from test_datapath_gpu import _samples
    from dsl_amd import _lib as L
    from dsl_amd.datapath import GpuBatchPipeline
    pipe = GpuBatchPipeline(UNLABELED)
    seen = set()
    for seed in range(12):
        rng = np.random.RandomState(900 + seed)
        samples = _samples(rng, [(120, 160), (150, 100), (97, 131), (64, 64)])
        for s in samples:
            s['gt_bboxes'], s['gt_labels'] = np.zeros((0, 4), np.float32), np.zeros((0,), np.int64)
        np.random.seed(700 + seed)
        random.seed(800 + seed)
        batch = pipe(samples)
        torch.cuda.synchronize()
        got = batch['img'].cpu().numpy()
        for i, (s, m) in enumerate(zip(samples, batch['img_metas'])):
            img = DO.resize_bilinear_u8(s['img'], DO.rescale_size((s['img'].shape[1], s['img'].shape[0]), (300, 160) if m['scale_idx'] == 0 else (300, 200)))
            h, w = img.shape[:2]
            if m['PS']:
                img = DO.patch_shuffle_image(img, m['PS_place'], m['PS_mode'])
            if m['flip']:
                img = img[:, ::-1]
            passes = pipe.last_passes[i]
            seen |= {p.get('op') for p in passes if 'op' in p}
            img, erased = _replay(np.ascontiguousarray(img), passes)
            want = DO.imnormalize(img, NORM['mean'], NORM['std'], NORM['to_rgb']).transpose(2, 0, 1)
            g = got[i, :, :h, :w]
            assert np.array_equal(g[:, ~erased], want[:, ~erased]), (seed, i, [p['kind'] for p in passes])
        if len(seen) == 8:
            break
    assert seen == {'AutoContrast', 'Equalize', 'Solarize', 'Color', 'Contrast', 'Brightness', 'Sharpness', 'Posterize'}, seen
    assert all(t.skipped_ops == 0 for t in pipe.transforms if hasattr(t, 'skipped_ops'))


def test_unlabeled_batch_feeds_the_dsl_iteration():
    """The batch the unlabeled pipeline builds goes through the detector's forward as the DSL hook's unlabeled images do."""
    from util import fcos_model_cfg
    from test_datapath_gpu import _samples
    from dsl_amd import detectors  # noqa: F401
    from dsl_amd.datapath import GpuBatchPipeline
    from dsl_amd.registry import build_detector
    from oracle import fcos_oracle as O
    np.random.seed(5)
    random.seed(5)
    batch = GpuBatchPipeline(UNLABELED)(_samples(np.random.RandomState(1), [(120, 160), (150, 100)]))
    model = build_detector(fcos_model_cfg())
    model.load_state_dict(O.synth_state_dict(0))
    model = model.cuda()
    out = model.train_step(dict(img=batch['img'], img_metas=batch['img_metas'], gt_bboxes=[b.cuda() for b in batch['gt_bboxes']],
                                gt_labels=[l.cuda() for l in batch['gt_labels']]), None)
    assert np.isfinite(float(out['loss'].detach()))
