"""Row f3, CPU side: the oracle's and the product's PatchShuffle against the reference class's own outputs
(tests/golden/patch_shuffle.json, made by tests/golden/make_golden.py ps), the box arithmetic of Resize / RandomFlip / Pad, and
the resize restatement's fixed points."""
import json
import os
import random

import numpy as np
import pytest

from oracle import datapath_oracle as DO

HERE = os.path.dirname(os.path.abspath(__file__))


def cases():
    return json.load(open(os.path.join(HERE, 'golden', 'patch_shuffle.json')))


def _img(c):
    h, w, _ = c['input']['img_shape']
    return np.frombuffer(bytes.fromhex(c['input']['img_hex']), np.uint8).reshape(h, w, 3)


def test_oracle_patch_shuffle_vs_reference_golden():
    n_cut = 0
    for c in cases():
        o = c['output']
        img = _img(c)
        h, w = img.shape[:2]
        gtb = np.array(c['input']['gt_bboxes'], np.float32).reshape(-1, 4)
        gtl = np.array(c['input']['gt_labels'], np.int64)
        igb = np.array(c['input']['gt_bboxes_ignore'], np.float32).reshape(-1, 4)
        if not o['PS']:
            continue
        got_img = DO.patch_shuffle_image(img, o['PS_place'], o['PS_mode'])
        want = np.frombuffer(bytes.fromhex(o['img_hex']), np.uint8).reshape(got_img.shape)
        assert np.array_equal(got_img, want)
        b, l = DO.patch_shuffle_boxes(gtb, gtl, h, w, o['PS_place'], o['PS_mode'])
        assert np.array_equal(np.asarray(b, np.float32).reshape(-1, 4), np.array(o['gt_bboxes'], np.float32).reshape(-1, 4))
        assert np.array_equal(np.asarray(l), np.array(o['gt_labels'], np.int64))
        ib, _ = DO.patch_shuffle_boxes(igb, None, h, w, o['PS_place'], o['PS_mode'])
        assert np.array_equal(np.asarray(ib, np.float32).reshape(-1, 4), np.array(o['gt_bboxes_ignore'], np.float32).reshape(-1, 4))
        n_cut += len(o['gt_bboxes']) > len(gtb)
    assert n_cut >= 3          # boxes across the split (cut in two) occur in the fixture


def test_product_patch_shuffle_draws_and_moves_boxes_like_the_reference():
    """Same seeds -> the same random draws (np.random.rand, random.choice in the reference's order) and the same boxes."""
    from dsl_amd.datapath import PatchShuffle
    for c in cases():
        o = c['output']
        np.random.seed(c['seed'][0])
        random.seed(c['seed'][1])
        h, w, _ = c['input']['img_shape']
        r = dict(img_shape=(h, w, 3), gt_bboxes=np.array(c['input']['gt_bboxes'], np.float32).reshape(-1, 4),
                 gt_labels=np.array(c['input']['gt_labels'], np.int64),
                 gt_bboxes_ignore=np.array(c['input']['gt_bboxes_ignore'], np.float32).reshape(-1, 4),
                 bbox_fields=['gt_bboxes_ignore', 'gt_bboxes'])
        r = PatchShuffle(ratio=c['ratio'], ranges=c['ranges'], mode=['flip', 'flop'])(r)
        assert bool(r['PS']) == o['PS'] and r['PS_mode'] == o['PS_mode']
        if o['PS']:
            assert float(r['PS_place']) == o['PS_place']
        assert np.array_equal(np.asarray(r['gt_bboxes'], np.float32).reshape(-1, 4), np.array(o['gt_bboxes'], np.float32).reshape(-1, 4))
        assert np.array_equal(np.asarray(r['gt_labels']), np.array(o['gt_labels'], np.int64))
        assert np.array_equal(np.asarray(r['gt_bboxes_ignore'], np.float32).reshape(-1, 4),
                              np.array(o['gt_bboxes_ignore'], np.float32).reshape(-1, 4))


def test_rescale_size_and_boxes():
    # mmcv.rescale_size((w, h), (1333, 800)): COCO's 640 x 480 -> 1067 x 800; 500 x 375 -> 1067 x 800; tall images hit the long edge
    assert DO.rescale_size((640, 480), (1333, 800)) == (1067, 800)
    assert DO.rescale_size((480, 640), (1333, 800)) == (800, 1067)
    assert DO.rescale_size((1000, 200), (1333, 800)) == (1333, 267)
    from dsl_amd.datapath import Pad, RandomFlip, Resize, rescale_size
    assert rescale_size((640, 480), (1333, 800)) == (1067, 800)
    r = dict(img_shape=(480, 640, 3), gt_bboxes=np.array([[10., 20., 630., 470.], [0., 0., 640., 480.]], np.float32),
             bbox_fields=['gt_bboxes'], scale=(1333, 800))
    r = Resize(img_scale=[(1333, 640), (1333, 800)], multiscale_mode='value', keep_ratio=True)(r)
    sf = np.array([1067 / 640, 800 / 480] * 2, np.float32)
    assert r['img_shape'] == (800, 1067, 3) and np.array_equal(r['scale_factor'], sf)
    want = DO.resize_bboxes(np.array([[10., 20., 630., 470.], [0., 0., 640., 480.]], np.float32), sf, (800, 1067))
    assert np.array_equal(r['gt_bboxes'], want) and want[1, 2] == 1067 and want[1, 3] == 800
    r['flip'] = True
    r = RandomFlip(flip_ratio=0.5)(r)
    assert np.array_equal(r['gt_bboxes'], DO.bbox_flip_horizontal(want, 1067))
    r = Pad(size_divisor=32)(r)
    assert r['pad_shape'] == (800, 1088, 3) == DO.pad_to_multiple((800, 1067), 32) + (3,)


def test_resize_restatement_fixed_points():
    """Properties the fixed-point bilinear must have whatever the library: identity at scale 1, constants stay constant,
    exact 2x upsampling of a two-level step keeps the levels at the borders, monotone ramps stay monotone."""
    rng = np.random.RandomState(0)
    img = rng.randint(0, 256, (17, 23, 3)).astype(np.uint8)
    assert np.array_equal(DO.resize_bilinear_u8(img, (23, 17)), img)
    const = np.full((9, 11, 3), 137, np.uint8)
    assert np.all(DO.resize_bilinear_u8(const, (40, 31)) == 137)
    ramp = np.repeat(np.arange(0, 250, 10, dtype=np.uint8)[None, :, None], 6, 0).repeat(3, 2)
    up = DO.resize_bilinear_u8(ramp, (80, 13)).astype(int)
    assert np.all(np.diff(up[0, :, 0]) >= 0) and up[0, 0, 0] == 0 and up[0, -1, 0] == 240
    down = DO.resize_bilinear_u8(img, (11, 8))
    assert down.shape == (8, 11, 3) and down.dtype == np.uint8
    assert abs(float(down.mean()) - float(img.mean())) < 12


def test_normalize_formula():
    img = np.arange(2 * 3 * 3, dtype=np.uint8).reshape(2, 3, 3) * 7
    out = DO.imnormalize(img, [123.675, 116.28, 103.53], [58.395, 57.12, 57.375], to_rgb=True)
    r = (np.float32(img[0, 0, 2]) - np.float32(123.675)) * np.float32(1.0 / np.float64(np.float32(58.395)))
    assert out.dtype == np.float32 and out[0, 0, 0] == r


def test_ubaug_arithmetic_pinned_to_pillow(golden):
    """The numpy restatements of Pillow's enhancement / HSV / box-blur arithmetic (what UBAug runs under torchvision) against
    Pillow's own outputs (tests/golden/ubaug_pil.npz): bit for bit."""
    from oracle import datapath_oracle as D
    d = golden('ubaug_pil.npz')
    rgb = d['rgb']
    for name, fn in (('brightness', D.adjust_brightness), ('contrast', D.adjust_contrast), ('saturation', D.adjust_saturation),
                     ('hue', D.adjust_hue), ('blur', D.gaussian_blur)):
        for i, f in enumerate(d['f_' + name]):
            assert np.array_equal(fn(rgb, float(f)), d[f'{name}_{i}']), (name, f)
    assert np.array_equal(D.to_grayscale3(rgb), d['gray'])
    assert D.erase_value([-0.3, 1.7, -2.2, 0.5, 3.9]).tolist() == [180, 177, 207, 127, 226]      # torch: tensor.mul(255).byte()


def test_randaug_ops_pinned_to_pillow(golden):
    """The five histogram / filter ops of the no-box colour branch (autoaug_fast.py:219-224, 244-250, 371-372, 407) against Pillow's own
    outputs for every level the policy draws (tests/golden/randaug_pil.npz): bit for bit."""
    from oracle import datapath_oracle as D
    d = golden('randaug_pil.npz')
    for k in range(int(d['n_img'])):
        a = d[f'img{k}']
        assert np.array_equal(D.autocontrast(a), d[f'autocontrast{k}']), k
        assert np.array_equal(D.equalize(a), d[f'equalize{k}']), k
        for level in range(1, 10):
            assert np.array_equal(D.solarize(a, D.randaug_level('Solarize', level)), d[f'solarize{k}_{level}']), (k, level)
            assert np.array_equal(D.posterize(a, D.randaug_level('Posterize', level)), d[f'posterize{k}_{level}']), (k, level)
            assert np.array_equal(D.adjust_sharpness(a, D.randaug_level('Sharpness', level)), d[f'sharpness{k}_{level}']), (k, level)
    assert [D.randaug_level('Posterize', v) for v in range(1, 10)] == [4, 4, 3, 3, 2, 2, 2, 1, 1]
    assert [D.randaug_level('Solarize', v) for v in (1, 5, 9)] == [231, 128, 26]


def test_colour_branch_renders_all_nine_ops():
    """RandomAugmentBBox_Fast on an image without boxes (semi_aug.py:464-477, 494-497): every one of RANDOM_COLOR_POLICY_OPS becomes a
    pass (Identity: none); nothing is skipped."""
    from dsl_amd import _lib as L
    from dsl_amd.datapath import RandomAugmentBBox_Fast
    t = RandomAugmentBBox_Fast(aug_type='affine')
    t.rng = np.random.RandomState(5)
    seen = {}
    for _ in range(400):
        r = t(dict(img_shape=(40, 50, 3), gt_bboxes=np.zeros((0, 4), np.float32), gt_labels=np.zeros((0,), np.int64)))
        for p_ in r['_aug']:
            seen.setdefault(p_['op'], set()).add(p_['kind'])
    assert t.skipped_ops == 0
    assert seen == dict(AutoContrast={L.AUG_AUTOCONTRAST}, Equalize={L.AUG_EQUALIZE}, Solarize={L.AUG_SOLARIZE}, Color={L.AUG_SATURATION},
                        Contrast={L.AUG_CONTRAST}, Brightness={L.AUG_BRIGHTNESS}, Sharpness={L.AUG_SHARPNESS}, Posterize={L.AUG_POSTERIZE})
