"""Row f3 on the GPU: dsl_image_prep (resize -> PatchShuffle -> flip -> normalise -> pad / merge-pad, one launch per batch)
against the CPU restatement, bit for bit; the batch it builds feeds the detector's train_step."""
import random

import numpy as np
import pytest
import torch

from oracle import datapath_oracle as DO

pytestmark = pytest.mark.gpu
NORM = dict(mean=[123.675, 116.28, 103.53], std=[58.395, 57.12, 57.375], to_rgb=True)
PIPE = [dict(type='LoadImageFromFile'), dict(type='LoadAnnotations', with_bbox=True),
        dict(type='Resize', img_scale=[(1333, 640), (1333, 800)], multiscale_mode='value', keep_ratio=True),
        dict(type='PatchShuffle', ratio=0.5, ranges=[0.0, 1.0], mode=['flip', 'flop']),
        dict(type='RandomFlip', flip_ratio=0.5), dict(type='Normalize', **NORM), dict(type='Pad', size_divisor=32),
        dict(type='DefaultFormatBundle'), dict(type='Collect', keys=['img', 'gt_bboxes', 'gt_labels', 'gt_bboxes_ignore'])]


def _samples(rng, sizes):
    out = []
    for i, (h, w) in enumerate(sizes):
        n = int(rng.randint(1, 6))
        x1, y1 = rng.uniform(0, w - 20, n), rng.uniform(0, h - 20, n)
        b = np.stack([x1, y1, x1 + rng.uniform(4, w / 2, n), y1 + rng.uniform(4, h / 2, n)], 1).astype(np.float32)
        out.append(dict(img=rng.randint(0, 256, (h, w, 3)).astype(np.uint8), gt_bboxes=b, gt_labels=rng.randint(0, 80, n),
                        gt_bboxes_ignore=b[:1] + 1, filename=f'im{i}.jpg'))
    return out


@pytest.mark.parametrize('seed', [0, 1, 2, 3])
def test_image_prep_matches_oracle_bit_for_bit(seed):
    from dsl_amd.datapath import GpuBatchPipeline
    rng = np.random.RandomState(seed)
    sizes = [(480, 640), (375, 500), (640, 427)] if seed % 2 == 0 else [(97, 131), (64, 64), (200, 50), (33, 77)]
    samples = _samples(rng, sizes)
    np.random.seed(100 + seed)
    random.seed(200 + seed)
    batch = GpuBatchPipeline(PIPE)(samples)
    torch.cuda.synchronize()
    metas = batch['img_metas']
    spec = []
    for s, m in zip(samples, metas):
        ps = (m['PS_place'], m['PS_mode']) if m['PS'] else None
        spec.append(dict(img=s['img'], scale=(1333, 640) if m['scale_idx'] == 0 else (1333, 800), ps=ps, flip=m['flip']))
    want, shapes = DO.prepare_batch(spec, NORM['mean'], NORM['std'], True, 32)
    got = batch['img'].cpu().numpy()
    assert got.shape == want.shape
    for i, (h, w) in enumerate(shapes):
        assert metas[i]['img_shape'][:2] == (h, w)
    assert np.array_equal(got, want), float(np.abs(got - want).max())
    # boxes: the same stages on the oracle side
    for s, m, gb, gl in zip(samples, metas, batch['gt_bboxes'], batch['gt_labels']):
        h0, w0 = s['img'].shape[:2]
        b = DO.resize_bboxes(s['gt_bboxes'].copy(), m['scale_factor'], m['img_shape'][:2])
        lab = s['gt_labels'].astype(np.int64)
        if m['PS']:
            b, lab = DO.patch_shuffle_boxes(b, lab, m['img_shape'][0], m['img_shape'][1], m['PS_place'], m['PS_mode'])
        if m['flip']:
            b = DO.bbox_flip_horizontal(b, m['img_shape'][1])
        assert np.array_equal(gb.numpy(), np.asarray(b, np.float32).reshape(-1, 4)) and np.array_equal(gl.numpy(), lab)


def test_every_stage_combination_small():
    """All (flip, PatchShuffle mode, resampling or not) combinations on odd sizes, forced parameters."""
    from dsl_amd import _lib as L
    rng = np.random.RandomState(5)
    for (h, w), (nh, nw) in (((19, 27), (19, 27)), ((19, 27), (31, 44)), ((40, 23), (17, 10)), ((8, 8), (64, 64))):
        img = rng.randint(0, 256, (h, w, 3)).astype(np.uint8)
        src = torch.from_numpy(img).cuda()
        for flip in (0, 1):
            for mode, crop in ((0, 0), (1, max(1, nw // 3)), (2, max(1, nh // 2))):
                for to_rgb in (0, 1):
                    it = (L.ImagePrepItem * 1)()
                    it[0].src, it[0].src_h, it[0].src_w, it[0].new_h, it[0].new_w = src.data_ptr(), h, w, nh, nw
                    it[0].flip, it[0].ps_mode, it[0].ps_crop, it[0].to_rgb = flip, mode, crop, to_rgb
                    inv = (1.0 / np.asarray(NORM['std'], np.float32).astype(np.float64)).astype(np.float32)
                    for c in range(3):
                        it[0].mean[c], it[0].inv_std[c] = NORM['mean'][c], float(inv[c])
                    tab = torch.frombuffer(bytearray(bytes(it)), dtype=torch.uint8).cuda()
                    hc, wc = nh + 5, nw + 9
                    out = torch.full((1, 3, hc, wc), 7.0, device='cuda')
                    L.check(L.lib.dsl_image_prep(L.ptr(tab), 1, L.ptr(out), hc, wc, L.stream_ptr()))
                    torch.cuda.synchronize()
                    ref = DO.resize_bilinear_u8(img, (nw, nh))
                    if mode == 1:
                        ref = np.concatenate([ref[:, crop:], ref[:, :crop]], 1)
                    if mode == 2:
                        ref = np.concatenate([ref[crop:], ref[:crop]], 0)
                    if flip:
                        ref = ref[:, ::-1]
                    want = np.zeros((3, hc, wc), np.float32)
                    want[:, :nh, :nw] = DO.imnormalize(ref, NORM['mean'], NORM['std'], bool(to_rgb)).transpose(2, 0, 1)
                    assert np.array_equal(out[0].cpu().numpy(), want), ((h, w), (nh, nw), flip, mode, to_rgb)


def test_batch_feeds_the_training_step():
    from dsl_amd import detectors  # noqa: F401
    from dsl_amd.datapath import GpuBatchPipeline
    from dsl_amd.registry import build_detector
    from util import fcos_model_cfg
    from oracle import fcos_oracle as O
    rng = np.random.RandomState(9)
    pipe = [dict(type='Resize', img_scale=[(200, 120), (200, 150)], multiscale_mode='value', keep_ratio=True),
            dict(type='PatchShuffle', ratio=0.5, ranges=[0.0, 1.0], mode=['flip', 'flop']), dict(type='RandomFlip', flip_ratio=0.5),
            dict(type='Normalize', **NORM), dict(type='Pad', size_divisor=32)]
    np.random.seed(3)
    random.seed(4)
    batch = GpuBatchPipeline(pipe)(_samples(rng, [(96, 128), (120, 90)]))
    model = build_detector(fcos_model_cfg())
    model.load_state_dict(O.synth_state_dict(0))
    model = model.cuda()
    losses = model.forward_train(batch['img'], batch['img_metas'], batch['gt_bboxes'], batch['gt_labels'], batch['gt_bboxes_ignore'])
    sum(losses.values()).backward()
    torch.cuda.synchronize()
    assert all(np.isfinite(float(v)) for v in losses.values()) and torch.isfinite(model.store.grad).all()


def test_unbuilt_augmentation_modes_are_refused():
    """RandomAugmentBBox_Fast is built for the aug_types the DSL configs use; the others fail at build time, not silently."""
    from dsl_amd.datapath import GpuBatchPipeline
    with pytest.raises(NotImplementedError, match='aug_type'):
        GpuBatchPipeline([dict(type='RandomAugmentBBox_Fast', aug_type='strong'), dict(type='Normalize', **NORM)])
