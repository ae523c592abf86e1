"""GPU data path: the per-sample stages of the reference's training pipelines and the loader's
merge/pad, with the image work as ONE HIP launch per batch (csrc/datapath.hip, dsl_image_prep) and the box arithmetic - a few
dozen numbers per image - on the host, exactly as the reference computes it.

Mirrors, by name and argument meaning (configs/fcos_semi/RLA_*.py:68-82; mmdet/datasets/pipelines/transforms.py):
  Resize(img_scale, multiscale_mode='value'|'range', keep_ratio=True)   :41-332
  PatchShuffle(ratio, ranges, mode)                                     :2143-2248
  RandomFlip(flip_ratio)  (horizontal)                                  :334-470
  Normalize(mean, std, to_rgb)                                          :652-690
  Pad(size_divisor)                                                     :581-650
  MultiDataLoader._merge_data2one_batch (pad to the batch's largest)    mmdet/datasets/builder.py:236-267
A transform's __call__(results) only DRAWS its random parameters and updates boxes / meta; `GpuBatchPipeline` then renders all
images of the batch.  Not here (they resample through PIL / imgaug, neither is in this image, and nothing pins their
output): RandomAugmentBBox_Fast (semi_aug.py:344-531) and UBAug (transforms.py:2098-2140) - the unlabeled stream's photometric
/ affine augmentation stays on the CPU side of the boundary.
Fails loudly without the HIP library: there is no CPU rendering path in the product (the CPU restatement is oracle/datapath_oracle.py,
test infrastructure).
"""
import ctypes as C
import random

import numpy as np
import torch

from . import _lib as L
from .registry import Registry

PIPELINES = Registry('pipeline')


def rescale_size(old_wh, scale):
    """mmcv.rescale_size (call site transforms.py:221-226)."""
    w, h = old_wh
    if isinstance(scale, (float, int)):
        sf = float(scale)
    else:
        sf = min(max(scale) / max(h, w), min(scale) / min(h, w))
    return int(w * float(sf) + 0.5), int(h * float(sf) + 0.5)


@PIPELINES.register_module()
class Resize:
    def __init__(self, img_scale=None, multiscale_mode='range', ratio_range=None, keep_ratio=True, bbox_clip_border=True, **kw):
        self.img_scale = [img_scale] if isinstance(img_scale, tuple) else list(img_scale or [])
        assert multiscale_mode in ('value', 'range') and ratio_range is None and keep_ratio, 'configs use keep_ratio Resize'
        self.multiscale_mode, self.bbox_clip_border = multiscale_mode, bbox_clip_border

    def _random_scale(self):
        """:185-216."""
        if len(self.img_scale) == 1:
            return self.img_scale[0], 0
        if self.multiscale_mode == 'value':                    # random_select (:107-124)
            idx = np.random.randint(len(self.img_scale))
            return self.img_scale[idx], idx
        longs, shorts = [max(s) for s in self.img_scale], [min(s) for s in self.img_scale]      # random_sample (:126-150)
        return (np.random.randint(min(longs), max(longs) + 1), np.random.randint(min(shorts), max(shorts) + 1)), None

    def __call__(self, r):
        if 'scale' not in r:
            r['scale'], r['scale_idx'] = self._random_scale()
        h, w = r['img_shape'][:2]
        new_w, new_h = rescale_size((w, h), r['scale'])
        sf = np.array([new_w / w, new_h / h, new_w / w, new_h / h], dtype=np.float32)
        r['img_shape'] = r['pad_shape'] = (new_h, new_w, 3)
        r['scale_factor'], r['keep_ratio'] = sf, True
        for key in r.get('bbox_fields', []):
            b = r[key] * sf
            if self.bbox_clip_border:
                b[:, 0::2] = np.clip(b[:, 0::2], 0, new_w)
                b[:, 1::2] = np.clip(b[:, 1::2], 0, new_h)
            r[key] = b
        return r


@PIPELINES.register_module()
class PatchShuffle:
    def __init__(self, ratio=0.5, ranges=(0.2, 0.8), mode=('flip', 'flop')):
        assert isinstance(ratio, float)
        self.ratio, self.ranges, self.mode = ratio, list(ranges), list(mode)

    def __call__(self, r):
        if np.random.rand(1) > self.ratio:
            r['PS'], r['PS_place'], r['PS_mode'] = False, None, None
            return r
        r['PS'] = True
        h, w = r['img_shape'][:2]
        place = np.random.rand(1)[0] * abs(self.ranges[1] - self.ranges[0]) + self.ranges[0]
        mode = random.choice(self.mode)
        r['PS_place'], r['PS_mode'] = place, mode
        if mode == 'flip':
            crop_h, crop_w = h, min(int(round(w * place)), w)
            if crop_w == w or crop_w == 0:
                return r
        elif mode == 'flop':
            crop_h, crop_w = min(int(round(h * place)), h), w
            if crop_h == h or crop_h == 0:
                return r
        else:
            raise NotImplementedError
        r['_ps'] = (1, crop_w) if mode == 'flip' else (2, crop_h)       # what the launch needs: mode, split
        for key in r.get('bbox_fields', []):
            if len(r[key]) == 0:
                continue
            bb, out, lab = r[key], [], []
            for i in range(bb.shape[0]):
                x1, y1, x2, y2 = bb[i]
                if (x1 - crop_w + 1) * (x2 - crop_w + 1) >= 0 and (y1 - crop_h + 1) * (y2 - crop_h + 1) >= 0:
                    if mode == 'flip':
                        if x1 - crop_w + 1 < 0:
                            x1, x2 = x1 + w - crop_w, x2 + w - crop_w
                        if x2 - crop_w + 1 > 0:
                            x1, x2 = x1 - crop_w, x2 - crop_w
                    else:
                        if y1 - crop_h + 1 < 0:
                            y1, y2 = y1 + h - crop_h, y2 + h - crop_h
                        if y2 - crop_h + 1 > 0:
                            y1, y2 = y1 - crop_h, y2 - crop_h
                    out.append([x1, y1, x2, y2])
                    if key == 'gt_bboxes':
                        lab.append(r['gt_labels'][i])
                elif mode == 'flip':
                    out += [[x1 + w - crop_w, y1, w - 1, y2], [0, y1, x2 - crop_w, y2]]
                    if key == 'gt_bboxes':
                        lab += [r['gt_labels'][i]] * 2
                else:
                    out += [[x1, y1 + h - crop_h, x2, h - 1], [x1, 0, x2, y2 - crop_h]]
                    if key == 'gt_bboxes':
                        lab += [r['gt_labels'][i]] * 2
            r[key] = np.array(out).astype(np.float32)
            if key == 'gt_bboxes':
                r['gt_labels'] = np.array(lab).astype(np.int64)
        return r


@PIPELINES.register_module()
class RandomFlip:
    def __init__(self, flip_ratio=None, direction='horizontal'):
        assert direction == 'horizontal', 'the configs flip horizontally'
        self.flip_ratio = flip_ratio

    def __call__(self, r):
        if 'flip' not in r:
            r['flip'] = bool(self.flip_ratio is not None and np.random.rand() < self.flip_ratio)
            r['flip_direction'] = 'horizontal' if r['flip'] else None
        if r['flip']:
            w = r['img_shape'][1]
            for key in r.get('bbox_fields', []):
                b = r[key]
                f = b.copy()
                f[..., 0::4] = w - b[..., 2::4]
                f[..., 2::4] = w - b[..., 0::4]
                r[key] = f
        return r


@PIPELINES.register_module()
class Normalize:
    def __init__(self, mean, std, to_rgb=True):
        self.mean, self.std, self.to_rgb = np.array(mean, dtype=np.float32), np.array(std, dtype=np.float32), to_rgb

    def __call__(self, r):
        r['img_norm_cfg'] = dict(mean=self.mean, std=self.std, to_rgb=self.to_rgb)
        return r


@PIPELINES.register_module()
class Pad:
    def __init__(self, size=None, size_divisor=None, pad_val=0):
        assert size is None and size_divisor is not None and pad_val == 0, 'the configs pad to a multiple with zeros'
        self.size_divisor = size_divisor

    def __call__(self, r):
        h, w = r['img_shape'][:2]
        d = self.size_divisor
        r['pad_shape'] = (int(np.ceil(h / d)) * d, int(np.ceil(w / d)) * d, 3)
        r['pad_size_divisor'] = d
        return r


_SKIP = ('LoadImageFromFile', 'LoadAnnotations', 'DefaultFormatBundle', 'Collect', 'ImageToTensor')
_CPU_SIDE = ('RandomAugmentBBox_Fast', 'UBAug')


class GpuBatchPipeline:
    """pipeline: the config's list of transform dicts (train_pipeline / unlabel_train_pipeline).  __call__(samples) with samples =
    list of dict(img = uint8 [H, W, 3] BGR tensor / array, gt_bboxes [G, 4] fp32, gt_labels [G] int64, gt_bboxes_ignore [K, 4],
    filename) returns the batch dict the detector's train_step takes: img [N, 3, Hc, Wc] fp32 on the device, img_metas and the
    per-image box lists."""

    def __init__(self, pipeline, device='cuda'):
        self.device = device
        self.transforms = []
        for cfg in pipeline:
            t = cfg['type']
            if t in _SKIP:
                continue
            if t in _CPU_SIDE:
                raise NotImplementedError(f'{t} resamples through PIL / imgaug: it runs on the CPU side, before the image is handed over')
            self.transforms.append(PIPELINES.build(cfg))
        norm = [t for t in self.transforms if isinstance(t, Normalize)]
        assert len(norm) == 1, 'the pipeline needs exactly one Normalize'
        self.norm = norm[0]

    def __call__(self, samples):
        n = len(samples)
        results, srcs = [], []
        for s in samples:
            img = s['img']
            img = torch.from_numpy(np.ascontiguousarray(img)) if isinstance(img, np.ndarray) else img
            assert img.dtype == torch.uint8 and img.dim() == 3 and img.shape[2] == 3
            srcs.append(img.to(self.device, non_blocking=True).contiguous())
            h, w = img.shape[:2]
            r = dict(filename=s.get('filename'), ori_filename=s.get('filename'), ori_shape=(h, w, 3), img_shape=(h, w, 3),
                     gt_bboxes=np.asarray(s.get('gt_bboxes', np.zeros((0, 4))), np.float32).reshape(-1, 4),
                     gt_labels=np.asarray(s.get('gt_labels', np.zeros((0,))), np.int64).reshape(-1),
                     gt_bboxes_ignore=np.asarray(s.get('gt_bboxes_ignore', np.zeros((0, 4))), np.float32).reshape(-1, 4),
                     bbox_fields=['gt_bboxes_ignore', 'gt_bboxes'], scale_factor=np.ones(4, np.float32))
            for k in ('scale', 'flip'):          # forced parameters (tests; the test pipeline's MultiScaleFlipAug)
                if k in s:
                    r[k] = s[k]
            for t in self.transforms:
                r = t(r)
            r.setdefault('pad_shape', r['img_shape'])
            r.setdefault('flip', False)
            r.setdefault('flip_direction', None)
            results.append(r)
        hc, wc = max(r['pad_shape'][0] for r in results), max(r['pad_shape'][1] for r in results)       # merge/pad
        items = (L.ImagePrepItem * n)()
        inv = (1.0 / self.norm.std.astype(np.float64)).astype(np.float32)
        for i, (r, src) in enumerate(zip(results, srcs)):
            it = items[i]
            it.src, it.src_h, it.src_w = src.data_ptr(), src.shape[0], src.shape[1]
            it.new_h, it.new_w = r['img_shape'][0], r['img_shape'][1]
            it.flip, it.to_rgb = int(bool(r['flip'])), int(bool(self.norm.to_rgb))
            it.ps_mode, it.ps_crop = r.get('_ps', (0, 0))
            for c in range(3):
                it.mean[c], it.inv_std[c] = float(self.norm.mean[c]), float(inv[c])
        tab = torch.frombuffer(bytearray(bytes(items)), dtype=torch.uint8).to(self.device)
        out = torch.empty(n, 3, hc, wc, dtype=torch.float32, device=self.device)
        L.check(L.lib.dsl_image_prep(L.ptr(tab), n, L.ptr(out), hc, wc, L.stream_ptr()), 'dsl_image_prep')
        self._keep = (srcs, tab)            # alive until the launch has run (stream order)
        meta_keys = ('filename', 'ori_filename', 'ori_shape', 'img_shape', 'pad_shape', 'scale_factor', 'scale_idx', 'flip',
                     'flip_direction', 'img_norm_cfg', 'PS', 'PS_place', 'PS_mode')
        metas = []
        for r in results:
            m = {k: r.get(k) for k in meta_keys}
            m['batch_input_shape'] = (hc, wc)
            metas.append(m)
        return dict(img=out, img_metas=metas, gt_bboxes=[torch.from_numpy(r['gt_bboxes']) for r in results],
                    gt_labels=[torch.from_numpy(r['gt_labels']) for r in results],
                    gt_bboxes_ignore=[torch.from_numpy(r['gt_bboxes_ignore']) for r in results])
