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
  RandomAugmentBBox_Fast(aug_type='affine'|'default')                  mmdet/datasets/pipelines/semi_aug.py:344-531
  UBAug()                                                               :2098-2140
A transform's __call__(results) only DRAWS its random parameters and updates boxes / meta; `GpuBatchPipeline` then renders all
images of the batch: one launch for the labeled stream, and for the unlabeled stream (the last two transforms) one launch per
augmentation pass over uint8 canvases (dsl_image_prep_u8 -> dsl_image_aug ... -> dsl_image_normalize).  UBAug's colour,
grayscale and blur passes are Pillow's 8-bit arithmetic, pinned to Pillow's outputs; imgaug's Affine follows its documentation
(unpinned: imgaug is not in this image); RandomErasing's noise is this library's own stream.
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


def _affine_forward(kind, value, w, h):
    """Forward matrix (input pixel -> output pixel) of one child of the reference's imgaug AFFINE_TRANSFORM[_WEAK]
    (semi_aug.py:36-88) about the image centre (w / 2 - 0.5, h / 2 - 0.5); rotation / shear in degrees, positive rotation
    clockwise in image coordinates.  imgaug is not available to this build: the convention follows its documentation, UNPINNED."""
    cx, cy = w / 2.0 - 0.5, h / 2.0 - 0.5
    T = lambda tx, ty: np.array([[1, 0, tx], [0, 1, ty], [0, 0, 1]], np.float64)
    M = np.eye(3)
    if kind == 'translate_x':
        M = T(value * w, 0)
    elif kind == 'translate_y':
        M = T(0, value * h)
    elif kind == 'rotate':
        a = np.deg2rad(value)
        M = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]])
    elif kind == 'shear':
        M = np.array([[1, -np.tan(np.deg2rad(value)), 0], [0, 1, 0], [0, 0, 1]])
    return T(cx, cy) @ M @ T(-cx, -cy)


def _draw_affine(rng, weak):
    """iaa.OneOf of the four Affine children with their parameter ranges (semi_aug.py:36-88), interpolation order from [0, 1]."""
    kind = ('translate_x', 'translate_y', 'rotate', 'shear')[rng.randint(4)]
    if kind.startswith('translate'):
        lim = 0.05 if weak else 0.1
    else:
        lim = 10.0 if weak else 30.0           # DEGREE = 30 (:35); the in-box transform uses 10
    return kind, float(rng.uniform(-lim, lim)), int(rng.randint(2))


@PIPELINES.register_module()
class RandomAugmentBBox_Fast:
    """mmdet/datasets/pipelines/semi_aug.py:344-531.  Built: aug_type 'affine' (the DSL config's, RLA_*.py:93) and 'default'.
    __call__ draws the parameters and moves the boxes on the host; the image work is recorded as passes (`r['_aug']`) that
    GpuBatchPipeline renders with dsl_image_aug.  An image WITHOUT boxes takes the reference's colour branch (:494-497,
    RandAug policy of one op at probability 1, autoaug_fast.py): Color / Contrast / Brightness / Sharpness with the factor
    0.18 level + 0.1 (`_enhancer_impl`, :392-399), Solarize with threshold 256 - int(25.6 level) (:371-372), Posterize keeping
    4 - int(0.4 level) bits (:244-247), AutoContrast and Equalize (:219-224) - all nine ops are rendered, Pillow's arithmetic bit for
    bit (tests/golden/ubaug_pil.npz, randaug_pil.npz)."""
    COLOR_OPS = ('Identity', 'AutoContrast', 'Equalize', 'Solarize', 'Color', 'Contrast', 'Brightness', 'Sharpness', 'Posterize')

    def __init__(self, aug_type='strong', magnitude=10, weighted_inbox_selection=False):
        if aug_type not in ('affine', 'default'):
            raise NotImplementedError(f"RandomAugmentBBox_Fast(aug_type={aug_type!r}): 'affine' (configs/fcos_semi) and 'default' are built")
        self.aug_type, self.magnitude, self.weighted = aug_type, magnitude, weighted_inbox_selection
        self.skipped_ops = 0            # ops drawn but not rendered: none since round 6 (kept for the replay test's assertion)
        self.rng = np.random

    def __call__(self, r):
        h, w = r['img_shape'][:2]
        passes = r.setdefault('_aug', [])
        boxes = r['gt_bboxes']
        if boxes.shape[0] == 0:
            op = self.COLOR_OPS[self.rng.randint(len(self.COLOR_OPS))]
            level = self.rng.randint(1, self.magnitude)
            f = float(level) * 1.8 / 10 + 0.1
            if op in ('Color', 'Contrast', 'Brightness', 'Sharpness'):
                kind = dict(Color=L.AUG_SATURATION, Contrast=L.AUG_CONTRAST, Brightness=L.AUG_BRIGHTNESS, Sharpness=L.AUG_SHARPNESS)[op]
                passes.append(dict(kind=kind, f=f, op=op))
            elif op == 'Solarize':
                passes.append(dict(kind=L.AUG_SOLARIZE, f=float(256 - int(level * 256 / 10)), op=op))
            elif op == 'Posterize':
                passes.append(dict(kind=L.AUG_POSTERIZE, f=float(4 - int(level * 4 / 10)), op=op))
            elif op in ('AutoContrast', 'Equalize'):
                passes.append(dict(kind=L.AUG_AUTOCONTRAST if op == 'AutoContrast' else L.AUG_EQUALIZE, op=op))
            return r
        if self.aug_type == 'default':
            return r
        # array_to_bb (:150-155): imgaug's BoundingBox orders its corners (x1 <= x2, y1 <= y2) on construction
        boxes = np.stack([np.minimum(boxes[:, 0], boxes[:, 2]), np.minimum(boxes[:, 1], boxes[:, 3]),
                          np.maximum(boxes[:, 0], boxes[:, 2]), np.maximum(boxes[:, 1], boxes[:, 3])], 1)
        if self.rng.randint(2) == 0:           # bbox_affine_transform (:431-452): one box, the weak transform inside it
            n = boxes.shape[0]
            if self.weighted:
                area = (boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])
                k = int(self.rng.choice(n, p=area / area.sum()))
            else:
                k = int(self.rng.randint(n))
            x0, y0, x1, y1 = (int(v) for v in boxes[k])
            x0, y0, x1, y1 = max(x0, 0), max(y0, 0), min(x1, w), min(y1, h)     # (a box sticking out is clamped: the reference's
                                                                                # negative slice index would wrap around)
            if x1 > x0 and y1 > y0:
                kind, val, order = _draw_affine(self.rng, weak=True)
                passes.append(dict(kind=L.AUG_AFFINE, M=_affine_forward(kind, val, x1 - x0, y1 - y0), order=order, roi=(x0, y0, x1, y1)))
        else:                                   # affine_transform (:454-461): the whole image and its boxes
            kind, val, order = _draw_affine(self.rng, weak=False)
            M = _affine_forward(kind, val, w, h)
            passes.append(dict(kind=L.AUG_AFFINE, M=M, order=order, roi=(0, 0, w, h)))
            b = boxes.astype(np.float64)
            out = np.zeros_like(b)
            for i, (bx1, by1, bx2, by2) in enumerate(b):       # imgaug: the four corners move, the box is their hull
                c = np.array([[bx1, by1, 1], [bx2, by1, 1], [bx2, by2, 1], [bx1, by2, 1]], np.float64) @ M.T
                out[i] = [c[:, 0].min(), c[:, 1].min(), c[:, 0].max(), c[:, 1].max()]
            boxes = out
        boxes = boxes.astype(np.float64)
        boxes[:, 0::2] = np.clip(boxes[:, 0::2], 0, w)             # :519-527
        boxes[:, 1::2] = np.clip(boxes[:, 1::2], 0, h)
        keep = ((boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])) > 0
        r['gt_bboxes'] = boxes[keep].astype(np.float32)
        r['gt_labels'] = r['gt_labels'][keep]
        return r


@PIPELINES.register_module()
class UBAug:
    """mmdet/datasets/pipelines/transforms.py:2098-2140 (Unbiased Teacher's strong augmentation): RandomApply(ColorJitter(0.4, 0.4,
    0.4, 0.1), 0.8), RandomGrayscale(0.2), RandomApply(GaussianBlur(0.1 .. 2.0), 0.5), three RandomErasing.  The parameters are
    drawn here the way torchvision draws them (ColorJitter: a random order of the four adjustments, factors U(0.6, 1.4) and hue
    U(-0.1, 0.1); RandomErasing.get_params: ten attempts per rectangle); the image passes are Pillow's arithmetic, pinned by
    tests/golden/ubaug_pil.npz.  The reference hands mmcv's BGR array to ToPILImage as if it were RGB: the passes treat the
    stored channel order as RGB in the same way."""

    def __init__(self):
        self.rng = np.random

    @staticmethod
    def _erase_rect(rng, h, w, scale, ratio):
        area = h * w
        for _ in range(10):
            ea = area * rng.uniform(scale[0], scale[1])
            ar = np.exp(rng.uniform(np.log(ratio[0]), np.log(ratio[1])))
            eh, ew = int(round(np.sqrt(ea * ar))), int(round(np.sqrt(ea / ar)))
            if eh < h and ew < w:
                i, j = int(rng.randint(0, h - eh + 1)), int(rng.randint(0, w - ew + 1))
                return (j, i, j + ew, i + eh)
        return None

    def __call__(self, r):
        h, w = r['img_shape'][:2]
        passes, rng = r.setdefault('_aug', []), self.rng
        if rng.rand() < 0.8:
            fac = {L.AUG_BRIGHTNESS: rng.uniform(0.6, 1.4), L.AUG_CONTRAST: rng.uniform(0.6, 1.4), L.AUG_SATURATION: rng.uniform(0.6, 1.4),
                   L.AUG_HUE: rng.uniform(-0.1, 0.1)}
            kinds = [L.AUG_BRIGHTNESS, L.AUG_CONTRAST, L.AUG_SATURATION, L.AUG_HUE]
            for i in rng.permutation(4):
                f = float(fac[kinds[i]])
                passes.append(dict(kind=kinds[i], f=float(int(f * 255)) if kinds[i] == L.AUG_HUE else f))
        if rng.rand() < 0.2:
            passes.append(dict(kind=L.AUG_GRAY))
        if rng.rand() < 0.5:
            sigma = float(rng.uniform(0.1, 2.0))
            fr = blur_box_radius(sigma)
            if fr != 0:
                passes += [dict(kind=L.AUG_BLUR_H, f=fr)] * 3 + [dict(kind=L.AUG_BLUR_V, f=fr)] * 3
        rects = []
        for p_, scale, ratio in ((0.7, (0.05, 0.2), (0.3, 3.3)), (0.5, (0.02, 0.2), (0.1, 6)), (0.3, (0.02, 0.2), (0.05, 8))):
            if rng.rand() < p_:
                rc = self._erase_rect(rng, h, w, scale, ratio)
                if rc is not None:
                    rects.append(rc)
        if rects:
            passes.append(dict(kind=L.AUG_ERASE, rects=rects, seed=int(rng.randint(0, 2 ** 31 - 1))))
        return r


def blur_box_radius(radius, passes=3):
    """Pillow BoxBlur.c `_gaussian_blur_radius` (float arithmetic as in the C source): the fractional box radius of
    ImageFilter.GaussianBlur(radius)."""
    radius = np.float32(radius)
    sigma2 = np.float32(radius * radius / np.float32(passes))
    big_l = np.float32(np.sqrt(12.0 * np.float64(sigma2) + 1.0))
    l_ = np.float32(np.floor((np.float64(big_l) - 1.0) / 2.0))
    a = np.float32((2 * l_ + 1) * (l_ * (l_ + 1) - 3 * sigma2))
    a = np.float32(a / np.float32(6 * (sigma2 - (l_ + 1) * (l_ + 1))))
    return float(np.float32(l_ + a))


_SKIP = ('LoadImageFromFile', 'LoadAnnotations', 'DefaultFormatBundle', 'Collect', 'ImageToTensor')


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
        n_pass = max(len(r.get('_aug', ())) for r in results)
        if n_pass == 0:
            L.check(L.lib.dsl_image_prep(L.ptr(tab), n, L.ptr(out), hc, wc, L.stream_ptr()), 'dsl_image_prep')
        else:
            # the unlabeled stream (RandomAugmentBBox_Fast, UBAug): Resize / PatchShuffle / Flip into uint8 canvases, one launch per
            # augmentation pass over the batch (an image with fewer passes is copied), then Normalize / Pad
            a = torch.empty(n, hc, wc, 3, dtype=torch.uint8, device=self.device)
            b = torch.empty_like(a)
            sums = torch.empty(int(L.lib.dsl_image_aug_scratch_bytes(n)) // 8, dtype=torch.int64, device=self.device)
            L.check(L.lib.dsl_image_prep_u8(L.ptr(tab), n, L.ptr(a), hc, wc, L.stream_ptr()), 'dsl_image_prep_u8')
            # every pass's item table in ONE host -> device copy
            its = (L.AugItem * (n * n_pass))()
            need_mean = [0] * n_pass
            for k in range(n_pass):
                for i, r in enumerate(results):
                    it, ps = its[k * n + i], r.get('_aug', ())
                    it.h, it.w = r['img_shape'][0], r['img_shape'][1]
                    if k >= len(ps):
                        continue
                    ps_k = ps[k]
                    it.kind = ps_k['kind']
                    need_mean[k] |= int(it.kind == L.AUG_CONTRAST) | 2 * int(it.kind in (L.AUG_AUTOCONTRAST, L.AUG_EQUALIZE))
                    it.f[0] = float(ps_k.get('f', 0.0))
                    if it.kind == L.AUG_AFFINE:
                        x0, y0, x1, y1 = ps_k['roi']
                        Mi = np.linalg.inv(ps_k['M'])
                        for q in range(6):
                            it.m[q] = float(Mi[q // 3, q % 3])
                        it.roi[0], it.roi[1], it.roi[2], it.roi[3] = x0, y0, x1, y1
                        it.order, it.cval = int(ps_k['order']), 125
                    if it.kind == L.AUG_ERASE:
                        it.seed = ps_k['seed']
                        for q, rc in enumerate(ps_k['rects'][:3]):
                            for e in range(4):
                                it.rect[q][e] = int(rc[e])
            at = torch.frombuffer(bytearray(bytes(its)), dtype=torch.uint8).to(self.device)
            keep = [a, b, sums, at]
            isz = C.sizeof(L.AugItem)
            for k in range(n_pass):
                L.check(L.lib.dsl_image_aug(at.data_ptr() + k * n * isz, n, L.ptr(a), L.ptr(b), hc, wc, L.ptr(sums), need_mean[k],
                                            L.stream_ptr()), 'dsl_image_aug')
                a, b = b, a
            L.check(L.lib.dsl_image_normalize(L.ptr(a), L.ptr(tab), n, L.ptr(out), hc, wc, L.stream_ptr()), 'dsl_image_normalize')
            srcs = srcs + keep
        self._keep = (srcs, tab)            # alive until the launch has run (stream order)
        self.last_passes = [list(r.get('_aug', ())) for r in results]      # what was rendered (tests replay it on the oracle)
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
