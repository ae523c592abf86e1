"""CPU oracle of the GPU data path - TEST INFRASTRUCTURE: only tests/ may import it.

Restates, in numpy, the per-sample stages of the reference's training pipeline (configs/fcos_semi/RLA_*.py:68-82) and the
loader's merge/pad:
  rescale_size / imrescale      mmcv.image.geometric (mmcv-full 1.3.10, un-vendored; call site transforms.py:218-247)
  resize_bilinear_u8            cv2.resize(INTER_LINEAR) on uint8: OpenCV's fixed-point path (resize.cpp: coefficient tables
                                scaled by 2^11, HResizeLinear in int32, VResizeLinear (((b0*(S0>>4))>>16) + ((b1*(S1>>4))>>16) + 2) >> 2)
                                - PARITY UNPINNED: cv2 is not installed here, the reference's tests hold no vectors for it
  resize_bboxes                 transforms.py:249-258
  patch_shuffle                 transforms.py:2143-2248 - PINNED: tests/golden/patch_shuffle.json holds the reference class's own
                                outputs (tests/golden/make_golden.py ps)
  flip_horizontal / bbox_flip   transforms.py:397-429, mmcv.imflip
  imnormalize                   mmcv.imnormalize_: BGR->RGB, subtract(mean), multiply(1/std) in fp32 (transforms.py:652-690)
  pad / merge_pad               mmcv.impad_to_multiple (transforms.py:581-650); MultiDataLoader._merge_data2one_batch
                                (datasets/builder.py:236-267)
"""
import numpy as np


def rescale_size(old_wh, scale):
    w, h = old_wh
    if isinstance(scale, (float, int)):
        sf = float(scale)
    else:
        max_long, max_short = max(scale), min(scale)
        sf = min(max_long / max(h, w), max_short / min(h, w))
    return int(w * float(sf) + 0.5), int(h * float(sf) + 0.5)


def _axis_tables(dsize, ssize):
    scale = 1.0 / (float(dsize) / float(ssize))
    d = np.arange(dsize, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    lo = s < 0
    f[lo], s[lo] = 0.0, 0
    hi = s >= ssize - 1
    f[hi], s[hi] = 0.0, ssize - 1
    a0 = np.rint((np.float32(1.0) - f) * np.float32(2048.0)).astype(np.int64)
    a1 = np.rint(f * np.float32(2048.0)).astype(np.int64)
    a0[hi], a1[hi] = 2048, 0
    s1 = np.minimum(s + 1, ssize - 1)
    return s, s1, a0, a1


def resize_bilinear_u8(img, new_wh):
    """img uint8 [H, W, C] -> uint8 [new_h, new_w, C]."""
    new_w, new_h = new_wh
    h, w = img.shape[:2]
    if (new_h, new_w) == (h, w):
        return img.copy()
    sx0, sx1, ax0, ax1 = _axis_tables(new_w, w)
    sy0, sy1, by0, by1 = _axis_tables(new_h, h)
    src = img.astype(np.int64)
    hz = src[:, sx0] * ax0[None, :, None] + src[:, sx1] * ax1[None, :, None]          # [H, new_w, C] int
    r0, r1 = hz[sy0], hz[sy1]
    v = (((by0[:, None, None] * (r0 >> 4)) >> 16) + ((by1[:, None, None] * (r1 >> 4)) >> 16) + 2) >> 2
    return np.clip(v, 0, 255).astype(np.uint8)


def resize_bboxes(bboxes, scale_factor, img_shape, clip=True):
    b = bboxes * scale_factor
    if clip:
        b[:, 0::2] = np.clip(b[:, 0::2], 0, img_shape[1])
        b[:, 1::2] = np.clip(b[:, 1::2], 0, img_shape[0])
    return b


def patch_shuffle_crop(h, w, place, mode):
    """Split position of PatchShuffle (transforms.py:2185-2199); None = the image is returned unchanged."""
    if mode == 'flip':
        crop_h, crop_w = h, min(int(round(w * place)), w)
        if crop_w == w or crop_w == 0:
            return None
    else:
        crop_h, crop_w = min(int(round(h * place)), h), w
        if crop_h == h or crop_h == 0:
            return None
    return crop_h, crop_w


def patch_shuffle_image(img, place, mode):
    h, w = img.shape[:2]
    c = patch_shuffle_crop(h, w, place, mode)
    if c is None:
        return img
    crop_h, crop_w = c
    if mode == 'flip':
        return np.concatenate([img[:, crop_w:], img[:, :crop_w]], 1)
    return np.concatenate([img[crop_h:], img[:crop_h]], 0)


def patch_shuffle_boxes(bboxes, labels, h, w, place, mode):
    """transforms.py:2201-2246: boxes on one side of the split move with their part, boxes across it are cut in two."""
    c = patch_shuffle_crop(h, w, place, mode)
    if c is None or len(bboxes) == 0:
        return bboxes, labels
    crop_h, crop_w = c
    out, lab = [], []
    for i in range(bboxes.shape[0]):
        x1, y1, x2, y2 = bboxes[i]
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
            if labels is not None:
                lab.append(labels[i])
        elif mode == 'flip':
            out += [[x1 + w - crop_w, y1, w - 1, y2], [0, y1, x2 - crop_w, y2]]
            if labels is not None:
                lab += [labels[i], labels[i]]
        else:
            out += [[x1, y1 + h - crop_h, x2, h - 1], [x1, 0, x2, y2 - crop_h]]
            if labels is not None:
                lab += [labels[i], labels[i]]
    return np.array(out).astype(np.float32), (None if labels is None else np.array(lab).astype(np.int64))


def bbox_flip_horizontal(bboxes, w):
    f = bboxes.copy()
    f[..., 0::4] = w - bboxes[..., 2::4]
    f[..., 2::4] = w - bboxes[..., 0::4]
    return f


def imnormalize(img_u8, mean, std, to_rgb=True):
    mean32, std32 = np.asarray(mean, np.float32), np.asarray(std, np.float32)
    inv = (1.0 / std32.astype(np.float64)).astype(np.float32)
    x = img_u8.astype(np.float32)
    if to_rgb:
        x = x[..., ::-1]
    return (x - mean32) * inv


def pad_to_multiple(shape_hw, divisor):
    return int(np.ceil(shape_hw[0] / divisor)) * divisor, int(np.ceil(shape_hw[1] / divisor)) * divisor


def prepare_batch(samples, mean, std, to_rgb=True, size_divisor=32):
    """samples: list of dict(img uint8 HWC, scale (w, h) tuple or None, ps=(place, mode) or None, flip bool).
    Returns [N, 3, Hc, Wc] fp32 and the per-sample (new_h, new_w)."""
    outs, shapes = [], []
    for s in samples:
        img = s['img']
        h, w = img.shape[:2]
        if s.get('scale') is not None:
            img = resize_bilinear_u8(img, rescale_size((w, h), s['scale']))
        if s.get('ps') is not None:
            img = patch_shuffle_image(img, *s['ps'])
        if s.get('flip'):
            img = img[:, ::-1]
        shapes.append(img.shape[:2])
        outs.append(imnormalize(img, mean, std, to_rgb).transpose(2, 0, 1))
    padded = [pad_to_multiple(sh, size_divisor) for sh in shapes]
    hc, wc = max(p[0] for p in padded), max(p[1] for p in padded)
    batch = np.zeros((len(samples), 3, hc, wc), np.float32)
    for i, o in enumerate(outs):
        batch[i, :, :o.shape[1], :o.shape[2]] = o
    return batch, shapes
