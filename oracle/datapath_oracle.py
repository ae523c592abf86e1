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


# ----------------------------------------------------------------------------------------------------------------------
# The unlabeled stream's augmentations (configs/fcos_semi/RLA_*.py:93-94): UBAug (transforms.py:2098-2140) and
# RandomAugmentBBox_Fast(aug_type='affine') (semi_aug.py:344-531).
#   UBAug's image arithmetic is Pillow's (torchvision only draws the parameters: ColorJitter -> ImageEnhance / HSV shift,
#   RandomGrayscale -> convert('L'), GaussianBlur -> ImageFilter.GaussianBlur): restated below from Pillow's C sources
#   (Blend.c, Convert.c rgb2hsv / hsv2rgb, BoxBlur.c) and PINNED: tests/golden/ubaug_pil.npz holds Pillow's own outputs
#   (tests/golden/make_golden.py ubaug; Pillow is importable in the build container).
#   RandomErasing's noise is torch's RNG stream (unreproducible by construction): the semantics restated are the value mapping
#   mul(255).byte() (truncate toward zero, wrap mod 256).
#   imgaug's Affine is NOT available: the matrix convention below follows imgaug's documentation - parity UNPINNED.
def pil_blend(deg, img, f):
    """Image.blend(degenerate, image, f) of ImageEnhance (Blend.c): float32 deg + f * (img - deg), clipped, TRUNCATED."""
    f = np.float32(f)
    t = deg.astype(np.float32) + f * (img.astype(np.float32) - deg.astype(np.float32))
    return np.clip(t, 0, 255).astype(np.uint8)


def pil_luma(rgb):
    r, g, b = [rgb[..., i].astype(np.int64) for i in range(3)]
    return ((r * 19595 + g * 38470 + b * 7471 + 0x8000) >> 16).astype(np.uint8)


def adjust_brightness(rgb, f):
    return pil_blend(np.zeros_like(rgb), rgb, f)


def adjust_contrast(rgb, f):
    mean = int(pil_luma(rgb).astype(np.float64).mean() + 0.5)
    return pil_blend(np.full_like(rgb, mean), rgb, f)


def adjust_saturation(rgb, f):
    return pil_blend(np.repeat(pil_luma(rgb)[..., None], 3, -1), rgb, f)


def to_grayscale3(rgb):
    return np.repeat(pil_luma(rgb)[..., None], 3, -1)


def rgb_to_hsv_u8(rgb):
    r, g, b = [rgb[..., i].astype(np.float32) for i in range(3)]
    mx, mn = np.maximum(np.maximum(r, g), b), np.minimum(np.minimum(r, g), b)
    cr = (mx - mn).astype(np.float32)
    safe = np.where(cr > 0, cr, 1).astype(np.float32)
    rc, gc, bc = [((mx - c) / safe).astype(np.float32) for c in (r, g, b)]
    h = np.where(r == mx, (bc - gc).astype(np.float32),
                 np.where(g == mx, (2.0 + rc.astype(np.float64) - bc.astype(np.float64)).astype(np.float32),
                          (4.0 + gc.astype(np.float64) - rc.astype(np.float64)).astype(np.float32))).astype(np.float32)
    hd = np.fmod(h.astype(np.float64) / 6.0 + 1.0, 1.0).astype(np.float32)
    uh = np.where(cr > 0, np.clip((hd.astype(np.float64) * 255.0).astype(np.int64), 0, 255), 0)
    us = np.where(cr > 0, np.clip(((cr / np.where(mx > 0, mx, 1)).astype(np.float32).astype(np.float64) * 255.0).astype(np.int64), 0, 255), 0)
    return np.stack([uh, us, mx.astype(np.int64)], -1).astype(np.uint8)


def hsv_to_rgb_u8(hsv):
    H, S, V = [hsv[..., i].astype(np.float64) for i in range(3)]
    fh, fs = H * 6.0 / 255.0, S / 255.0
    i = np.floor(fh).astype(np.int64)
    f = fh - i
    rnd = lambda x: np.clip(np.floor(x + 0.5).astype(np.int64), 0, 255)
    p, q, t = rnd(V * (1 - fs)), rnd(V * (1 - fs * f)), rnd(V * (1 - fs * (1 - f)))
    v = V.astype(np.int64)
    sel = i % 6
    r = np.choose(sel, [v, q, p, p, t, v])
    g = np.choose(sel, [t, v, v, q, p, p])
    b = np.choose(sel, [p, p, t, v, v, q])
    gray = S == 0
    return np.stack([np.where(gray, v, c) for c in (r, g, b)], -1).astype(np.uint8)


def adjust_hue(rgb, f):
    """torchvision adjust_hue on a PIL image: the 8-bit H channel += uint8(f * 255) (wraps), back to RGB."""
    hsv = rgb_to_hsv_u8(rgb).astype(np.int64)
    hsv[..., 0] = (hsv[..., 0] + int(f * 255)) & 255          # int(): toward zero, as np.uint8(negative float) does
    return hsv_to_rgb_u8(hsv.astype(np.uint8))


def gauss_box_radius(radius, passes=3):
    """BoxBlur.c _gaussian_blur_radius: the (fractional) box radius whose `passes`-fold box blur has the Gaussian's variance."""
    radius = np.float32(radius)
    sigma2 = np.float32(radius * radius / np.float32(passes))
    L = np.float32(np.sqrt(12.0 * np.float64(sigma2) + 1.0))
    l_ = np.float32(np.floor((np.float64(L) - 1.0) / 2.0))
    a = np.float32((2 * l_ + 1) * (l_ * (l_ + 1) - 3 * sigma2))
    a = np.float32(a / np.float32(6 * (sigma2 - (l_ + 1) * (l_ + 1))))
    return np.float32(l_ + a)


def box_blur_pass_h(img, fr):
    """One horizontal pass of ImagingLineBoxBlur8 in direct form: 24-bit fixed point, edge pixels replicated."""
    r = int(fr)
    ww = int(np.uint32(np.float32(1 << 24) / np.float32(np.float32(fr) * 2 + 1)))
    fw = ((1 << 24) - (r * 2 + 1) * ww) // 2
    W = img.shape[1]
    x = np.arange(W)
    acc = np.zeros(img.shape, np.int64)
    for k in range(-r, r + 1):
        acc += img[:, np.clip(x + k, 0, W - 1)].astype(np.int64)
    far = img[:, np.clip(x - r - 1, 0, W - 1)].astype(np.int64) + img[:, np.clip(x + r + 1, 0, W - 1)].astype(np.int64)
    bulk = (acc * ww + far * fw) & 0xffffffff
    return (((bulk + (1 << 23)) & 0xffffffff) >> 24).astype(np.uint8)


def gaussian_blur(img, radius):
    """ImageFilter.GaussianBlur(radius): three horizontal then three vertical box passes."""
    fr = gauss_box_radius(radius)
    out = img
    if fr != 0:
        for _ in range(3):
            out = box_blur_pass_h(out, fr)
        out = out.transpose(1, 0, 2)
        for _ in range(3):
            out = box_blur_pass_h(out, fr)
        out = np.ascontiguousarray(out.transpose(1, 0, 2))
    return out


# RandAug's histogram / filter ops on a no-box image (mmdet/datasets/pipelines/autoaug_fast.py:219-224 auto_contrast / equalize,
# :244-250 posterize, :371-372 solarize, :407 sharpness; reached from semi_aug.py:494-497): Pillow's ImageOps / ImageEnhance
# arithmetic, restated from PIL/ImageOps.py and Filter.c / Blend.c and PINNED: tests/golden/randaug_pil.npz holds Pillow's own outputs
# (tests/golden/make_golden.py randaug).  Every op acts per band, so the stored channel order is immaterial.
def band_histogram(img):
    """Image.histogram(): 256 bins per band."""
    return np.stack([np.bincount(img[..., c].reshape(-1), minlength=256) for c in range(img.shape[-1])]).astype(np.int64)


def autocontrast_lut(h):
    """ImageOps.autocontrast(img) with cutoff 0, one band's histogram h[256] -> lut[256]: the darkest value present maps to 0, the
    lightest to 255; int() of the DOUBLE ix * scale + offset, clamped."""
    nz = np.nonzero(h)[0]
    if nz.size == 0:
        return np.arange(256, dtype=np.uint8)
    lo, hi = int(nz[0]), int(nz[-1])
    if hi <= lo:
        return np.arange(256, dtype=np.uint8)
    scale = 255.0 / (hi - lo)
    offset = -lo * scale
    return np.clip((np.arange(256, dtype=np.float64) * scale + offset).astype(np.int64), 0, 255).astype(np.uint8)


def equalize_lut(h):
    """ImageOps.equalize(img), one band: step = (pixels - count of the last occupied bin) // 255; lut[i] = (step // 2 + sum(h[:i]))
    // step, identity when fewer than two bins are occupied or step is 0; Image.point clips the table to 8 bits."""
    histo = h[h > 0]
    if histo.size <= 1:
        return np.arange(256, dtype=np.uint8)
    step = (int(histo.sum()) - int(histo[-1])) // 255
    if step == 0:
        return np.arange(256, dtype=np.uint8)
    n = step // 2 + np.concatenate([[0], np.cumsum(h)[:-1]])
    return np.clip(n // step, 0, 255).astype(np.uint8)


def autocontrast(img):
    h = band_histogram(img)
    return np.stack([autocontrast_lut(h[c])[img[..., c]] for c in range(3)], -1)


def equalize(img):
    h = band_histogram(img)
    return np.stack([equalize_lut(h[c])[img[..., c]] for c in range(3)], -1)


def solarize(img, threshold):
    """ImageOps.solarize: values >= threshold are inverted."""
    return np.where(img.astype(np.int64) < threshold, img, 255 - img).astype(np.uint8)


def posterize(img, bits):
    """ImageOps.posterize: keep the `bits` high bits of every channel."""
    return (img & np.uint8((~(2 ** (8 - bits) - 1)) & 255)).astype(np.uint8)


def smooth3x3(img):
    """img.filter(ImageFilter.SMOOTH) (Filter.c ImagingFilter3x3, 8-bit bands): kernel (1 1 1 / 1 5 1 / 1 1 1) / 13 as float32 weights,
    float32 accumulation 0.5 + row(y + 1) + row(y) + row(y - 1) with each row's three products summed left to right first, truncated;
    the one-pixel frame is copied from the input."""
    k1, k5 = np.float32(1.0) / np.float32(13.0), np.float32(5.0) / np.float32(13.0)
    f = img.astype(np.float32)
    out = img.copy()
    H, W = img.shape[:2]
    if H < 3 or W < 3:
        return out
    def row(r, kc):
        return (r[:, :-2] * k1 + r[:, 1:-1] * kc).astype(np.float32) + r[:, 2:] * k1
    ss = np.float32(0.5) + row(f[2:], k1)
    ss = (ss + row(f[1:-1], k5)).astype(np.float32)
    ss = (ss + row(f[:-2], k1)).astype(np.float32)
    out[1:-1, 1:-1] = np.clip(ss, 0, 255).astype(np.uint8)
    return out


def adjust_sharpness(img, f):
    """ImageEnhance.Sharpness(img).enhance(f): blend(SMOOTH-filtered image, image, f)."""
    return pil_blend(smooth3x3(img), img, f)


def randaug_level(op, level):
    """autoaug_fast.py's level -> parameter maps (PARAMETER_MAX = 10): Solarize threshold 256 - int(level * 256 / 10) (:371-372),
    Posterize bits 4 - int(level * 4 / 10) (:244-247), the enhancers' factor level * 1.8 / 10 + 0.1 (:392-399)."""
    if op == 'Solarize':
        return 256 - int(level * 256 / 10)
    if op == 'Posterize':
        return 4 - int(level * 4 / 10)
    return float(level) * 1.8 / 10 + 0.1


def erase_value(z):
    """ToPILImage's mul(255).byte() of the N(0, 1) fill RandomErasing(value='random') writes: toward zero, wrap mod 256."""
    return (np.trunc(np.asarray(z, np.float32) * np.float32(255)).astype(np.int64) & 255).astype(np.uint8)


def affine_matrix(kind, value, w, h):
    """Forward 3x3 matrix (input pixel -> output pixel) of ONE imgaug Affine child of AFFINE_TRANSFORM (semi_aug.py:36-62): the
    transform acts about the image centre (w / 2 - 0.5, h / 2 - 0.5) (imgaug 0.4 `_AffineMatrixGenerator`); rotation and shear in
    degrees, positive rotation clockwise in image coordinates; shear moves x with y.  UNPINNED (imgaug is not installable here)."""
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


def warp_affine_u8(img, M_fwd, order, cval=125):
    """Inverse-map warp of a uint8 [H, W, C] image: order 0 nearest (round half to even), 1 bilinear in float32; constant border."""
    h, w = img.shape[:2]
    Mi = np.linalg.inv(M_fwd).astype(np.float32)
    ys, xs = np.mgrid[0:h, 0:w].astype(np.float32)
    sx = Mi[0, 0] * xs + Mi[0, 1] * ys + Mi[0, 2]
    sy = Mi[1, 0] * xs + Mi[1, 1] * ys + Mi[1, 2]

    def at(xi, yi):
        ok = (xi >= 0) & (yi >= 0) & (xi < w) & (yi < h)
        v = img[np.clip(yi, 0, h - 1), np.clip(xi, 0, w - 1)].astype(np.float32)
        return np.where(ok[..., None], v, np.float32(cval))
    if order == 0:
        return at(np.rint(sx).astype(np.int64), np.rint(sy).astype(np.int64)).astype(np.uint8)
    x0, y0 = np.floor(sx), np.floor(sy)
    fx, fy = (sx - x0)[..., None], (sy - y0)[..., None]
    x0, y0 = x0.astype(np.int64), y0.astype(np.int64)
    top = at(x0, y0) * (1 - fx) + at(x0 + 1, y0) * fx
    bot = at(x0, y0 + 1) * (1 - fx) + at(x0 + 1, y0 + 1) * fx
    return np.clip(np.rint(top * (1 - fy) + bot * fy), 0, 255).astype(np.uint8)


def affine_boxes(boxes, M_fwd, w, h):
    """imgaug BoundingBoxesOnImage under an affine map: the 4 corners are moved, the new box is their axis-aligned hull; then
    semi_aug.py:519-527: clip to the image, drop boxes of zero area.  Returns (boxes, keep mask)."""
    b = np.asarray(boxes, np.float64).reshape(-1, 4)
    out = np.zeros_like(b)
    for i, (x1, y1, x2, y2) in enumerate(b):
        c = np.array([[x1, y1, 1], [x2, y1, 1], [x2, y2, 1], [x1, y2, 1]], np.float64) @ M_fwd.T
        out[i] = [c[:, 0].min(), c[:, 1].min(), c[:, 0].max(), c[:, 1].max()]
    out[:, 0::2] = np.clip(out[:, 0::2], 0, w)
    out[:, 1::2] = np.clip(out[:, 1::2], 0, h)
    keep = ((out[:, 2] - out[:, 0]) * (out[:, 3] - out[:, 1])) > 0
    return out[keep].astype(np.float32), keep
