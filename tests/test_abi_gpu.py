"""Boundary contract on the device (include/dsl_hip.h "Conventions"): the caller owns every buffer, the library allocates no device
memory - checked with hipMemGetInfo around training steps at shapes the process has not seen before (until
round 6 the weight-gradient kernels' pixel tables were hipMalloc'd by the library per new geometry and never freed)."""
import ctypes as C

import numpy as np
import pytest
import torch

from util import fcos_model_cfg

pytestmark = pytest.mark.gpu
T = torch.from_numpy


def _foreign_bytes():
    """Device memory in use that torch's caching allocator does not hold: HIP runtime + whatever a library allocated itself."""
    torch.cuda.synchronize()
    free, total = torch.cuda.mem_get_info()
    return total - free - torch.cuda.memory_reserved()


def test_no_device_allocation_by_the_library_over_multi_scale_steps():
    from dsl_amd import detectors  # noqa: F401
    from dsl_amd.optim import FlatSGD
    from dsl_amd.registry import build_detector
    from oracle import fcos_oracle as O
    model = build_detector(fcos_model_cfg())
    model.load_state_dict(O.synth_state_dict(0))
    model = model.cuda()
    opt = FlatSGD(model, lr=0.01, momentum=0.9, weight_decay=1e-4, paramwise_cfg=dict(bias_lr_mult=2., bias_decay_mult=0.))
    rng = np.random.RandomState(5)

    def step(h, w):
        img = torch.randn(2, 3, h, w, device='cuda') * 40
        gtb = [T(O.synth_boxes(rng, 3, H=h, W=w, lo=8, hi=min(h, w))) for _ in range(2)]
        gtl = [T(rng.randint(0, 80, len(b)).astype('int64')) for b in gtb]
        out = model.train_step(dict(img=img, img_metas=[dict(img_shape=(h, w, 3), pad_shape=(h, w, 3), scale_factor=1.0)] * 2,
                                    gt_bboxes=gtb, gt_labels=gtl), opt)
        out['loss'].backward()
        opt.step()
        return float(out['loss'])

    # warm-up at one shape: code objects, the library's streams and events, torch's own pools
    for _ in range(2):
        step(256, 320)
    before = _foreign_bytes()
    # multi-scale training (img_scale ranges): 20 steps over shapes this process has not seen - every one brings new weight-gradient
    # geometries (new pixel tables), new plans, new workspaces: all of it through torch's allocator, i.e. the CALLER's memory
    shapes = [(256 + 32 * (i % 5), 320 + 32 * (i % 4)) for i in range(20)]
    vals = [step(h, w) for h, w in shapes]
    assert all(np.isfinite(v) for v in vals)
    after = _foreign_bytes()
    print('device memory not held by the caller: before', before, 'after', after, 'delta', after - before)
    assert after - before <= 0, (before, after)


def test_wgrad_without_its_pixel_table_is_refused():
    """dsl_wgrad_desc.pixtab is required where dsl_wgrad_pixtab_bytes() > 0: a launch without it fails loudly (no silent allocation)."""
    from dsl_amd import _lib as L
    from dsl_amd import ops
    n, h, w, ci, co = 1, 32, 48, 256, 256
    dy = torch.randn(n, h, w, co, device='cuda').bfloat16()
    x = torch.randn(n, h, w, ci, device='cuda').bfloat16()
    dw = torch.zeros(co, 9 * ci, device='cuda')
    d = ops.wgrad_desc(dy, x, dw, n=n, grid=[(h, w)], src_hw=[(h, w)], cs=ci, cy=co, cd=co, kh=3, kw=3, stride=1, pad=1)
    need = L.lib.dsl_wgrad_pixtab_bytes(C.byref(d))
    assert need == n * h * w * 8 and d.pixtab and d.pixtab_bytes >= need
    L.check(L.lib.dsl_conv2d_wgrad(C.byref(d), L.stream_ptr()), 'dsl_conv2d_wgrad')
    torch.cuda.synchronize()
    ref = dw.clone()
    assert float(ref.abs().max()) > 0
    # the table's contents: base pixel index of tap (0, 0) and the validity bits, as the header documents them
    tab = d._keep[-1].view(torch.int32).view(-1, 2).cpu()
    assert tab.shape[0] == n * h * w
    y, xx = 5, 0
    e = tab[y * w + xx]
    assert int(e[0]) == (y - 1) * w + (xx - 1)
    assert (int(e[1]) & 0xffffffff) == ((w << 16) | (0b110 << 8) | 0b111)
    saved = (d.pixtab, d.pixtab_bytes)
    d.pixtab, d.pixtab_bytes = None, 0
    rc = L.lib.dsl_conv2d_wgrad(C.byref(d), L.stream_ptr())
    assert rc != 0 and b'pixtab' in L.lib.dsl_last_error()
    d.pixtab, d.pixtab_bytes = saved
    dw.zero_()
    L.check(L.lib.dsl_conv2d_wgrad(C.byref(d), L.stream_ptr()), 'dsl_conv2d_wgrad')
    torch.cuda.synchronize()
    assert torch.equal(dw, ref)
