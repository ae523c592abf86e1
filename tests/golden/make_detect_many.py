"""Fixture of tests/test_sweep_gpu.py::test_detect_with_more_valid_pairs_than_the_candidate_buffer: the ORACLE's uncapped
get_bboxes / multiclass_nms (oracle/fcos_oracle.py, itself pinned to the reference's outputs by tests/test_oracle_golden.py) on a
seeded head output with ~80 000 valid (location, class) pairs per image.  The pure-Python greedy NMS over that many candidates takes
20 s here and minutes on a loaded GPU box, so its result is committed: python tests/golden/make_detect_many.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import fcos_oracle as O  # noqa: E402

SIZES = [(24, 32), (12, 16), (6, 8), (3, 4), (2, 2)]
STRIDES = (8, 16, 32, 64, 128)
B, SHAPE = 2, (192, 256)


def inputs():
    g = torch.Generator().manual_seed(12)
    cls = [torch.randn(B, 80, h, w, generator=g) for h, w in SIZES]
    reg = [torch.exp(torch.randn(B, 4, h, w, generator=g) * 0.5 + 1.2) * s for (h, w), s in zip(SIZES, STRIDES)]
    ctr = [torch.randn(B, 1, h, w, generator=g) for h, w in SIZES]
    return cls, reg, ctr


def digest(ts):
    return float(sum(float(t.double().abs().sum()) for t in ts))


if __name__ == '__main__':
    cls, reg, ctr = inputs()
    ref = O.get_bboxes(cls, reg, ctr, SHAPE, [[1.0, 1.0, 1.0, 1.0]] * B, nms_pre=1000, score_thr=0.05, iou_thr=0.5, max_per_img=100,
                       rescale=True, strides=STRIDES)
    out = dict(digest=np.float64(digest(cls + reg + ctr)))
    for i, (b, l) in enumerate(ref):
        out[f'det{i}'], out[f'lab{i}'] = b.numpy(), l.numpy()
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'detect_many.npz'), **out)
    print('wrote detect_many.npz', {k: getattr(v, 'shape', v) for k, v in out.items()})
