"""Which CUs does a CU-masked HIP stream run on?  (hipExtStreamCreateWithCUMask, 256 mask bits on MI355X)
Prints, for a few masks, the set of (XCC, SE, SH, CU) a grid of one-per-CU workgroups landed on."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from dsl_amd import _lib as L


def run(bits, nblocks=1024):
    out = torch.zeros(nblocks, 2, dtype=torch.int32, device='cuda')
    if bits is None:
        L.check(L.lib.dsl_probe_cu_mask(None, 0, L.ptr(out), nblocks), 'probe')
    else:
        m = (C.c_uint32 * 8)()
        for b in bits:
            m[b >> 5] |= 1 << (b & 31)
        L.check(L.lib.dsl_probe_cu_mask(m, 8, L.ptr(out), nblocks), 'probe')
    o = out.cpu().numpy()
    xcc, hw = o[:, 0], o[:, 1]
    cu, sh, se = (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7
    ids = sorted(set(zip(xcc.tolist(), se.tolist(), sh.tolist(), cu.tolist())))
    per = {}
    for x, *_ in ids:
        per[x] = per.get(x, 0) + 1
    return ids, per


if __name__ == '__main__':
    torch.zeros(1, device='cuda')
    for name, bits in (('none', None), ('bits 0..95', range(96)), ('bits i%8<3', [b for b in range(256) if b % 8 < 3]),
                       ('bits i//8<12', [b for b in range(256) if b // 8 < 12]), ('bits 0..7', range(8)),
                       ('bits 0,8,16,..', range(0, 256, 8))):
        ids, per = run(bits)
        print(f'{name:16s}: {len(ids)} distinct CUs; per XCC {per}')
        if len(ids) <= 32:
            print('    ', ids)
