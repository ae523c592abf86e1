import os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from dsl_amd import _lib as L
from dsl_amd import ops
def bf(t): return t.bfloat16().float()
def nhwc(t): return t.permute(0, 2, 3, 1).contiguous().bfloat16().cuda()
g = torch.Generator().manual_seed(13)
N, Ci, Co, H, W, k, s, p = 1, 128, 256, 13, 21, 3, 2, 1
x = bf(torch.randn(N, Ci, H, W, generator=g)); w = bf(torch.randn(Co, Ci, k, k, generator=g)).requires_grad_()
y = F.conv2d(x, w, None, s, p); Ho, Wo = y.shape[2:]
dy = bf(torch.randn(N, Co, Ho, Wo, generator=g)); y.backward(dy)
ref = w.grad.permute(0, 2, 3, 1)
for cfg in (2,):
    dw = torch.empty(Co, k, k, Ci, dtype=torch.float32, device='cuda')
    ops.conv2d_wgrad(nhwc(dy), nhwc(x), dw, n=N, grid=[(Ho, Wo)], src_hw=[(H, W)], cs=Ci, cy=Co, cd=Co, kh=k, kw=k, stride=s, pad=p, force_cfg=cfg)
    torch.cuda.synchronize()
    got = dw.cpu()
    bad = ~torch.isclose(got, ref, rtol=1e-2, atol=0.05 * float(ref.abs().max()))
    print('cfg', cfg, 'bad', int(bad.sum()), 'of', bad.numel())
    idx = bad.nonzero()
    import collections
    print('bad co:', sorted(set(idx[:, 0].tolist()))[:40])
    print('bad taps:', collections.Counter((idx[:, 1] * 3 + idx[:, 2]).tolist()))
    print('bad ci:', sorted(set(idx[:, 3].tolist()))[:70])
    print('sample', [(tuple(i.tolist()), float(got[tuple(i.tolist())]), float(ref[tuple(i.tolist())])) for i in idx[:6]])
