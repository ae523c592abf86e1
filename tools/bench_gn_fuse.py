"""Standalone cost of the head's conv -> GN -> ReLU layer and of its backward (data gradient + GroupNorm backward), with the
GroupNorm block records written by the convolution's epilogue (dsl_conv_desc.gn_ws / gn_x) and without: full size, N = 2."""
import ctypes as C
import sys
import torch
sys.path.insert(0, '.')
from dsl_amd import _lib as L, ops

LV = [(100, 168), (50, 84), (25, 42), (13, 21), (7, 11)]
N, Cc = 2, 256
P = sum(h * w for h, w in LV) * N
g = torch.Generator().manual_seed(0)
rb = lambda *s: (torch.randn(*s, generator=g)).bfloat16().cuda()
x, gnext = rb(P, Cc), rb(P, Cc)
w = (torch.randn(Cc, 9 * Cc, generator=g) * 0.03).bfloat16().cuda()
ga, be = torch.ones(Cc).cuda(), torch.zeros(Cc).cuda()
pre, y, dy, dx = (torch.empty(P, Cc, dtype=torch.bfloat16, device='cuda') for _ in range(4))
stats = torch.empty(5 * N * 32, 2, device='cuda')
dgam, dbet, dbias = (torch.empty(Cc, device='cuda') for _ in range(3))


def timeit(fn, it=60):
    for _ in range(8):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / it * 1e3


for fused in (0, 1):
    gd = ops.gn_desc(pre, y, ga, be, stats, n=N, hw=LV)
    cd = ops.conv_desc(x, w, pre, n=N, grid=LV, src_hw=LV, dst_hw=LV, cs=Cc, cd=Cc, cd_pad=Cc, ldd=Cc, kh=3, kw=3, stride=1, pad=1)
    if fused:
        assert L.lib.dsl_conv2d_gn_fusable(C.byref(cd))
        cd.gn_ws, gd.conv_stats = gd.workspace, 1
    s = L.stream_ptr()
    tc = timeit(lambda: L.lib.dsl_conv2d(C.byref(cd), s))
    tg = timeit(lambda: L.lib.dsl_groupnorm_relu_fwd(C.byref(gd), s))
    tb = timeit(lambda: (L.lib.dsl_conv2d(C.byref(cd), s), L.lib.dsl_groupnorm_relu_fwd(C.byref(gd), s)))
    print(f'forward  fused={fused}: conv {tc:6.1f} us, GroupNorm {tg:6.1f} us, both {tb:6.1f} us')
    gb = ops.gn_desc(pre, y, ga, be, stats, n=N, hw=LV, dy=dy, dx=dx, dgamma=dgam, dbeta=dbet, dbias=dbias)
    cb = ops.conv_desc(gnext, w, dy, n=N, grid=LV, src_hw=LV, dst_hw=LV, cs=Cc, cd=Cc, cd_pad=Cc, ldd=Cc, kh=3, kw=3, stride=1, pad=1, mode=1)
    if fused:
        cb.gn_x = L.ptr(pre)
        assert L.lib.dsl_conv2d_gn_fusable(C.byref(cb))
        cb.gn_ws, cb.gn_gamma, cb.gn_beta, cb.gn_stats = gb.workspace, L.ptr(ga), L.ptr(be), L.ptr(stats)
        gb.conv_stats = 1
    tc = timeit(lambda: L.lib.dsl_conv2d(C.byref(cb), s))
    tg = timeit(lambda: L.lib.dsl_groupnorm_relu_bwd(C.byref(gb), s))
    tb = timeit(lambda: (L.lib.dsl_conv2d(C.byref(cb), s), L.lib.dsl_groupnorm_relu_bwd(C.byref(gb), s)))
    print(f'backward fused={fused}: dgrad {tc:6.1f} us, GroupNorm {tg:6.1f} us, both {tb:6.1f} us')
