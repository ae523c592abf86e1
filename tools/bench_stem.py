"""Times the frozen stem at the benchmark's shape (N x 3 x 800 x 1344): dsl_stem_pool (one kernel) against the three launches it
replaced (dsl_pack_image, dsl_conv2d small-C, dsl_maxpool3x3s2)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dsl_amd import _lib as L
from dsl_amd import ops

N, H, W = int(sys.argv[1]) if len(sys.argv) > 1 else 2, 800, 1344
g = torch.Generator().manual_seed(0)
x = (torch.randn(N, 3, H, W, generator=g) * 50).cuda()
w = torch.randn(64, 3, 7, 7, generator=g) * 0.05
sc, bi = (torch.rand(64, generator=g) + 0.5).cuda(), torch.randn(64, generator=g).cuda()
wg = torch.zeros(64, 7, 24)
wg[:, :, :21] = w.permute(0, 2, 3, 1).reshape(64, 7, 21)
wg = torch.cat([wg.reshape(64, 21, 8).permute(1, 0, 2), torch.zeros(1, 64, 8)], 0).bfloat16().cuda().contiguous()
Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
PH, PW = (Ho - 1) // 2 + 1, (Wo - 1) // 2 + 1
out = torch.empty(N, PH, PW, 64, dtype=torch.bfloat16, device='cuda')
x8 = torch.empty(N, H, W, 8, dtype=torch.bfloat16, device='cuda')
y = torch.empty(N, Ho, Wo, 64, dtype=torch.bfloat16, device='cuda')
wp = torch.zeros(64, 7 * 64)
wp[:, :392] = torch.cat([w.permute(0, 2, 3, 1), torch.zeros(64, 7, 7, 5)], -1).reshape(64, 392)
wp = wp.bfloat16().cuda()
cd = ops.conv_desc(x8, wp, y, n=N, grid=[(Ho, Wo)], src_hw=[(H, W)], dst_hw=[(Ho, Wo)], cs=8, cd=64, cd_pad=64, ldd=64, kh=7, kw=7,
                   stride=2, pad=3, flags=L.CONV_RELU_OUT | L.CONV_SMALL_C, scale=sc, bias=bi)


def fused():
    L.lib.dsl_stem_pool(L.ptr(x), L.ptr(wg), L.ptr(sc), L.ptr(bi), L.ptr(out), 64, N, H, W, L.stream_ptr())


def three():
    import ctypes as C
    L.lib.dsl_pack_image(L.ptr(x), L.ptr(x8), N, H, W, L.stream_ptr())
    L.lib.dsl_conv2d(C.byref(cd), L.stream_ptr())
    L.lib.dsl_maxpool3x3s2(L.ptr(y), L.ptr(out), N, Ho, Wo, 64, L.stream_ptr())


def timeit(fn, it=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3


tf, t3 = timeit(fused), timeit(three)
gflop = 2 * N * Ho * Wo * 64 * 147 / 1e9
mb_f = (N * 3 * H * W * 4 + N * PH * PW * 64 * 2) / 1e6
print(f'N={N}: fused {tf:.1f} us ({gflop / tf * 1e3:.0f} TFLOP/s of the 147-product convolution, {mb_f / tf * 1e3:.0f} GB/s on its '
      f'{mb_f:.0f} MB)   three launches {t3:.1f} us')
