"""Times the head-tower conv (v3 kernel) under the ablation knobs.  Needs tools/build_ablate.sh first.
Re-executes itself per knob because DSL_ABLATE is read at launch."""
import os
import subprocess
import sys
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
if len(sys.argv) > 1 and sys.argv[1] == 'child':
    sys.path.insert(0, ROOT)
    import ctypes as C
    import torch
    from dsl_amd import _lib as L
    from dsl_amd import ops
    N = 2
    LEVELS = [(100, 168), (50, 84), (25, 42), (13, 21), (7, 11)]
    shapes = {'head': (256, 256, 3, LEVELS), 'fpn': (256, 256, 3, LEVELS[:1]), 'l3': (256, 256, 3, [(50, 84)]),
              'l2': (128, 128, 3, [(100, 168)]), 'l4': (512, 512, 3, [(25, 42)]), 'l2b': (128, 512, 1, [(100, 168)]),
              'l3b': (1024, 256, 1, [(50, 84)])}
    for which in sys.argv[2:]:
        ci, co, k, lv = shapes[which]
        P = sum(h * w for h, w in lv) * N
        x = torch.randn(P, ci, device='cuda').bfloat16()
        w = (torch.randn(co, k, k, ci, device='cuda') * 0.05).bfloat16()
        y = torch.empty(P, co, device='cuda', dtype=torch.bfloat16)
        ws = torch.empty(128 << 20, dtype=torch.uint8, device='cuda')
        d = ops.conv_desc(x, w, y, n=N, grid=lv, src_hw=lv, dst_hw=lv, cs=ci, cd=co, cd_pad=co, ldd=co, kh=k, kw=k,
                          stride=1, pad=k // 2, flags=L.CONV_RELU_OUT, workspace=ws)
        for _ in range(3):
            L.lib.dsl_conv2d(C.byref(d), L.stream_ptr())
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            L.lib.dsl_conv2d(C.byref(d), L.stream_ptr())
        e1.record()
        torch.cuda.synchronize()
        print(f'  {which:6s} {e0.elapsed_time(e1) / 20 * 1e3:7.1f} us', end='')
    print()
else:
    env = dict(os.environ, DSL_HIP_LIB=os.path.join(ROOT, 'dsl_amd', 'lib', 'libdsl_hip_ablate.so'))
    names = {0: 'full', 1: 'no pixel DMA', 2: 'no weight DMA', 3: 'no DMA', 4: 'no MFMA', 5: 'weights DMA only', 6: 'pixel DMA only', 7: 'nothing', 8: 'one K tile', 24: 'one K tile, no epilogue', 16: 'no epilogue', 23: 'LDS reads+barriers only', 19: 'MFMA+LDS, no epilogue', 51: 'MFMA+LDS, no barrier, no epi', 83: 'MFMA+barrier, no LDS reads, no epi', 115: 'MFMA only, no epi', 128: 'launch + dispatch only', 256: 'launch + decode prologue', 16: 'no epilogue', 48: 'no barrier, no epi', 80: 'no LDS reads, no epi'}
    for knob in ([int(a) for a in os.environ['KNOBS'].split(',')] if 'KNOBS' in os.environ else (0, 4, 1, 2, 3, 5, 6, 7, 8, 24, 16, 23)):
        env['DSL_ABLATE'] = str(knob)
        print(f'{names.get(knob, str(knob)):26s}', flush=True)
        subprocess.run([sys.executable, __file__, 'child'] + (sys.argv[1:] or ['head', 'fpn', 'l3']), env=env)
