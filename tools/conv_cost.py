"""Per-launch cost of a backbone conv shape under the ablation knobs, without host launch overhead in the way: 64 identical
launches replayed by dsl_run_ops (C loop) per timing.  Needs tools/build_ablate.sh.
  KNOBS=0,128,256,8,24,16 python tools/conv_cost.py l3a l3b l3c      (shapes below; suffix 1 = one image)"""
import os
import subprocess
import sys
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SHAPES = {'p3': (256, 256, 3, (100, 168)), 'l3a': (1024, 256, 1, (50, 84)), 'l3b': (256, 256, 3, (50, 84)), 'l3c': (256, 1024, 1, (50, 84)),
          'l2a': (512, 128, 1, (100, 168)), 'l2b': (128, 128, 3, (100, 168)), 'l2c': (128, 512, 1, (100, 168)),
          'l4a': (2048, 512, 1, (25, 42)), 'l4b': (512, 512, 3, (25, 42)), 'l4c': (512, 2048, 1, (25, 42)),
          # weight-row stride experiments (row stride = K * 2 bytes): 960 / 1088 / 1984 input channels vs 1024 / 2048
          'x960': (960, 256, 1, (50, 84)), 'x1088': (1088, 256, 1, (50, 84)), 'x1984': (1984, 512, 1, (25, 42)), 'x2112': (2112, 512, 1, (25, 42)),
          'y192': (192, 256, 3, (50, 84)), 'y320': (320, 256, 3, (50, 84))}
if len(sys.argv) > 1 and sys.argv[1] == 'child':
    sys.path.insert(0, ROOT)
    import ctypes as C
    import torch
    from dsl_amd import _lib as L
    from dsl_amd import ops
    from dsl_amd.engine import OpList
    for which in sys.argv[2:]:
        n = 1 if which.endswith('1') else 2
        ci, co, k, hw = SHAPES[which.rstrip('1')]
        P = hw[0] * hw[1] * n
        x = torch.randn(P, ci, device='cuda').bfloat16()
        w = (torch.randn(co, k, k, ci, device='cuda') * 0.05).bfloat16()
        y = torch.empty(P, co, device='cuda', dtype=torch.bfloat16)
        ws = torch.empty(128 << 20, dtype=torch.uint8, device='cuda')
        force = int(os.environ.get('FORCE', '0'))
        d = ops.conv_desc(x, w, y, n=n, grid=[hw], src_hw=[hw], dst_hw=[hw], cs=ci, cd=co, cd_pad=co, ldd=co, kh=k, kw=k,
                          stride=1, pad=k // 2, flags=L.CONV_RELU_OUT | (force << 8), workspace=ws)
        ol = OpList()
        for _ in range(64):
            ol.conv(d)
        ol.run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4):
            ol.run()
        e1.record()
        torch.cuda.synchronize()
        print(f'  {which:5s} {e0.elapsed_time(e1) / 256 * 1e3:6.1f}', end='')
    print()
else:
    env = dict(os.environ)
    if 'NOABL' not in os.environ:         # NOABL=1: the product library (knob 0 only makes sense)
        env['DSL_HIP_LIB'] = os.path.join(ROOT, 'dsl_amd', 'lib', 'libdsl_hip_ablate.so')
    names = {0: 'full', 128: 'launch + dispatch only', 256: 'launch + decode prologue', 8: 'one K tile', 24: 'one K tile, no epilogue',
             16: 'no epilogue', 3: 'no DMA', 4: 'no MFMA', 7: 'no DMA, no MFMA'}
    for knob in [int(a) for a in os.environ.get('KNOBS', '0,128,256,8,24,16,3,4').split(',')]:
        env['DSL_ABLATE'] = str(knob)
        print(f'{names.get(knob, str(knob)):26s}', end='', flush=True)
        subprocess.run([sys.executable, __file__, 'child'] + (sys.argv[1:] or ['l3a', 'l3a1', 'l3b', 'l3b1', 'l3c', 'l3c1']), env=env)
