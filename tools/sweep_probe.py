"""Times the teacher sweep of the pseudo-label refresh alone (batch-1 eval forward + detection post-processing + fuse) and,
under rocprofv3 (tools/exp_prof.sh style), gives its kernel sequence: `python tools/sweep_probe.py [rla] [n_img]`."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import dsl_amd.detectors  # noqa: F401
from dsl_amd.registry import build_detector
from dsl_amd.sweep import detect_device

rla = 'rla' in sys.argv[1:]
n = int(sys.argv[-1]) if sys.argv[-1].isdigit() else 1
model = build_detector(bench.model_cfg(dsl=True, rla=rla)).cuda()
model.eval()
img = torch.randn(n, 3, 800, 1344, device='cuda') * 50
metas = [dict(img_shape=(800, 1333, 3), scale_factor=1.0)] * n
for _ in range(5):
    detect_device(model, img, metas, rescale=True)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(21)]
ev[0].record()
for i in range(20):
    detect_device(model, img, metas, rescale=True)
    ev[i + 1].record()
torch.cuda.synchronize()
gaps = sorted(a.elapsed_time(b) for a, b in zip(ev[:-1], ev[1:]))
print(f'sweep rla={rla} n={n}: median {gaps[10]:.3f} ms (min {gaps[0]:.3f}, max {gaps[-1]:.3f})')
