import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np, torch
from util import fcos_model_cfg, levels_to_flat, rel_l2
from dsl_amd import detectors
from dsl_amd.registry import build_detector
from oracle import fcos_oracle as O
T = torch.from_numpy
H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (64, 96)
model = build_detector(fcos_model_cfg()); model.load_state_dict(O.synth_state_dict(0)); model = model.cuda()
B = 2
rng = np.random.RandomState(1)
g = torch.Generator().manual_seed(3)
img = torch.randn(B, 3, H, W, generator=g) * 40
gtb = [T(O.synth_boxes(rng, 4, H=H, W=W, lo=8, hi=min(H, W))) for _ in range(B)]
gtl = [T(rng.randint(0, 80, len(b)).astype('int64')) for b in gtb]
metas = [dict()] * B
losses = model.forward_train(img.cuda(), metas, gtb, gtl)
sum(losses.values()).backward(); torch.cuda.synchronize()
sd = O.synth_state_dict(0)
l32, g32, _ = O.train_step(sd, img, gtb, gtl, None, emulate_bf16=False)
l16, g16, _ = O.train_step(sd, img, gtb, gtl, None, emulate_bf16=True)
print('losses hip ', {k: float(v) for k, v in losses.items()})
print('losses emu ', l16)
print('losses fp32', l32)
named = dict(model.named_parameters())
print(f'{"key":50s} hip-vs-fp32  emu-vs-fp32  hip-vs-emu')
for k in O.trainable_keys(sd):
    m = named[k].grad.cpu()
    print(f'{k:50s} {rel_l2(m, g32[k]):.4f}       {rel_l2(g16[k], g32[k]):.4f}       {rel_l2(m, g16[k]):.4f}')
