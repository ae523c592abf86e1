"""Why is the SECOND model a process builds sometimes 15 % slower (profiles/r04_td_first.txt)?  Two models one after the other in one
process, the first released in between; argv[1]: 0 = just drop the reference, 1 = gc + torch.cuda.empty_cache() in between,
2 = keep the first model alive."""
import gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from dsl_amd.data import mark_ready
from dsl_amd.optim import FlatSGD
from dsl_amd.registry import build_detector
from dsl_amd import detectors  # noqa: F401
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 0
WINDOWS = int(sys.argv[2]) if len(sys.argv) > 2 else 1
batch = bench.synth_batch(0, 2)
ev = torch.cuda.Event(); ev.record()


def run():
    model = build_detector(bench.model_cfg()).cuda()
    opt = FlatSGD(model, lr=0.01, momentum=0.9, weight_decay=1e-4, paramwise_cfg=dict(bias_lr_mult=2., bias_decay_mult=0.))
    def step():
        mark_ready(batch['img'], event=ev); out = model.train_step(batch, opt); out['loss'].backward(); opt.step()
    for _ in range(8): step()
    wins = []
    for _ in range(WINDOWS):          # several timed windows with a device synchronisation between them: does a slow model stay slow?
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(40): step()
        torch.cuda.synchronize(); wins.append(round(2 / ((time.perf_counter() - t0) / 40), 1))
    return wins if WINDOWS > 1 else wins[0], model, opt


res = []
keep = []
for i in range(3):
    v, m, o = run()
    res.append(v)
    if mode == 2:
        keep.append((m, o))
    del m, o
    if mode == 1:
        gc.collect(); torch.cuda.synchronize(); torch.cuda.empty_cache()
print('mode', mode, 'img/s of three models built one after the other:', res, 'reserved MB', torch.cuda.memory_reserved() >> 20)
