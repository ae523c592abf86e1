"""When does the pipelined frozen prefix of step i+1 actually run, relative to step i's backward segments?  Timing events on the
caller's stream after every backward segment and on the prefix stream around the prefix's op list (un-profiled run)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from dsl_amd import detectors  # noqa: F401
from dsl_amd.data import mark_ready
from dsl_amd.optim import FlatSGD
from dsl_amd.registry import build_detector
model = build_detector(bench.model_cfg()).cuda()
model.lazy_log = True; model.eager_backward = True
opt = FlatSGD(model, lr=0.01, momentum=0.9, weight_decay=1e-4, paramwise_cfg=dict(bias_lr_mult=2., bias_decay_mult=0.))
batch = bench.synth_batch(0, 2)
_ev = torch.cuda.Event(); _ev.record()
def step():
    mark_ready(batch['img'], event=_ev); out = model.train_step(batch, opt); out['loss'].backward(); opt.step()
for _ in range(6): step()
torch.cuda.synchronize()
plan = model._prefix_plan
rec = []
segs = plan.bwd_segments
class Seg:
    def __init__(self, ol, k): self.ol, self.k = ol, k
    def run(self):
        self.ol.run(); e = torch.cuda.Event(enable_timing=True); e.record(); rec.append(('seg%d' % self.k, e))
plan.bwd_segments = [(Seg(ol, k), info) for k, (ol, info) in enumerate(segs)]
pre = plan.prefix
class Pre:
    def __getattr__(self, name): return getattr(pre, name)
    def run(self):
        from dsl_amd.engine import OpList
        w, rest = OpList(), OpList()
        w.items, rest.items = pre.items[:1], pre.items[1:]
        w.keep = rest.keep = pre.keep
        w.run()                                   # the WAIT on SLOT_TAIL alone
        a = torch.cuda.Event(enable_timing=True); a.record()
        rest.run()
        b = torch.cuda.Event(enable_timing=True); b.record()
        rec.append(('prefix_after_wait', a)); rec.append(('prefix_end', b))
plan.prefix = Pre()
t0 = torch.cuda.Event(enable_timing=True)
for i in range(4):
    if i == 1: t0.record()
    m = torch.cuda.Event(enable_timing=True); m.record(); rec.append(('step%d_start' % i, m))
    step()
torch.cuda.synchronize()
for name, e in rec:
    try: print('%-22s %8.3f ms' % (name, t0.elapsed_time(e)))
    except Exception as ex: print(name, 'n/a')
