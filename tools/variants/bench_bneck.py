"""Standalone timing of the frozen layer1 (three bottlenecks at 200 x 336 x N): the fused launches (dsl_bottleneck64) against the ten
dsl_conv2d launches they replace, back-to-back replays of the engine's own prefix op list.  Usage: python tools/bench_bneck.py [N]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from dsl_amd import detectors  # noqa: F401
from dsl_amd import _lib as L
from dsl_amd.registry import build_detector
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2
res = {}
for fused in ('1', '0'):
    os.environ['DSL_BNECK64'] = fused
    model = build_detector(bench.model_cfg()).cuda()
    eng = model._get_engine()
    plan = eng.plan(model.store, N, 800, 1344, training=True)
    ops_ = [o for o in plan.prefix.items if o.kind in (L.OP_CONV, L.OP_BNECK64, L.OP_FORK, L.OP_JOIN)]
    from dsl_amd.engine import OpList
    ol = OpList(); ol.items = ops_; ol.keep = plan.prefix.keep
    for _ in range(5): ol.run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    R = 50
    for _ in range(R): ol.run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / R * 1e6
    nk = sum(1 for o in ops_ if o.kind in (L.OP_CONV, L.OP_BNECK64))
    res[fused] = (dt, nk, plan._l1out[plan._parity].float().abs().mean().item())
    print(f'DSL_BNECK64={fused}: layer1 {dt:.1f} us per pass, {nk} launches, mean|out| {res[fused][2]:.5f}')
    del model, eng, plan
    torch.cuda.empty_cache()
