"""Per-op timing table of the training-step plan: every conv / wgrad / GN op of the forward and backward lists is
replayed alone (20 reps, HIP events) and compared with its own floor max(flops / MFMA rate, bytes / HBM rate).
Usage (GPU box): python tools/op_table.py [imgs_per_gpu]"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from dsl_amd import _lib as L
from dsl_amd import detectors  # noqa: F401
from dsl_amd.optim import FlatSGD
from dsl_amd.registry import build_detector

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2
model = build_detector(bench.model_cfg()).cuda()
model.lazy_log = True
opt = FlatSGD(model, lr=0.01, momentum=0.9, weight_decay=1e-4, paramwise_cfg=dict(bias_lr_mult=2., bias_decay_mult=0.))
batch = bench.synth_batch(0, N)
for _ in range(2):
    out = model.train_step(batch, opt)
    out['loss'].backward()
    opt.step()
torch.cuda.synchronize()
plan = [p for p in model._engine.plans.values() if p.training][0]
MFMA, HBM = 1.25e15, 4.0e12       # half of dense bf16 peak, ~2/3 of achievable HBM


def timeit(op, reps=20):
    arr = (L.Op * 1)(op)
    saved = op.i[6]
    arr[0].i[6] = 0              # replay on the main stream
    for _ in range(3):
        L.lib.dsl_run_ops(arr, 1, L.stream_ptr())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        L.lib.dsl_run_ops(arr, 1, L.stream_ptr())
    e1.record()
    torch.cuda.synchronize()
    op.i[6] = saved
    return e0.elapsed_time(e1) / reps * 1e-3


rows = []
lists = [('fwd', plan.fwd)] + [(f'bwd{i}', ol) for i, (ol, _) in enumerate(plan.bwd_segments)]
for lname, ol in lists:
    for op in ol.items:
        if op.kind == L.OP_CONV:
            d = C.cast(op.desc, C.POINTER(L.ConvDesc)).contents
            px = sum(d.n * d.gh[s] * d.gw[s] for s in range(d.nseg))
            spx = sum(d.n * d.sh[s] * d.sw[s] for s in range(d.nseg))
            dpx = sum(d.n * d.dh[s] * d.dw[s] for s in range(d.nseg))
            cin = 3 if (d.flags & L.CONV_SMALL_C) else d.cs
            fl = 2.0 * px * d.cd * d.kh * d.kw * cin
            by = spx * d.cs * 2 + dpx * d.cd * (4 if d.flags & L.CONV_OUT_F32 else 2) + d.cd_pad * d.kh * d.kw * d.cs * 2
            if d.addend:
                by += dpx * d.cd * 2
            if d.mask:
                by += dpx * d.cd * 2
            desc = f"{'dgrad' if d.mode else 'conv '} {d.kh}x{d.kw} s{d.stride} os{d.os} {d.cs:4d}->{d.cd:4d} px{px:6d} nseg{d.nseg}"
        elif op.kind == L.OP_WGRAD:
            d = C.cast(op.desc, C.POINTER(L.WgradDesc)).contents
            px = sum(d.n * d.gh[s] * d.gw[s] for s in range(d.nseg))
            spx = sum(d.n * d.sh[s] * d.sw[s] for s in range(d.nseg))
            fl = 2.0 * px * d.cd * d.kh * d.kw * d.cs
            by = px * d.cy * 2 + spx * d.cs * 2 + d.cd * d.kh * d.kw * d.cs * 4
            desc = f"wgrad {d.kh}x{d.kw} s{d.stride}     {d.cs:4d}->{d.cd:4d} px{px:6d} nseg{d.nseg}"
        elif op.kind == L.OP_WGRAD_GROUP:
            cnt = op.i[0]
            arr = C.cast(op.desc, C.POINTER(L.WgradDesc))
            d = arr[0]
            px = sum(d.n * d.gh[s] * d.gw[s] for s in range(d.nseg))
            spx = sum(d.n * d.sh[s] * d.sw[s] for s in range(d.nseg))
            fl = 2.0 * px * d.cd * d.kh * d.kw * d.cs * cnt
            by = (px * d.cy * 2 + spx * d.cs * 2 + d.cd * d.kh * d.kw * d.cs * 4) * cnt
            desc = f"wgrad {d.kh}x{d.kw} s{d.stride} x{cnt:<3d} {d.cs:4d}->{d.cd:4d} px{px:6d} nseg{d.nseg}"
        elif op.kind == L.OP_WGRAD_MULTI:
            f_, b_, nb, nr, ns = C.c_double(), C.c_double(), C.c_int(), C.c_int(), C.c_int()
            L.lib.dsl_wgrad_multi_info(C.c_void_p(op.p[0]), C.byref(f_), C.byref(b_), C.byref(nb), C.byref(nr), C.byref(ns))
            fl, by = f_.value, b_.value
            desc = f"wgrad multi: {ns.value} sub-launches, {nb.value} + {nr.value} workgroups"
        elif op.kind in (L.OP_GN_FWD, L.OP_GN_BWD):
            d = C.cast(op.desc, C.POINTER(L.GnDesc)).contents
            px = sum(d.n * d.h[s] * d.w[s] for s in range(d.nseg))
            fl = 0.0
            by = px * d.c * 2 * (3 if op.kind == L.OP_GN_FWD else 5)     # fwd: 2 reads + 1 write; bwd: 2x2 reads + 1 write
            desc = f"{'gn_fwd' if op.kind == L.OP_GN_FWD else 'gn_bwd'} c{d.c} px{px}"
        else:
            continue
        t = timeit(op)
        floor = max(fl / MFMA, by / HBM)
        rows.append((lname, desc, t, fl, by, floor))
tot = sum(r[2] for r in rows)
totf = sum(r[5] for r in rows)
print(f'# {len(rows)} ops, sum of standalone times {tot*1e3:.3f} ms, sum of floors {totf*1e3:.3f} ms')
print(f'# floor = max(flops / {MFMA/1e12:.0f} TF, bytes / {HBM/1e12:.1f} TB/s)')
print(f"{'list':5s} {'op':52s} {'us':>7s} {'TF':>6s} {'GB/s':>6s} {'floor':>6s} {'lost':>6s}")
for r in sorted(rows, key=lambda r: -(r[2] - r[5])):
    lname, desc, t, fl, by, floor = r
    print(f'{lname:5s} {desc:52s} {t*1e6:7.1f} {fl/t/1e12:6.0f} {by/t/1e9:6.0f} {floor*1e6:6.1f} {(t-floor)*1e6:6.1f}')

cats = {}
for lname, desc, t, fl, by, floor in rows:
    k = desc.split()[0] + (' ' + desc.split()[1] if desc.split()[0] in ('conv', 'dgrad', 'wgrad') else '')
    a = cats.setdefault(k, [0, 0.0, 0.0])
    a[0] += 1
    a[1] += t
    a[2] += floor
print('# by category: count, standalone ms, floor ms')
for k, (c, t, f) in sorted(cats.items(), key=lambda kv: -kv[1][1]):
    print(f'#   {k:12s} {c:4d} {t*1e3:7.3f} {f*1e3:7.3f}')
