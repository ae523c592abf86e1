"""HBM traffic per launch of the dominant kernels from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of
bench.py, corrected as /opt/skills/guides/MI355X_MICROARCH.md (HBM section) prescribes for gfx950: FETCH_SIZE counts
128-byte read requests at 64 B -> doubled; both counters are in KiB.  WRITE_SIZE is uncalibrated (used as is).
A bench.py log (its JSON line) given as 4th argument adds the ALGORITHMIC bytes per launch bench.py derives for its kernel
classes next to the measured ones (traffic well above the algorithmic bytes = wasted re-reads).
Usage: python tools/pmc_traffic.py fetch_results.db write_results.db out.json [bench.log]"""
import json
import re
import sqlite3
import sys


def per_kernel(path, counter):
    db = sqlite3.connect(path)
    agg = {}
    for name, cn, cv in db.execute('select name, counter_name, counter_value from pmc_events'):
        if cn != counter:
            continue
        m = re.search(r'(\w+_kernel(<[^>]*>)?)', name)
        k = m.group(1) if m else name[:40]
        a = agg.setdefault(k, [0, 0.0])
        a[0] += 1
        a[1] += cv
    return {k: (n, s / n) for k, (n, s) in agg.items()}


fetch = per_kernel(sys.argv[1], 'FETCH_SIZE')
write = per_kernel(sys.argv[2], 'WRITE_SIZE')
out = {}
for k in fetch:
    if k in write:
        n, f = fetch[k]
        _, w = write[k]
        out[k] = dict(launches=n, fetch_size_kib=round(f, 1), write_size_kib=round(w, 1),
                      hbm_bytes_per_launch=round((2 * f + w) * 1024))
algo = {}
if len(sys.argv) > 4:
    for line in open(sys.argv[4]):
        if line.startswith('{') and '"roofline"' in line:
            r = json.loads(line)['roofline']
            algo['conv_pipe_kernel<128, 128, 2, 4, 2, false>'] = r.get('algorithmic_bytes_per_launch')
            for key, kn in (('head_tile_256x192', 'conv_pipe_kernel<256, 192, 4, 2, 2, false>'),):
                if key in r:
                    algo[kn] = round(r[key]['algorithmic_mb_per_launch'] * 1e6)
            if 'wgrad_kernels' in r:
                algo['wgrad (all launches, average)'] = round(r['wgrad_kernels']['algorithmic_mb_per_launch'] * 1e6)
for k, v in out.items():
    if k in algo and algo[k]:
        v['algorithmic_bytes_per_launch'] = algo[k]
        v['measured_over_algorithmic'] = round(v['hbm_bytes_per_launch'] / algo[k], 2)
wg = [(v['launches'], v['hbm_bytes_per_launch']) for k, v in out.items() if k.startswith('wgrad')]
if wg and algo.get('wgrad (all launches, average)'):
    tot = sum(n * b for n, b in wg)
    nl = sum(v['launches'] for k, v in out.items() if k.startswith(('wgrad_glds', 'wgrad_pipe')))
    out['wgrad (all launches, average)'] = dict(launches=nl, hbm_bytes_per_launch=round(tot / max(nl, 1)),
                                                algorithmic_bytes_per_launch=algo['wgrad (all launches, average)'],
                                                measured_over_algorithmic=round(tot / max(nl, 1) / algo['wgrad (all launches, average)'], 2),
                                                note='main + reduce kernels of the weight gradients per main launch')
# whole-step HBM traffic: every kernel's bytes x launches over the profiled run, per training step (the run is --steps 3 --warmup 2: five
# steps; the few one-off initialisation kernels of the process are in the sum - an upper bound by < 1 %)
STEPS = 5
total = sum(v['launches'] * v['hbm_bytes_per_launch'] for k, v in out.items() if not k.startswith('wgrad (all'))
step_total = dict(hbm_bytes_per_step=round(total / STEPS), steps_profiled=STEPS,
                  hbm_floor_ms_at_8TBs=round(total / STEPS / 8e12 * 1e3, 3), hbm_floor_ms_at_6p3TBs=round(total / STEPS / 6.3e12 * 1e3, 3))
json.dump(dict(step=step_total, source='rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of `python bench.py --steps 3 '
                      '--warmup 2 --no-prof --no-dsl`; read bytes = 2 x FETCH_SIZE (gfx950 correction), KiB units',
               kernels=out), open(sys.argv[3], 'w'), indent=1)
print('whole step:', step_total)
for k, v in sorted(out.items(), key=lambda kv: -kv[1]['hbm_bytes_per_launch'] * kv[1]['launches'])[:8]:
    print(k, v)
