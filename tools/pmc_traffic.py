"""HBM traffic per launch of the dominant kernels from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of
bench.py, corrected as /opt/skills/guides/MI355X_MICROARCH.md (HBM section) prescribes for gfx950: FETCH_SIZE counts
128-byte read requests at 64 B -> doubled; both counters are in KiB.  WRITE_SIZE is uncalibrated (used as is).
Usage: python tools/pmc_traffic.py fetch_results.db write_results.db out.json"""
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
json.dump(dict(source='rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of `python bench.py --steps 3 '
                      '--warmup 2 --no-prof`; read bytes = 2 x FETCH_SIZE (gfx950 correction), KiB units',
               kernels=out), open(sys.argv[3], 'w'), indent=1)
for k, v in sorted(out.items(), key=lambda kv: -kv[1]['hbm_bytes_per_launch'] * kv[1]['launches'])[:8]:
    print(k, v)
