"""Summarise a rocprofv3 rocpd sqlite database (kernel trace) into a per-kernel table.
usage: python tools/rocprof_summary.py gpurun_out/prof/x_results.db > profiles/x_kernel_stats.txt"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
rows = db.execute('select name, start, end from kernels').fetchall()
agg = {}
for name, s, e in rows:
    name = re.sub(r'\s+', ' ', name)
    a = agg.setdefault(name, [0, 0.0, 1e30, 0.0])
    d = (e - s) * 1e-3
    a[0] += 1
    a[1] += d
    a[2] = min(a[2], d)
    a[3] = max(a[3], d)
tot = sum(a[1] for a in agg.values())
print(f'# rocprofv3 --kernel-trace --stats summary of {sys.argv[1]}')
print(f'# total kernel time {tot/1e3:.3f} ms over {len(rows)} dispatches')
print(f'{"calls":>7s} {"total_ms":>10s} {"avg_us":>9s} {"min_us":>9s} {"max_us":>9s} {"pct":>6s}  kernel')
for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f'{a[0]:7d} {a[1]/1e3:10.3f} {a[1]/a[0]:9.2f} {a[2]:9.2f} {a[3]:9.2f} {100*a[1]/tot:6.2f}  {name[:150]}')
