"""Kernel sequence of the last training step in a rocprofv3 kernel trace (rocpd sqlite): start offset, queue, duration, name.
Usage: python tools/step_sequence.py results.db [marker_kernel_substring]"""
import re
import sqlite3
import sys
db = sqlite3.connect(sys.argv[1])
marker = sys.argv[2] if len(sys.argv) > 2 else 'loss_kernel'
rows = db.execute('select name, queue_id, start, end from kernels order by start').fetchall()
marks = [r[3] for r in rows if marker in r[0]]
t0, t1 = marks[-2], marks[-1]
qs = sorted({r[1] for r in rows if t0 <= r[2] <= t1})
prev_end = {q: t0 for q in qs}
for n, q, s, e in rows:
    if s < t0 or e > t1:
        continue
    m = re.search(r'(\w+_kernel(<[^>]*>)?)', n)
    name = m.group(1) if m else n[:50]
    gap = (s - prev_end[q]) / 1e3
    prev_end[q] = e
    print(f'{(s - t0) / 1e3:8.1f} us  q{qs.index(q) + 1}  {(e - s) / 1e3:7.1f} us  gap {gap:6.1f}  {name}')
