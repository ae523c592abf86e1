"""Per-queue busy time of the last steps in a rocprofv3 kernel trace (rocpd sqlite): how much of the step each
HIP stream (hardware queue) keeps the GPU busy, how much they overlap, and the idle gaps on the critical queue.
Usage: python tools/stream_timeline.py results.db [n_last_steps] [marker_kernel_substring]"""
import re
import sqlite3
import sys
db = sqlite3.connect(sys.argv[1])
nlast = int(sys.argv[2]) if len(sys.argv) > 2 else 5
marker = sys.argv[3] if len(sys.argv) > 3 else 'loss_kernel'
rows = db.execute('select name, queue_id, start, end from kernels order by start').fetchall()
marks = [r[3] for r in rows if marker in r[0]]
t0, t1 = marks[-nlast - 1], marks[-1]
sel = [r for r in rows if r[2] >= t0 and r[3] <= t1]
span = (t1 - t0) / 1e6
print(f'{nlast} steps, {span / nlast:.3f} ms/step in the trace, {len(sel) / nlast:.0f} kernels/step')


def union(iv):
    iv = sorted(iv)
    tot, cs, ce = 0, *iv[0]
    for s, e in iv[1:]:
        if s <= ce:
            ce = max(ce, e)
        else:
            tot += ce - cs
            cs, ce = s, e
    return tot + ce - cs


queues = sorted({r[1] for r in sel})
for q in queues:
    iv = [(r[2], r[3]) for r in sel if r[1] == q]
    print(f'queue {q}: {len(iv) / nlast:6.0f} kernels/step, sum {sum(e - s for s, e in iv) / 1e6 / nlast:7.3f} ms/step, '
          f'busy {union(iv) / 1e6 / nlast:7.3f} ms/step')
allu = union([(r[2], r[3]) for r in sel])
print(f'all queues: busy (union) {allu / 1e6 / nlast:.3f} ms/step, idle {span / nlast - allu / 1e6 / nlast:.3f} ms/step, '
      f'sum {sum(r[3] - r[2] for r in sel) / 1e6 / nlast:.3f} ms/step')
agg = {}
for n, q, s, e in sel:
    m = re.search(r'(\w+_kernel(<[^>]*>)?)', n)
    k = (m.group(1) if m else n[:60], q)
    a = agg.setdefault(k, [0, 0])
    a[0] += 1
    a[1] += e - s
print('top kernels (per step):')
for (n, q), (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:22]:
    print(f'  q{q} {c / nlast:5.1f} x {t / c / 1e3:7.1f} us = {t / 1e6 / nlast:6.3f} ms  {n}')
