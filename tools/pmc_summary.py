"""Per-kernel averages of PMC counters from a rocprofv3 rocpd database."""
import re
import sqlite3
import sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute('select name, dispatch_id, duration, counter_name, counter_value from pmc_events').fetchall()
agg = {}
for name, did, dur, cn, cv in rows:
    m = re.search(r'(\w+_kernel(<[^>]*>)?)', name)
    short = m.group(1) if m else name[:40]
    if 'at::native' in name:
        continue
    a = agg.setdefault(short, {})
    a.setdefault(cn, []).append(cv)
    a.setdefault('_dur', {})[did] = dur
for k, a in agg.items():
    durs = list(a['_dur'].values())
    print(f'{k}: launches {len(durs)} avg {sum(durs)/len(durs)/1e3:.1f} us')
    for cn, v in sorted(a.items()):
        if cn != '_dur':
            print(f'    {cn:28s} {sum(v)/len(v):14.0f}')
