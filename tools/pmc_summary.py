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
summary = {}
for k, a in agg.items():
    durs = list(a['_dur'].values())
    print(f'{k}: launches {len(durs)} avg {sum(durs)/len(durs)/1e3:.1f} us')
    for cn, v in sorted(a.items()):
        if cn != '_dur':
            print(f'    {cn:28s} {sum(v)/len(v):14.0f}')
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in a and 'SQ_BUSY_CYCLES' in a:
        # The SQ counters of this rocprofv3 are sums over ONE shader engine (8 CUs = 32 SIMDs; checked: SQ_INSTS_MFMA x 32 = the
        # kernel's MFMA count, SQ_BUSY_CYCLES = kernel duration x clock): MFMA-pipe busy cycles / (busy cycles x 32 SIMDs)
        mf, bz = sum(a['SQ_VALU_MFMA_BUSY_CYCLES']), sum(a['SQ_BUSY_CYCLES'])
        frac = mf / (bz * 32.0) if bz else 0.0
        print(f'    {"mfma_busy_frac":28s} {frac:14.4f}')
        summary[k] = dict(launches=len(durs), avg_us=round(sum(durs) / len(durs) / 1e3, 2), mfma_busy_frac=round(frac, 4),
                          wait_any_frac=round(sum(a.get('SQ_WAIT_ANY', [0])) / max(sum(a.get('SQ_WAVE_CYCLES', [1])), 1), 4))
if len(sys.argv) > 2:
    import json
    json.dump(dict(source='rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY ... of `python bench.py --steps 3 '
                          '--warmup 2 --no-prof --no-dsl`; mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CYCLES x 32 SIMDs per shader engine)',
                   kernels=summary), open(sys.argv[2], 'w'), indent=1)
