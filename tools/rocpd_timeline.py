#!/usr/bin/env python
"""Per-scan timeline from a rocprofv3 kernel trace (rocpd database): period between consecutive fgrow launches, busy time and the
gaps in front of every kernel of the scan.  Usage: python tools/rocpd_timeline.py X_results.db [first_scan]"""
import sqlite3, sys, collections
import numpy as np
c = sqlite3.connect(sys.argv[1])
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 60
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
disp = next(t for t in tabs if t.startswith("rocpd_kernel_dispatch"))
sym = next(t for t in tabs if t.startswith("rocpd_info_kernel_symbol"))
names = {r[0]: r[1] for r in c.execute('select id, kernel_name from "%s"' % sym)}
rows = [(s, e, names[k]) for k, s, e in c.execute('select kernel_id, start, end from "%s" order by start' % disp)]
short = lambda n: n.split('ENS_')[0].replace('_ZN3mht', '').lstrip('0123456789').replace('.kd', '')[:22]
starts = [i for i, r in enumerate(rows) if 'fgrow' in r[2]]
per = collections.defaultdict(list)
periods, busy = [], []
for a, b in zip(starts[skip:-1], starts[skip + 1:]):
    seq = rows[a:b]
    periods.append((rows[b][0] - rows[a][0]) / 1e3)
    busy.append(sum(e - s for s, e, _ in seq) / 1e3)
    prev_end = rows[a - 1][1] if a else rows[a][0]
    for s, e, n in seq:
        per[short(n) + ' gap'].append((s - prev_end) / 1e3)
        per[short(n) + ' dur'].append((e - s) / 1e3)
        prev_end = e
print('%d scans: period mean %.1f us (p50 %.1f), kernels busy %.1f us' % (len(periods), np.mean(periods), np.median(periods), np.mean(busy)))
for k, v in per.items():
    print('  %-28s n/scan %.2f  mean %.2f  p50 %.2f' % (k, len(v) / len(periods), np.mean(v), np.median(v)))
