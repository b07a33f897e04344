#!/usr/bin/env python
"""Batched (multi-sector) launches in a rocprofv3 kernel trace: duration per kernel, and how much of the wall time of the batched
phase has 0 / 1 / 2+ kernels running (do the groups' launch chains overlap?).  Usage: python tools/overlap_stats.py X_results.db"""
import sqlite3, sys
import numpy as np
c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
disp = next(t for t in tabs if t.startswith("rocpd_kernel_dispatch"))
sym = next(t for t in tabs if t.startswith("rocpd_info_kernel_symbol"))
names = {r[0]: r[1] for r in c.execute('select id, kernel_name from "%s"' % sym)}
rows = [(s, e, names[k]) for k, s, e in c.execute('select kernel_id, start, end from "%s" order by start' % disp) if 'batch' in names[k]]
if not rows:
    sys.exit("no batched launches in the trace")
rows = rows[len(rows) // 3:]      # steady state
by = {}
for s, e, n in rows:
    key = 'fgrow_batch' if 'fgrow' in n else 'cluster_batch' if 'cluster' in n else 'blp_batch'
    by.setdefault(key, []).append((e - s) / 1e3)
for k, v in by.items():
    print('%-14s n %5d  mean %.1f us  p50 %.1f  p95 %.1f' % (k, len(v), np.mean(v), np.median(v), np.percentile(v, 95)))
ev = sorted([(s, 1) for s, e, n in rows] + [(e, -1) for s, e, n in rows])
t0, t1 = ev[0][0], ev[-1][0]
lvl, last, acc = 0, t0, {}
for t, d in ev:
    acc[min(lvl, 3)] = acc.get(min(lvl, 3), 0) + (t - last)
    lvl += d
    last = t
tot = t1 - t0
print('wall %.1f ms: ' % (tot / 1e6) + ', '.join('%d kernel(s) running %.0f %%' % (k, 100.0 * v / tot) for k, v in sorted(acc.items())))
print('sum of kernel durations / wall = %.2f' % (sum(e - s for s, e, n in rows) / tot))
