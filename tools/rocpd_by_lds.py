#!/usr/bin/env python
"""Like rocpd_stats.py, but dispatches of one kernel are split by their dynamic LDS size and grid (launch variants)."""
import collections, sqlite3, sys
c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
disp = next(t for t in tabs if t.startswith("rocpd_kernel_dispatch"))
sym = next(t for t in tabs if t.startswith("rocpd_info_kernel_symbol"))
names = {r[0]: r[1] for r in c.execute('select id, kernel_name from "%s"' % sym)}
agg = collections.defaultdict(list)
for kid, s, e, gx, gy, lds in c.execute('select kernel_id, start, end, grid_size_x, grid_size_y, group_segment_size from "%s" order by start' % disp):
    agg[(names[kid][:44], gx, gy, lds)].append((e - s) / 1e3)
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    v = sorted(v)
    print("%-46s grid %7d x %2d lds %6d  calls %5d avg %8.2f med %8.2f p90 %8.2f max %8.2f" % (k[0], k[1], k[2], k[3], len(v), sum(v) / len(v), v[len(v) // 2], v[int(len(v) * 0.9)], v[-1]))
