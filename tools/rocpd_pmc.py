#!/usr/bin/env python
"""Per-kernel average of one PMC counter from a rocprofv3 rocpd database (`rocprofv3 --pmc NAME -d DIR -o X`).
Usage: python tools/rocpd_pmc.py X_results.db"""
import collections
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
t = lambda p: next(x for x in tabs if x.startswith(p))
pmc, disp, sym, info = t("rocpd_pmc_event"), t("rocpd_kernel_dispatch"), t("rocpd_info_kernel_symbol"), t("rocpd_info_pmc")
names = {r[0]: r[1] for r in c.execute('select id, kernel_name from "%s"' % sym)}
cname = {r[0]: r[1] for r in c.execute('select id, name from "%s"' % info)} if "name" in [q[1] for q in c.execute('pragma table_info("%s")' % info)] else {}
agg = collections.defaultdict(list)
for kid, ev, val, pid in c.execute('select d.kernel_id, d.event_id, p.value, p.pmc_id from "%s" d join "%s" p on p.event_id = d.event_id order by d.start' % (disp, pmc)):
    agg[(names[kid], cname.get(pid, str(pid)))].append(val)
print("%-56s %-14s %6s %12s %12s %12s %12s %14s" % ("kernel", "counter", "calls", "avg", "min", "p50", "max", "avg of last 25%"))
for (k, cn), v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    tail = v[-max(1, len(v) // 4):]
    print("%-56s %-14s %6d %12.1f %12.1f %12.1f %12.1f %14.1f" % (k[:56], cn, len(v), sum(v) / len(v), min(v), sorted(v)[len(v) // 2], max(v), sum(tail) / len(tail)))
