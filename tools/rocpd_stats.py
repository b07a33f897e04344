#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd SQLite database (`rocprofv3 --kernel-trace ... -d DIR -o NAME` -> NAME_results.db)
into a per-kernel table (calls, total, average, min, max duration in microseconds), like `--stats` would print.
Usage: python tools/rocpd_stats.py gpurun_out/prof/NAME_results.db [--after-first N]"""
import collections
import sqlite3
import sys


def main():
    path = sys.argv[1]
    c = sqlite3.connect(path)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    disp = next(t for t in tabs if t.startswith("rocpd_kernel_dispatch"))
    sym = next(t for t in tabs if t.startswith("rocpd_info_kernel_symbol"))
    names = {r[0]: r[1] for r in c.execute('select id, kernel_name from "%s"' % sym)}
    regs = {r[0]: (r[1], r[2]) for r in c.execute('select id, arch_vgpr_count, sgpr_count from "%s"' % sym)}
    agg = collections.defaultdict(list)
    meta = {}
    for kid, s, e, g, w, lds in c.execute('select kernel_id, start, end, grid_size_x, workgroup_size_x, group_segment_size '
                                          'from "%s" order by start' % disp):
        agg[names[kid]].append((e - s) / 1e3)
        meta[names[kid]] = (g, w, lds) + regs[kid]
    tot = sum(sum(v) for v in agg.values())
    print("%-58s %6s %11s %9s %9s %9s %9s %9s %6s  %s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "p50_us", "p95_us", "max_us", "%", "grid/wg/lds/vgpr/sgpr (last)"))
    pct = lambda v, q: sorted(v)[min(len(v) - 1, int(q * len(v)))]
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        print("%-58s %6d %11.1f %9.2f %9.2f %9.2f %9.2f %9.2f %6.1f  %s" % (k[:58], len(v), sum(v), sum(v) / len(v), min(v), pct(v, 0.5), pct(v, 0.95), max(v),
                                                                 100 * sum(v) / tot, "/".join(str(x) for x in meta[k])))


if __name__ == "__main__":
    main()
