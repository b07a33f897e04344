#!/bin/bash
# batch_quick3.sh TAG S : kernel times of the batched launch set with the two-tier ILP launch
tag=$1; S=$2
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
d=$out/kt_tt_$S; rm -rf $d
MHT_BLP_TWO_TIER=1 MHT_BENCH_GROUPS=1 rocprofv3 --kernel-trace -d $d -o kt -- python $root/bench.py --cpu-scans 0 --pmc off --sectors $S --steps 60 --warmup 10 > $out/bench_tt_$S.log 2>&1
python - <<PY
import sqlite3, glob, collections
db = glob.glob("$d/**/*_results.db", recursive=True)[0]
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
disp = next(t for t in tabs if t.startswith("rocpd_kernel_dispatch")); sym = next(t for t in tabs if t.startswith("rocpd_info_kernel_symbol"))
names = {r[0]: r[1] for r in c.execute('select id, kernel_name from "%s"' % sym)}
agg = collections.defaultdict(list)
for kid, s, e, g, lds in c.execute('select kernel_id, start, end, grid_size_x, group_segment_size from "%s" order by start' % disp):
    if "batch" in names[kid]: agg[(names[kid][:40], lds)].append((e - s) / 1e3)
for k, v in agg.items():
    v2 = v[len(v)//3:]
    print("S=$S", k, "calls", len(v), "avg(steady) %.1f p50 %.1f max %.1f" % (sum(v2)/len(v2), sorted(v2)[len(v2)//2], max(v2)))
PY
rm -rf $d
