#!/bin/bash
# kernel trace of the timed replay + per-scan timeline (start / end of the grow and ILP launches relative to the ILP launch's start)
set -u
tag=${1:-kt}; steps=${2:-200}
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $out/kt -o kt -- python $root/bench.py --cpu-scans 0 --sectors 0 --pmc off --steps $steps --warmup 40 > $out/bench_kt.log 2>&1
cd $root
db=$(find $out/kt -name '*_results.db' | head -1)
python tools/rocpd_stats.py $db > $out/kernel_stats.txt 2>&1
tail -1 $out/bench_kt.log | cut -c1-260 >> $out/kernel_stats.txt
python - $db > $out/timeline.txt <<'P'
import sqlite3, sys, numpy as np
c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
disp = next(t for t in tabs if t.startswith("rocpd_kernel_dispatch")); sym = next(t for t in tabs if t.startswith("rocpd_info_kernel_symbol"))
names = {r[0]: r[1] for r in c.execute('select id, kernel_name from "%s"' % sym)}
rows = [(s, e, names[k]) for k, s, e in c.execute('select kernel_id, start, end from "%s" order by start' % disp)]
blp = [(s, e) for s, e, n in rows if 'blp_uf' in n]
fg = [(s, e) for s, e, n in rows if 'fgrow_kernel' in n]
# replay part: the last 150 blp_uf launches; for each the next fgrow launch (by start)
out = []
fi = 0
for (bs, be) in blp[-150:]:
    nxt = [f for f in fg if f[0] > bs]
    if not nxt: break
    fs, fe = nxt[0]
    nb = [b for b in blp if b[0] > bs]
    out.append(((be - bs) / 1e3, (fs - bs) / 1e3, (fe - bs) / 1e3, ((nb[0][0] - bs) / 1e3) if nb else np.nan))
a = np.array(out)
print('per scan, us from the ILP launch start: ILP end %.1f (p50 %.1f) | next grow start %.1f (p50 %.1f) | next grow end %.1f (p50 %.1f) | next ILP start %.1f (p50 %.1f)' % (
    a[:,0].mean(), np.median(a[:,0]), a[:,1].mean(), np.median(a[:,1]), a[:,2].mean(), np.median(a[:,2]), np.nanmean(a[:,3]), np.nanmedian(a[:,3])))
for r in a[:12]: print('   ', ' '.join('%7.1f' % v for v in r))
P
rm -rf $out/kt
head -6 $out/kernel_stats.txt | cut -c1-150; cat $out/timeline.txt
