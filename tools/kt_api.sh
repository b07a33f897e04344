#!/bin/bash
# kernel trace of the STREAMED drop-in API path (device initiator): per-scan timeline of fgrow_adm / blp_uf / initiator / staging launches.   bash tools/kt_api.sh tag
set -u
tag=${1:-kt_api}
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $out/kt -o kt -- python $root/tools/api_profile.py 416 > $out/api_kt.log 2>&1
cd $root
db=$(find $out/kt -name '*_results.db' | head -1)
python tools/rocpd_stats.py $db > $out/kernel_stats.txt 2>&1
python - $db > $out/timeline.txt <<'P'
import sqlite3, sys, numpy as np
c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
disp = next(t for t in tabs if t.startswith("rocpd_kernel_dispatch")); sym = next(t for t in tabs if t.startswith("rocpd_info_kernel_symbol"))
names = {r[0]: r[1] for r in c.execute('select id, kernel_name from "%s"' % sym)}
rows = [(s, e, names[k]) for k, s, e in c.execute('select kernel_id, start, end from "%s" order by start' % disp)]
pick = lambda key: [(s, e) for s, e, n in rows if key in n]
blp, fg, ini, stg = pick('blp_uf'), pick('fgrow_adm'), pick('initiator_side'), pick('stage_scan')
out = []
for (bs, be) in blp[-300:-2]:
    nf = next((f for f in fg if f[0] > bs), None); nb = next((b for b in blp if b[0] > bs), None)
    ni = next((i for i in ini if i[1] > bs), None); ns = next((x for x in stg if x[1] > bs), None)
    if not (nf and nb and ni and ns): continue
    out.append([(be - bs) / 1e3, (nf[0] - bs) / 1e3, (nf[1] - bs) / 1e3, (nb[0] - bs) / 1e3, (ni[0] - bs) / 1e3, (ni[1] - bs) / 1e3, (ns[0] - bs) / 1e3, (ns[1] - bs) / 1e3])
a = np.array(out)
lab = ['ILP end', 'next grow start', 'next grow end', 'next ILP start (= period)', 'initiator start', 'initiator end', 'next staging start', 'next staging end']
print('streamed API, per scan, us from the ILP launch start (mean / p50):')
for i, l in enumerate(lab): print('  %-28s %7.1f %7.1f' % (l, a[:, i].mean(), np.median(a[:, i])))
P
rm -rf $out/kt
head -12 $out/kernel_stats.txt | cut -c1-150; cat $out/timeline.txt; grep "per scan" $out/api_kt.log
