#!/bin/bash
# Development: batched-launch kernel times for S sectors (one group) with a library variant.  batch_quick.sh TAG S VARIANT
tag=$1; S=$2; var=$3
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
d=$out/kt_$S$var; rm -rf $d
MHT_LIB_VARIANT=$var MHT_BENCH_GROUPS=1 rocprofv3 --kernel-trace -d $d -o kt -- python $root/bench.py --cpu-scans 0 --pmc off --sectors $S --steps 60 --warmup 10 > $out/bench_$S$var.log 2>&1
python $root/tools/rocpd_stats.py $(find $d -name '*_results.db' | head -1) > $out/stats_$S$var.txt 2>&1
echo "== S=$S variant=$var"; grep -E "batch" $out/stats_$S$var.txt | cut -c1-150
tail -1 $out/bench_$S$var.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('   multi_sector', d['multi_sector']['scans_per_sec'], d['multi_sector']['ok'], 'single', d['value'])"
rm -rf $d
