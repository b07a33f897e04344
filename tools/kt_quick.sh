#!/bin/bash
# kernel trace of the timed replay only (no API pass, no extras): gpurun_out/$1/kernel_stats.txt.   bash tools/kt_quick.sh tag [steps]
set -u
tag=${1:-kt}; steps=${2:-200}
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $out/kt -o kt -- python $root/bench.py --cpu-scans 0 --sectors 0 --pmc off --steps $steps --warmup 40 > $out/bench_kt.log 2>&1
cd $root
python tools/rocpd_stats.py $(find $out/kt -name '*_results.db' | head -1) > $out/kernel_stats.txt 2>&1
tail -1 $out/bench_kt.log | cut -c1-260 >> $out/kernel_stats.txt
rm -rf $out/kt
head -9 $out/kernel_stats.txt | cut -c1-150
