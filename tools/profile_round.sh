#!/bin/bash
# One profiling round on the GPU box: kernel trace + the two HBM counter passes of the guide (separate --pmc runs), summarised
# into gpurun_out/$1/.  Usage (through gpurun): bash tools/profile_round.sh r02a
set -u
tag=${1:-prof}
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
B="python $root/bench.py --cpu-scans 0 --sectors 0 --pmc off --extras off ${CFG:+--config $CFG}"
rocprofv3 --kernel-trace -d $out/kt -o kt -- $B --steps 200 --warmup 40 > $out/bench_kt.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $out/pf -o pf -- $B --steps 60 --warmup 8 > $out/bench_pf.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $out/pw -o pw -- $B --steps 60 --warmup 8 > $out/bench_pw.log 2>&1
cd $root
python tools/rocpd_stats.py $(find $out/kt -name '*_results.db' | head -1) > $out/kernel_stats.txt 2>&1
tail -1 $out/bench_kt.log >> $out/kernel_stats.txt
{ python tools/rocpd_pmc.py $(find $out/pf -name '*_results.db' | head -1); echo; python tools/rocpd_pmc.py $(find $out/pw -name '*_results.db' | head -1); } > $out/pmc.txt 2>&1
rm -rf $out/kt $out/pf $out/pw
head -12 $out/kernel_stats.txt; head -8 $out/pmc.txt; grep -A6 "WRITE" $out/pmc.txt | head -8
