#!/bin/bash
# Development: per-kernel times of the batched launch set (mht_group_step) for S sectors in ONE group, both grow variants.
# Usage (through gpurun): bash tools/batch_profile.sh TAG "4 16"
tag=${1:-bp}; Ss=${2:-"4 16"}
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for S in $Ss; do for w in ${WAVES:-0 1}; do
  d=$out/kt_S${S}_w${w}; rm -rf $d
  MHT_FG_WAVE=$w MHT_BENCH_GROUPS=${GROUPS_N:-1} rocprofv3 --kernel-trace -d $d -o kt -- python $root/bench.py --cpu-scans 0 --pmc off --extras off --sectors $S --steps 60 --warmup 10 > $out/bench_S${S}_w${w}.log 2>&1
  python $root/tools/rocpd_stats.py $(find $d -name '*_results.db' | head -1) > $out/stats_S${S}_w${w}.txt 2>&1
  echo "== S=$S wave=$w"; grep -E "batch|kernel  " $out/stats_S${S}_w${w}.txt | cut -c1-160
  tail -1 $out/bench_S${S}_w${w}.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('   multi_sector', d['multi_sector']['scans_per_sec'], 'single', d['value'])"
  rm -rf $d
done; done
