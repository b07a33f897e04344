#!/bin/bash
# the round's bench lines and profiles of the current build, into gpurun_out/final/   (through gpurun: bash tools/final_profiles.sh)
set -u
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/final
mkdir -p $out
cd $root
python bench.py --steps 20 --warmup 5 > $out/bench_driver_args.json 2> $out/bench_driver_args.err
python bench.py > $out/bench_default.json 2> $out/bench_default.err
MHT_BENCH_FORCE_DIST=1 python bench.py --gpus 1 --steps 20 --warmup 5 --sectors 0 --cpu-scans 0 --pmc off --extras off > $out/bench_force_dist.json 2> $out/bench_force_dist.err
# headline: kernel trace + HBM counters (separate passes)
bash tools/profile_round.sh final/cfg3 > $out/profile_cfg3.log 2>&1
# BASELINE config 5 (constant-turn forest): the same three passes
CFG=cfg5 bash tools/profile_round.sh final/cfg5 > $out/profile_cfg5.log 2>&1
# batched launch set (one group): kernel trace per S, and the HBM counters of the S = 16 launch set
WAVES="1" bash tools/batch_profile.sh final/batch "4 16" > $out/batch_kernels.txt 2>&1
( cd /tmp && export TMPDIR=/tmp
  for c in FETCH_SIZE WRITE_SIZE; do
    d=$out/bpmc_$c; rm -rf $d
    MHT_BENCH_GROUPS=1 rocprofv3 --pmc $c -d $d -o p -- python $root/bench.py --cpu-scans 0 --pmc off --extras off --sectors 16 --steps 20 --warmup 5 > $out/bench_bpmc_$c.log 2>&1
    echo "== $c, S = 16, one group"; python $root/tools/rocpd_pmc.py $(find $d -name '*_results.db' | head -1) | grep -E "batch|kernel " | cut -c1-200
    rm -rf $d
  done ) > $out/batch_pmc.txt 2>&1
{ python tools/api_timeline.py 2>&1 | tail -2; python tools/api_profile.py 416 2>&1 | grep "per scan\|scans:"; } > $out/api_path.txt 2>&1
python tools/ovl_timeline.py > $out/ovl_timeline.txt 2>&1
python tools/ilp_dist.py cfg3 70 2>&1 | tail -22 > $out/ilp_dist_cfg3.txt
python tools/ilp_dist.py cfg5 40 2>&1 | tail -22 > $out/ilp_dist_cfg5.txt
python tools/show_bench.py $out/bench_driver_args.json
