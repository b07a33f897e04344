#!/bin/bash
# the round's bench lines and profiles of the current build, into gpurun_out/final/
set -u
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/final
mkdir -p $out
cd $root
python bench.py --steps 20 --warmup 5 > $out/bench_driver_args.json 2> $out/bench_driver_args.err
python bench.py > $out/bench_default.json 2> $out/bench_default.err
MHT_BENCH_FORCE_DIST=1 python bench.py --gpus 1 --steps 20 --warmup 5 --sectors 0 --cpu-scans 0 --pmc off > $out/bench_force_dist.json 2> $out/bench_force_dist.err
bash tools/kt_quick.sh final/kt 200 > /dev/null 2>&1
{ python tools/api_timeline.py 2>&1 | tail -2; python tools/api_profile.py 416 2>&1 | grep "per scan\|scans:"; MHT_HOST_PROF=1 python tools/api_profile.py 416 2>&1 | grep "host prof" | tail -1; } > $out/api_path.txt
python tools/ovl_timeline.py > $out/ovl_timeline.txt 2>&1
grep -h '^{' $out/bench_force_dist.json > $out/bench_force_dist.line.json; tail -c 400 $out/bench_driver_args.err
