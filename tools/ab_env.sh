#!/bin/bash
# A/B of an environment switch on the tree's build: tools/ab_env.sh VAR=VALUE  (driver's arguments x 3, 400 scans x 1)
mkdir -p gpurun_out; rm -f gpurun_out/ab_env.log
pr='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d["config"].get("replay_matches_prepass"), d.get("api_scans_per_sec"))'
for rep in 1 2 3; do
  for v in "X_NONE=1" "$1"; do
    echo "== $v steps 20" >> gpurun_out/ab_env.log
    env $v python bench.py --steps 20 --warmup 5 --sectors 0 --cpu-scans 0 --pmc off 2>/dev/null | python -c "$pr" >> gpurun_out/ab_env.log 2>&1
  done
done
for v in "X_NONE=1" "$1"; do
  echo "== $v steps 400" >> gpurun_out/ab_env.log
  env $v python bench.py --steps 400 --warmup 20 --sectors 0 --cpu-scans 0 --pmc off 2>/dev/null | python -c "$pr" >> gpurun_out/ab_env.log 2>&1
done
cat gpurun_out/ab_env.log
