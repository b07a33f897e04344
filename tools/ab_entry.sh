#!/bin/bash
# A/B of two library builds at the driver's arguments and over 400 scans: libmht_amd.so.base (MHT_LIB_VARIANT=.base) against the tree's build
mkdir -p gpurun_out
for rep in 1 2 3; do
  for v in .base ""; do
    echo "== variant '$v' steps 20" >> gpurun_out/ab.log
    MHT_LIB_VARIANT=$v python bench.py --steps 20 --warmup 5 --sectors 0 --cpu-scans 0 --pmc off 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('stage_ms'), d.get('api_scans_per_sec'))" >> gpurun_out/ab.log 2>&1
  done
done
for v in .base ""; do
  echo "== variant '$v' steps 400" >> gpurun_out/ab.log
  MHT_LIB_VARIANT=$v python bench.py --steps 400 --warmup 20 --sectors 0 --cpu-scans 0 --pmc off 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('stage_ms'), d.get('api_scans_per_sec'))" >> gpurun_out/ab.log 2>&1
done
cat gpurun_out/ab.log
