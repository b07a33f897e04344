#!/bin/bash
# Development: aggregate multi-sector rate with the light ILP pass on / off, group counts 1 / 2 / 4.  usage: batch_light.sh "4 16"
root=${GRAFT_REPO_ROOT:-/root/repo}
for S in $1; do for lt in 0 1; do for ng in 1 2 4; do
  MHT_BLP_LIGHT=$lt MHT_BENCH_GROUPS=$ng python $root/bench.py --cpu-scans 0 --pmc off --sectors $S --steps 100 --warmup 10 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); m=d['multi_sector']; print('S=$S light=$lt groups=$ng multi_sector %.0f ok=%s single %.0f ratio %.2f grow_frac %.4f' % (m['scans_per_sec'], m['ok'], d['value'], m['scans_per_sec']/d['value'], m['roofline']['frac']))"
done; done; done
