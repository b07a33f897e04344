#!/bin/bash
# batch_quick2.sh TAG S : aggregate multi-sector rate for group counts / ILP tiers (no profiler)
tag=$1; S=$2
root=${GRAFT_REPO_ROOT:-/root/repo}
for tt in 0 1; do for ng in 1 2 4; do
  MHT_BLP_TWO_TIER=$tt MHT_BENCH_GROUPS=$ng python $root/bench.py --cpu-scans 0 --pmc off --sectors $S --steps 100 --warmup 10 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('S=$S two_tier=$tt groups=$ng multi_sector %.0f ok=%s single %.0f ratio %.2f' % (d['multi_sector']['scans_per_sec'], d['multi_sector']['ok'], d['value'], d['multi_sector']['scans_per_sec']/d['value']))"
done; done
