#!/bin/bash
# A/B of the two-queue mode (MHT_TWO_QUEUES=1: the overlapping grow launch on a hardware queue of its own behind a residency gate) on one box
cd ${GRAFT_REPO_ROOT:-/root/repo}
run() { python bench.py --steps $1 --warmup 5 --sectors 0 --cpu-scans 0 --pmc off 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  value %.0f  ms/scan %.5f  replay ok %s  api %.0f' % (d['value'], d['ms_per_step'], d['config']['replay_matches_prepass'], d['api_scans_per_sec']))"; }
for rep in 1 2 3; do
  echo "== single queue, steps 20"; run 20
  echo "== MHT_TWO_QUEUES=1, steps 20"; MHT_TWO_QUEUES=1 run 20
done
echo "== single queue, steps 400"; run 400
echo "== MHT_TWO_QUEUES=1, steps 400"; MHT_TWO_QUEUES=1 run 400
