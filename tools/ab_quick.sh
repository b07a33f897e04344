#!/bin/bash
# A/B of an environment switch on the headline replay: tools/ab_quick.sh NAME=VALUE [steps]  -> value (scans/s) of three alternating pairs
cd ${GRAFT_REPO_ROOT:-/root/repo}
sw=$1; steps=${2:-400}
for i in 1 2 3; do
  a=$(python bench.py --steps $steps --warmup 40 --sectors 0 --cpu-scans 0 --pmc off 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.0f %s' % (d['value'], d['config']['replay_matches_prepass']))")
  b=$(env $sw python bench.py --steps $steps --warmup 40 --sectors 0 --cpu-scans 0 --pmc off 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.0f %s' % (d['value'], d['config']['replay_matches_prepass']))")
  echo "default: $a | $sw: $b"
done
