#!/bin/bash
# A/B of an environment switch on the drop-in (streamed API) rate of the headline stream: tools/ab_api.sh NAME=VALUE [steps]
cd ${GRAFT_REPO_ROOT:-/root/repo}
sw=$1; steps=${2:-400}
one() { env "$@" python bench.py --steps $steps --warmup 40 --sectors 0 --cpu-scans 0 --pmc off --extras off 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.0f (value %.0f)' % (d['api_scans_per_sec'], d['value']))"; }
for i in 1 2 3; do echo "default: $(one A=1) | $sw: $(one $sw)"; done
