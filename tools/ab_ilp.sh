#!/bin/bash
# Development: A/B of library variants on one box: value and ILP stage time of the headline replay.  usage: ab_ilp.sh ".prev" "" ...
for rep in 1 2; do
for v in "$@"; do
  MHT_LIB_VARIANT=$v python bench.py --steps 400 --warmup 40 --sectors 0 --cpu-scans 0 --pmc off 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('variant %-8r value %6d  ilp %.2f us  gate %.2f  cluster %.2f  api %d' % ('$v', d['value'], 1e3*d['stage_ms']['ilp'], 1e3*d['stage_ms']['gate'], 1e3*d['stage_ms']['cluster'], d['api_scans_per_sec']))"
done; done
