# A/B of the one-launch-per-scan kernel (blp_grow_kernel) against the launch pair (MHT_NO_MERGE=1)
mkdir -p gpurun_out/s2
B="python bench.py --cpu-scans 0 --pmc off --sectors 0"
ext() { python -c "
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    c=d['config']
    print(sys.argv[1], round(d['value']), '%.2f us' % (1e3*d['ms_per_step']), 'api', round(d['api_scans_per_sec']), 'same', c['replay_matches_prepass'], 'ovl', c['grow_launches_overlapping_ilp'], 'merged', c.get('scans_as_one_launch'), d.get('warning'))
except Exception as e: print(sys.argv[1], 'FAILED', e)
" $1; }
run() { n=$1; shift; env "$@" timeout 300 $B > gpurun_out/s2/$n.json 2>gpurun_out/s2/$n.err; ext gpurun_out/s2/$n.json; }
