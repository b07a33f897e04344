cd $GRAFT_REPO_ROOT
for cfg in "256 96" "128 64" "256 64"; do set -- $cfg
MHT_EXTRA_HIPCC_FLAGS="-DMHT_FG_THREADS=$1 -DMHT_FG_CAP=$2" python -m pymht_amd.build --force >/dev/null 2>&1
echo "threads=$1 cap=$2"; for g in 1 2; do MHT_BENCH_GROUPS=$g python bench.py --cpu-scans 0 --sectors 4 --steps 300 --warmup 40 2>/dev/null | tail -1 | python -c "import json,sys;d=json.loads(sys.stdin.read());print(round(d['value']),round(d['multi_sector']['scans_per_sec']), d['multi_sector']['ok'], d['config']['replay_matches_prepass'], round(d['stage_ms']['gate']*1000,1))"; done
done
