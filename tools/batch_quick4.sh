#!/bin/bash
# batch_quick4.sh S : aggregate multi-sector rate for ILP LDS caps (MHT_BLP_CAPS = columns,rows,targets solved out of LDS)
S=$1
root=${GRAFT_REPO_ROOT:-/root/repo}
for caps in "" "1024,512,64" "768,384,32" "512,256,16"; do for ng in 1 2; do
  MHT_BLP_CAPS=$caps MHT_BENCH_GROUPS=$ng python $root/bench.py --cpu-scans 0 --pmc off --sectors $S --steps 100 --warmup 10 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); m=d['multi_sector']; print('S=$S caps=$caps groups=$ng multi %.0f ok=%s single %.0f ratio %.2f ilp_ms %.4f' % (m['scans_per_sec'], m['ok'], d['value'], m['scans_per_sec']/d['value'], d['stage_ms']['ilp']))"
done; done
