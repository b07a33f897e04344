#!/bin/bash
# multi-sector figures of bench.py under environment variants: tools/ab_sectors.sh "VAR=1 VAR2=x" ["..." ...]   ("-" = the default)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for sw in "$@"; do
  [ "$sw" = "-" ] && envs="" || envs="$sw"
  r=$(env $envs python bench.py --steps 20 --warmup 5 --sectors 4,16 --cpu-scans 0 --pmc off 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('value %.0f | ' % d['value'] + ' | '.join('S=%d %.0f scans/s (%.2fx) ok=%s grow %.0f us' % (m['sectors_per_gpu'], m['scans_per_sec'], m['x_single_sector'], m['ok'], m['roofline']['grow_us_per_scan_all_groups']) for m in d['multi_sector_all']))")
  echo "[$sw] $r"
done
