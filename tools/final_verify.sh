#!/bin/bash
# final verification of a build on the MI355X: smoke(), the GPU suite, the CPU suite, the bench line at the driver's arguments
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/final
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -v amdgpu | tail -1
python -m pytest tests -m gpu -q 2>&1 | tail -2
python -m pytest tests -m "not gpu" -q 2>&1 | tail -1
python bench.py --steps 20 --warmup 5 > gpurun_out/final/bench_driver_args.json 2> gpurun_out/final/bench.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/final/bench_driver_args.json').read().strip().splitlines()[-1])
print('value %.0f scans/s (%.4f ms/scan) | api %.0f | sectors %s | roofline frac %.4f traffic %.0f | cpu_baseline %.2f scans/s | replay_matches_prepass %s' % (
    d['value'], d['ms_per_step'], d['api_scans_per_sec'], [(m['sectors_per_gpu'], round(m['scans_per_sec']), m['ok']) for m in d['multi_sector_all']],
    d['roofline']['frac'], d['roofline']['traffic'] or 0, d['cpu_baseline']['value'], d['config']['replay_matches_prepass']))
PY
