"""Constant-turn fuzz campaign: random scenarios with random turn rates through the constant-turn forest (Tracker(models.ct, ...)) against
the live oracle (tests/fuzz_util.py::run_case_ct), scan by scan.   python tools/fuzz_ct.py [n_cases] [first_seed] [similar]
(third argument "similar": with similar-state pruning switched on and off at random)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from fuzz_util import run_case_ct

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 300000
similar = len(sys.argv) > 3 and sys.argv[3] == 'similar'
bad, scans_ilp = 0, 0
for case in range(n_cases):
    try:
        ok, desc, msg = run_case_ct(seed0 + case, similar=similar)
    except Exception as e:
        ok, desc, msg = False, 'seed %d' % (seed0 + case), 'ERROR ' + repr(e)[:300]
    if not ok or case % 50 == 0:
        print(desc, 'ok' if ok else 'BAD', msg, flush=True)
    bad += 0 if ok else 1
    if 'ilp=' in msg:
        scans_ilp += int(msg.split('ilp=')[1].split()[0])
print('%d constant-turn cases from seed %d, %d bad (ILPs solved in the last scans of the cases: %d)' % (n_cases, seed0, bad, scans_ilp))
