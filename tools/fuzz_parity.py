"""Development aid: randomised parity runs -- the drop-in Tracker (device forest, device initiator) against the oracle on random
small scenarios (tests/fuzz_util.py), scan by scan.   python tools/fuzz_parity.py [n_cases] [first_seed] [similar]
(third argument "similar": with similar-state pruning switched on and off at random)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from fuzz_util import run_case

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
similar = len(sys.argv) > 3 and sys.argv[3] == 'similar'
bad = 0
for case in range(n_cases):
    try:
        ok, desc, msg = run_case(seed0 + case, similar=similar)
    except Exception as e:
        ok, desc, msg = False, 'seed %d' % (seed0 + case), 'ERROR ' + repr(e)[:300]
    print(desc, 'ok' if ok else 'BAD', msg, flush=True)
    bad += 0 if ok else 1
print('%d cases, %d bad' % (n_cases, bad))
