"""Development aid: randomised parity runs of the STREAMED drop-in path (scans queued back to back, commit + admission inside the next grow
launch, initiator on the side stream, reports folded two scans late) against the live oracle (tests/fuzz_util.py: run_case_streamed).
python tools/fuzz_streamed.py [n_cases] [first_seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from fuzz_util import run_case_streamed

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
bad = 0
for case in range(n_cases):
    try:
        ok, desc, msg = run_case_streamed(seed0 + case)
    except Exception as e:
        ok, desc, msg = False, 'seed %d' % (seed0 + case), 'ERROR ' + repr(e)[:300]
    if not ok or case % 50 == 0:
        print(desc, 'ok' if ok else 'BAD', msg, flush=True)
    bad += 0 if ok else 1
print('%d streamed cases from seed %d, %d bad' % (n_cases, seed0, bad))
