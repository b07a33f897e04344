"""Development / campaign tool: AIS-aided fuzz cases (tests/fuzz_util.py::run_case_ais) on the GPU box.  usage: fuzz_ais.py SEED0 COUNT"""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "oracle"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from fuzz_util import run_case_ais

seed0, n = int(sys.argv[1]), int(sys.argv[2])      # (note the order: first seed, then count)
bad = 0
for s in range(seed0, seed0 + n):
    ok, desc, msg = run_case_ais(s)
    if not ok:
        bad += 1
    print(("ok  " if ok else "BAD ") + desc + " | " + msg, flush=True)
print("cases %d bad %d" % (n, bad))
