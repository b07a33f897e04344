"""Development: rare divergence hunt, part 3 -- single forests replayed without any fetch in between; on a mismatch with the prepass
final the per-scan commit log (mht_forest_debug_read "commit_log") is compared with the log of a replay that matched.
usage: group_stress3.py S N_SCANS REPEATS"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from pymht_amd import parallel
from pymht_amd.utils.scenario import make_config

S, N, REP = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
NAMES = "scan nT nAlive L_in nCh Lnext nC n_ilp branched limit itmax e_over singles teams n_dead M".split()
local = 0
scs, brs, fins = [], [], []
for q in range(S):
    sq = make_config("cfg3", seed=parallel.sector_seed(5446, 0) + 17 * q, n_scans=N, centre=(0.0, 20000.0 * q), confine=True)
    bq, stq, fq, _, _ = bench.prepass(sq, local)
    scs.append(sq); brs.append(bq); fins.append(fq)


def read_log(r):
    a = np.zeros(64 * 16, np.int32)
    r._lib_mod.check(r.lib.mht_forest_debug_read(r.h, b"commit_log", a.ctypes.data_as(C.c_void_p), a.nbytes))
    return a.reshape(64, 16)


good = [None] * S
nbad = 0
for rep_i in range(REP):
    if os.environ.get("STRESS_STREAM", "0") == "1":
        st = torch.cuda.Stream(device=local, priority=0)
        with torch.cuda.stream(st):
            rps = [bench.Replay(scs[q], brs[q], local) for q in range(S)]
    else:
        rps = [bench.Replay(scs[q], brs[q], local) for q in range(S)]
    for k in range(N):
        for r in rps:
            r.step()
    torch.cuda.synchronize()
    for q, r in enumerate(rps):
        rep, recs = r.report()
        got = [(int(x["id"]), int(x["sel_meas"])) for x in recs if int(x["status"]) == 0]
        lg = read_log(r)
        if got == fins[q] and not rep.error:
            if good[q] is None:
                good[q] = lg
            elif not np.array_equal(good[q][1:N + 1], lg[1:N + 1]):
                print("repeat %d sector %d: final matches but the logs differ" % (rep_i, q))
            continue
        nbad += 1
        st = np.bincount(np.asarray(recs["status"], np.int64), minlength=4)
        print("repeat %d sector %d: MISMATCH live %d/%d error %d status histogram %s" % (rep_i, q, len(got), len(fins[q]), rep.error, st.tolist()), flush=True)
        if good[q] is not None:
            for s in range(1, N + 1):
                a, b = good[q][s & 63], lg[s & 63]
                if not np.array_equal(a, b):
                    print("  first differing scan %d:" % s)
                    print("    good: " + " ".join("%s=%d" % (n, v) for n, v in zip(NAMES, a)))
                    print("    bad : " + " ".join("%s=%d" % (n, v) for n, v in zip(NAMES, b)))
                    if s + 1 <= N:
                        print("    next good: " + " ".join("%s=%d" % (n, v) for n, v in zip(NAMES, good[q][(s + 1) & 63])))
                        print("    next bad : " + " ".join("%s=%d" % (n, v) for n, v in zip(NAMES, lg[(s + 1) & 63])))
                    break
        else:
            print("  (no matching replay of this sector yet) log:")
            for s in range(1, N + 1):
                print("    " + " ".join("%s=%d" % (n, v) for n, v in zip(NAMES, lg[s & 63])))
    for r in rps:
        r.close()
print("S=%d N=%d: %d sector replays of %d diverged" % (S, N, nbad, REP * S))
