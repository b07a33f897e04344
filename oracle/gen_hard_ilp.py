"""TEST INFRASTRUCTURE (development container only): tests/golden/g7_ilp_hard.npz.

The 0-1 ILPs of the headline stream (BASELINE config 3, seed 5446) that a dual coordinate ascent does NOT certify within
8 rounds, plus a sample of those that need several rounds.  Source of the instances: oracle/mht_oracle.py (pinned bit
for bit against the reference by oracle/gen_golden.py) run with the host M-of-N initiator over 345 scans (~8 minutes);
optimum and uniqueness from an exact solver (gen_golden.gen_g4: HiGHS + no-good cut).  The fixture holds numbers only.

Run:  python oracle/gen_hard_ilp.py [n_scans]      (writes /tmp/ilp_all.pkl as a cache of the recorded instances)
"""
import os
import pickle
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import mht_oracle as orc  # noqa: E402
from m_of_n_oracle import Initiator  # noqa: E402
from pymht_amd.models import pv  # noqa: E402
from pymht_amd.utils.classDefinitions import MeasurementList  # noqa: E402
from pymht_amd.utils.scenario import make_config  # noqa: E402

CACHE = "/tmp/ilp_all.pkl"


def record(n_scans):
    sc = make_config("cfg3", seed=5446, n_scans=n_scans)

    class Adapter:
        def __init__(self):
            self.i = Initiator(2, 3, 20, pv.C_RADAR, pv.R_RADAR(), 4 * 2.5 ** 2)

        def processMeasurements(self, time_, z):
            return [(t.x_0, t.P_0, t.measurementNumber, t.measurement)
                    for t in self.i.processMeasurements(MeasurementList(time_, z))]

    o = orc.OracleTracker(sc["period"], sc["lambda_phi"], 1e-4, P_d=sc["P_d"], N=sc["N"], eta2=5.99, initiator=Adapter())
    for x in sc["x0"]:
        o.initiate_target(sc["t0"], x.copy(), orc.model_P0(), status="preinitialized")
    out = {}
    for k in range(n_scans):
        o.ilp_recorder = []
        o.add_scan(float(sc["times"][k]), sc["scans"][k])
        out[k] = o.ilp_recorder
        if k % 20 == 0:
            print("scan", k, "ILPs", len(out[k]), flush=True)
    return out


def ascent_rounds(cols, sizes, cost, max_rounds=8):
    """Rounds a dual coordinate ascent (auction steps on conflicted rows, see pymht_amd/csrc/mht_blp.hip) needs to reach
    the optimality certificate; max_rounds + 1 if it does not.  Only used to pick the instances."""
    nT, nH = len(sizes), len(cols)
    cost = np.asarray(cost, float)
    starts = np.concatenate([[0], np.cumsum(sizes)]).astype(int)
    nM = 1 + max((max(c) for c in cols if len(c)), default=-1)
    has = np.zeros((nH, nM + 1), bool)
    for h, c in enumerate(cols):
        has[h, list(c)] = True
    tgt = np.repeat(np.arange(nT), sizes)
    u = np.zeros(nM + 1)
    for it in range(max_rounds + 1):
        rc = cost + (has * u[None, :]).sum(axis=1)
        sel = np.array([starts[t] + np.argmin(rc[starts[t]:starts[t + 1]]) for t in range(nT)])
        usage = has[sel].sum(axis=0)
        confl = np.where(usage[:nM] >= 2)[0]
        slack = np.where((u[:nM] > 0) & (usage[:nM] == 0))[0]
        if len(confl) == 0 and len(slack) == 0:
            return it
        newu, busy, act = u.copy(), np.zeros(nT, bool), -np.ones(nT, int)
        for t in range(nT):
            rows = np.where(has[sel[t], :nM] & (usage[:nM] >= 2))[0]
            if len(rows):
                act[t] = rows[0]
        for m in confl:
            users = [t for t in range(nT) if has[sel[t], m]]
            if any(act[t] != m for t in users):
                continue
            regs = []
            for t in users:
                hs = np.arange(starts[t], starts[t + 1])
                alt = hs[~has[hs, m]]
                regs.append((rc[alt].min() if len(alt) else np.inf) - rc[sel[t]])
                busy[t] = True
            regs = sorted(regs, reverse=True)
            if np.isfinite(regs[1]):
                newu[m] = u[m] + regs[1] + 0.5 * min(regs[0] - regs[1], 1.0)
        for m in slack:
            hs = np.where(has[:, m])[0]
            gap = rc[hs] - rc[sel[tgt[hs]]]
            if busy[tgt[hs]].any():
                continue
            newu[m] = max(0.0, u[m] - (gap.min() * (1.0 + 2.0 ** -20) + 1e-9))
        u = newu
    return max_rounds + 1


if __name__ == "__main__":
    n_scans = int(sys.argv[1]) if len(sys.argv) > 1 else 345
    if os.path.exists(CACHE):
        rec = pickle.load(open(CACHE, "rb"))
    else:
        rec = record(n_scans)
        pickle.dump(rec, open(CACHE, "wb"))
    hard, multi = [], []
    for k in sorted(rec):
        for inst in rec[k]:
            r = ascent_rounds(inst["cols"], inst["sizes"], inst["cost"])
            if r > 8:
                hard.append(inst)
            elif r >= 4:
                multi.append(inst)
    print("%d instances, %d not certified in 8 rounds, %d need 4..8 rounds" % (sum(len(v) for v in rec.values()), len(hard), len(multi)))
    import gen_golden
    gen_golden.gen_g4(hard + multi[:40], name="g7_ilp_hard")
