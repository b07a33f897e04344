"""TEST INFRASTRUCTURE (development container only): tests/golden/g9_ilp_giant.npz (three instances).

(1) One 0-1 ILP of the giant-component regime: 29 targets / 17 935 columns (too large for the ILP kernel's LDS policy), recorded from
the oracle (pinned bit for bit against the reference by oracle/gen_golden.py) on a dense scenario found by tools/fuzz_parity.py
(68 objects inside a 180 m radius, N-scan 6, P_d 0.64, scan 5).  The dual is not tight on it (no certificate exists), so the solver
has to finish by branch and bound; an earlier version ran into the node limit here.  Exact optimum + uniqueness from HiGHS
(gen_golden.gen_g4).
(2) The instance with a wide LP gap (fuzz seed 40002, scan 3: 34 targets, 3 039 columns, LP optimum -48.178 vs ILP optimum -47.593,
five fractional targets): a depth-first search with a static Lagrangian bound needs ~10^6 nodes on it.
(3) Fuzz seed 90266, scan 4: 43 targets / 9 531 columns (HBM policy), 67 objects inside a 231 m radius without clutter: the branch
and bound needed 260 k nodes while it re-optimised the prices on its first 12 levels only.
The fixture holds numbers only.

Run:  python oracle/gen_giant_ilp.py          (~1 minute)
"""
import os
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
from pymht_amd.utils.scenario import make_scenario  # noqa: E402

def fuzz_scenario(seed):
    rng = np.random.default_rng(seed)           # the parameter draw of tools/fuzz_parity.py for this seed
    T = int(rng.integers(1, 70)); radius = float(rng.uniform(80, 900)); lam = float(rng.choice([0.0, 1e-6, 1e-5, 5e-5, 1.5e-4]))
    N = int(rng.integers(1, 8)); P_d = float(rng.uniform(0.5, 0.99)); eta2 = float(rng.choice([4.61, 5.99, 9.21]))
    period = float(rng.choice([1.0, 2.5, 4.0])); ns = int(rng.integers(4, 12))
    return make_scenario(T=T, radius=radius, lambda_phi=lam, n_scans=ns, P_d=P_d, period=period, seed=seed), N, eta2


class Adapter:
    def __init__(self):
        self.i = Initiator(2, 3, 20, pv.C_RADAR, pv.R_RADAR(), 4 * 2.5 ** 2)

    def processMeasurements(self, time_, z):
        return [(t.x_0, t.P_0, t.measurementNumber, t.measurement)
                for t in self.i.processMeasurements(MeasurementList(time_, z))]


def largest_ilp(seed, scan):
    sc, N, eta2 = fuzz_scenario(seed)
    o = orc.OracleTracker(sc["period"], sc["lambda_phi"], 1e-4, P_d=sc["P_d"], N=N, eta2=eta2, initiator=Adapter())
    for x in sc["x0"]:
        o.initiate_target(sc["t0"], x.copy(), orc.model_P0(), status="preinitialized")
    for k in range(scan + 1):
        o.ilp_recorder = []
        o.add_scan(float(sc["times"][k]), sc["scans"][k])
    inst = max(o.ilp_recorder, key=lambda i: len(i["cols"]))
    print("seed %d scan %d: %d targets, %d columns" % (seed, scan, len(inst["sizes"]), len(inst["cols"])))
    return inst


if __name__ == "__main__":
    import gen_golden
    gen_golden.gen_g4([largest_ilp(20025, 4), largest_ilp(40002, 2), largest_ilp(90266, 4)], name="g9_ilp_giant")
