"""TEST INFRASTRUCTURE (development container only): tests/golden/g8_initiator.npz.

Feeds the REAL reference M-of-N initiator (pymht/initiators/m_of_n.py, imported through oracle/refimport.py) and the
host-side restatement shipped in pymht_amd/initiators/m_of_n.py with the same streams of unused radar measurements
(moving objects that get confirmed, objects that die out, clutter, close pairs that get merged) and records, per scan, what
the reference returned: the new targets (state, covariance, measurement number: float32 values bit for bit) and the sizes of
its preliminary-track and initiator lists.  Aborts if the restatement differs anywhere.  The fixture holds numbers only.

Run:  python oracle/gen_initiator_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import refimport  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def make_stream(seed, n_obj, n_scans, radius, clutter, p_d, period):
    rng = np.random.default_rng(seed)
    pos = rng.uniform(-radius, radius, size=(n_obj, 2))
    vel = rng.normal(0.0, 7.0, size=(n_obj, 2))
    if n_obj >= 4:                      # a close pair moving together: exercises the duplicate test and the merge
        pos[1] = pos[0] + np.array([3.0, -2.0])
        vel[1] = vel[0]
    alive_from = rng.integers(0, max(1, n_scans // 2), size=n_obj)
    scans, times = [], []
    for k in range(n_scans):
        pos = pos + period * vel
        seen = (rng.uniform(size=n_obj) <= p_d) & (k >= alive_from)
        det = pos[seen] + rng.normal(0.0, 2.5, size=(int(seen.sum()), 2))
        ncl = rng.poisson(clutter)
        cl = rng.uniform(-radius, radius, size=(ncl, 2))
        z = np.concatenate([det, cl], axis=0)
        rng.shuffle(z, axis=0)
        scans.append(np.ascontiguousarray(z, dtype=np.float32).reshape(-1, 2))
        times.append(1000.0 + (k + 1) * period)
    return scans, times


if __name__ == "__main__":
    mods = refimport.load()
    ref_m, ref_cd, ref_pv = mods["m_of_n"], mods["classDefinitions"], mods["pv"]
    from m_of_n_oracle import Initiator
    from pymht_amd.models import pv
    from pymht_amd.utils.classDefinitions import MeasurementList
    fx, case = {}, 0
    for seed, n_obj, n_scans, radius, clutter, p_d, period, M, N in [
            (1, 6, 14, 400.0, 4, 0.9, 2.5, 2, 3), (2, 12, 16, 600.0, 10, 0.8, 2.5, 2, 3), (3, 3, 10, 200.0, 0, 1.0, 1.0, 2, 3),
            (4, 20, 12, 500.0, 25, 0.7, 4.0, 2, 4), (5, 0, 8, 300.0, 6, 0.9, 2.5, 2, 3), (6, 8, 12, 300.0, 3, 0.95, 2.5, 3, 4)]:
        scans, times = make_stream(seed, n_obj, n_scans, radius, clutter, p_d, period)
        ref = ref_m.Initiator(M, N, 20, ref_pv.C_RADAR, ref_pv.R_RADAR(), 4 * 2.5 ** 2)
        own = Initiator(M, N, 20, pv.C_RADAR, pv.R_RADAR(), 4 * 2.5 ** 2)
        p = "c%d_" % case
        fx[p + "M"], fx[p + "N"], fx[p + "n_scans"] = M, N, n_scans
        fx[p + "times"] = np.asarray(times)
        n_born = 0
        for k, (z, t) in enumerate(zip(scans, times)):
            r_out = ref.processMeasurements(ref_cd.MeasurementList(t, z), [])
            o_out = own.processMeasurements(MeasurementList(t, z), [])
            rx = np.array([np.asarray(b.x_0, dtype=np.float32) for b in r_out], dtype=np.float32).reshape(-1, 4)
            rP = np.array([np.asarray(b.P_0, dtype=np.float32) for b in r_out], dtype=np.float32).reshape(-1, 4, 4)
            rm = np.array([-1 if b.measurementNumber is None else int(b.measurementNumber) for b in r_out], dtype=np.int64)
            ox = np.array([np.asarray(b.x_0, dtype=np.float32) for b in o_out], dtype=np.float32).reshape(-1, 4)
            oP = np.array([np.asarray(b.P_0, dtype=np.float32) for b in o_out], dtype=np.float32).reshape(-1, 4, 4)
            om = np.array([-1 if b.measurementNumber is None else int(b.measurementNumber) for b in o_out], dtype=np.int64)
            assert all(np.asarray(b.x_0).dtype == np.float32 for b in r_out), "reference births are float32 states"
            assert rx.tobytes() == ox.tobytes() and rP.tobytes() == oP.tobytes() and np.array_equal(rm, om), (case, k)
            assert len(ref.preliminary_tracks) == len(own.preliminary_tracks) and len(ref.initiators) == len(own.initiators), (case, k)
            q = p + "s%02d_" % k
            fx[q + "z"], fx[q + "x"], fx[q + "P"], fx[q + "meas"] = z, rx, rP, rm
            fx[q + "n_prelim"], fx[q + "n_seeds"] = len(ref.preliminary_tracks), len(ref.initiators)
            n_born += len(r_out)
        print("case %d: %d scans, %d objects, %d targets born, restatement identical" % (case, n_scans, n_obj, n_born))
        case += 1
    fx["n_cases"] = case
    np.savez_compressed(os.path.join(GOLD, "g8_initiator.npz"), **fx)
    print("wrote g8_initiator.npz")
