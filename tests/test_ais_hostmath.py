"""CPU: the AIS fusion arithmetic of the kernels (csrc/mht_ais_math.h, host build in tests/hostmath) against known-answer vectors
recorded from the reference's Tracker.__fuseRadarAndAis (tests/golden/g19_ais_fusion.npz, oracle/gen_golden.py::gen_g19)."""
import ctypes as C
import os

import numpy as np

from ais_util import g19_case, check_children
from pymht_amd.models import pv


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def test_fusion_matches_reference_vectors(gold_dir, hostmath):
    g = np.load(os.path.join(gold_dir, "g19_ais_fusion.npz"))
    hostmath.mht_host_fuse_ais.restype = C.c_int
    Cm, R = np.ascontiguousarray(pv.C_RADAR, dtype=np.float32), np.ascontiguousarray(pv.R_RADAR(), dtype=np.float32)
    lam = float(g["lambda_phi"]) + float(g["lambda_nu"])
    worst, total = 0.0, 0
    for ci in range(int(g["n_cases"])):
        c = g19_case(g, ci)
        order = np.array(c["order"], dtype=np.int64)
        for l in range(len(c["x"])):
            cap = 512
            ox, oP = np.zeros((cap, 4)), np.zeros((cap, 16))
            orad, omsg, onl = np.zeros(cap, dtype=np.int32), np.zeros(cap, dtype=np.int32), np.zeros(cap)
            x = np.ascontiguousarray(c["x"][l], dtype=np.float64)
            P = np.ascontiguousarray(c["P"][l], dtype=np.float32)
            n = hostmath.mht_host_fuse_ais(_p(Cm), _p(R), C.c_double(float(g["eta2"])), C.c_double(lam), C.byref(c["groups"]), c["nG"], C.byref(c["marr"]),
                                           int(c["xf32"][l]), _p(x), _p(P), C.c_double(float(c["pd"][l])), 0, C.c_double(c["eta2_ais"]),
                                           C.c_double(c["lambda_ais"]), _p(c["z"]), len(c["z"]), cap, _p(ox), _p(oP), _p(orad), _p(onl), _p(omsg))
            assert 0 <= n <= cap
            mmsi = np.array([c["msgs"][order[i]].mmsi for i in omsg[:n]], dtype=np.int64)
            worst = max(worst, check_children(c, l, ox[:n], oP[:n].reshape(-1, 4, 4), orad[:n].astype(np.int64), onl[:n], mmsi))
            total += n
        # the identity filter (pyTarget.py:269-272): a track bound to one ship only takes that ship's messages
        own = int(c["msgs"][0].mmsi)
        n_own = hostmath.mht_host_fuse_ais(_p(Cm), _p(R), C.c_double(float(g["eta2"])), C.c_double(lam), C.byref(c["groups"]), c["nG"], C.byref(c["marr"]),
                                           int(c["xf32"][0]), _p(np.ascontiguousarray(c["x"][0], dtype=np.float64)), _p(np.ascontiguousarray(c["P"][0], dtype=np.float32)),
                                           C.c_double(float(c["pd"][0])), own, C.c_double(c["eta2_ais"]), C.c_double(c["lambda_ais"]), _p(c["z"]), len(c["z"]),
                                           cap, _p(ox), _p(oP), _p(orad), _p(onl), _p(omsg))
        a, b = int(c["ptr"][0]), int(c["ptr"][1])
        assert n_own == int((c["out_mmsi"][a:b] == own).sum())
    assert total == 590
    print("worst score difference", worst)
