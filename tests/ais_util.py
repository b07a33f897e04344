"""Helpers of the AIS tests: G19 cases as device inputs, comparison of fused children with the reference's."""
import numpy as np

from pymht_amd.ais import AisMessage, group_messages
from pymht_amd.models import pv

# The fused children's states and covariances are the reference's BIT FOR BIT: every float64 product in OpenBLAS' order, both inverses
# (np.linalg.inv of the 4x4 AIS and the 2x2 radar innovation covariance) as LAPACK dgesv runs them (csrc/mht_la64.h).  Scores to the NLLR
# tolerance: numpy's det is exp(sum(log|u_ii|)) and its float64 log a SIMD polynomial -- a few ulp of float64.
NLLR_ATOL = 1e-12


def g19_case(g, ci):
    p = "c%d_" % ci
    msgs = [AisMessage(float(t), s, int(m), bool(h)) for t, s, m, h in zip(g[p + "ais_time"], g[p + "ais_state"], g[p + "ais_mmsi"], g[p + "ais_high"])]
    groups, nG, marr, order = group_messages(msgs, float(g[p + "t_leaf"]), float(g[p + "t_scan"]), pv)
    return dict(p=p, msgs=msgs, groups=groups, nG=nG, marr=marr, order=order, x=g[p + "x"], xf32=g[p + "xf32"], P=g[p + "P"], pd=g[p + "pd"],
                z=np.ascontiguousarray(g[p + "z"], dtype=np.float32), lambda_ais=float(g[p + "lambda_ais"]), eta2_ais=float(g[p + "eta2_ais"]),
                ptr=g[p + "ptr"], out_x=g[p + "out_x"], out_P=g[p + "out_P"], out_radar=g[p + "out_radar"], out_nllr=g[p + "out_nllr"],
                out_mmsi=g[p + "out_mmsi"])


def check_children(c, leaf, x, P, radar, nllr, mmsi):
    """children of leaf `leaf` of case `c` against the reference's: same children in the same order, states and covariances bit for bit"""
    a, b = int(c["ptr"][leaf]), int(c["ptr"][leaf + 1])
    assert len(radar) == b - a, (leaf, len(radar), b - a)
    assert np.array_equal(radar, c["out_radar"][a:b]) and np.array_equal(mmsi, c["out_mmsi"][a:b]), leaf
    if b == a:
        return 0.0
    assert np.array_equal(x, c["out_x"][a:b]), (leaf, "states")
    assert np.array_equal(P, c["out_P"][a:b]), (leaf, "covariances")
    assert np.allclose(nllr, c["out_nllr"][a:b], rtol=0, atol=NLLR_ATOL), leaf
    return float(np.abs(nllr - c["out_nllr"][a:b]).max())
