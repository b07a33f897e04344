"""Helpers of the AIS tests: G19 cases as device inputs, comparison of fused children with the reference's."""
import numpy as np

from pymht_amd.ais import AisMessage, group_messages
from pymht_amd.models import pv

# float64 arithmetic in another order than OpenBLAS' (4x4 dgesv, gemm kernels): states and scores agree to ~1e-12 relative; the
# tolerances below are the north star's
X_REL = 1e-9
P_RTOL = 1e-9
NLLR_ATOL = 1e-9


def g19_case(g, ci):
    p = "c%d_" % ci
    msgs = [AisMessage(float(t), s, int(m), bool(h)) for t, s, m, h in zip(g[p + "ais_time"], g[p + "ais_state"], g[p + "ais_mmsi"], g[p + "ais_high"])]
    groups, nG, marr, order = group_messages(msgs, float(g[p + "t_leaf"]), float(g[p + "t_scan"]), pv)
    return dict(p=p, msgs=msgs, groups=groups, nG=nG, marr=marr, order=order, x=g[p + "x"], xf32=g[p + "xf32"], P=g[p + "P"], pd=g[p + "pd"],
                z=np.ascontiguousarray(g[p + "z"], dtype=np.float32), lambda_ais=float(g[p + "lambda_ais"]), eta2_ais=float(g[p + "eta2_ais"]),
                ptr=g[p + "ptr"], out_x=g[p + "out_x"], out_P=g[p + "out_P"], out_radar=g[p + "out_radar"], out_nllr=g[p + "out_nllr"],
                out_mmsi=g[p + "out_mmsi"])


def check_children(c, leaf, x, P, radar, nllr, mmsi):
    """children of leaf `leaf` of case `c` against the reference's: same children in the same order (decisions exact), values to tolerance"""
    a, b = int(c["ptr"][leaf]), int(c["ptr"][leaf + 1])
    assert len(radar) == b - a, (leaf, len(radar), b - a)
    assert np.array_equal(radar, c["out_radar"][a:b]) and np.array_equal(mmsi, c["out_mmsi"][a:b]), leaf
    if b == a:
        return 0.0
    scale = np.maximum(np.abs(c["out_x"][a:b]).max(axis=1, keepdims=True), 1.0)
    ex = float((np.abs(x - c["out_x"][a:b]) / scale).max())
    assert ex <= X_REL, (leaf, ex)
    assert np.allclose(P, c["out_P"][a:b], rtol=P_RTOL, atol=1e-12), leaf
    assert np.allclose(nllr, c["out_nllr"][a:b], rtol=0, atol=NLLR_ATOL), leaf
    return ex
