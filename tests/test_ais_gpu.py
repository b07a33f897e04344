"""GPU: the AIS-aided path through the C ABI.  `mht_fuse_ais` (Tracker.__fuseRadarAndAis, tracker.py:417-552) against the
known-answer vectors recorded from the reference (G19): the same children in the same order -- which messages gate, which radar
measurements gate behind them, pure-AIS children, the identity filter -- and their states / covariances / scores to the float64
tolerance of tests/ais_util.py (the 4x4 dgesv and the gemm order of OpenBLAS are not restated in the float64 part)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_fuse_seam_matches_reference_vectors(gpu_ctx, gold_dir):
    from ais_util import g19_case, check_children
    from pymht_amd.device import fuse_radar_and_ais
    from pymht_amd.models import pv
    g = np.load(os.path.join(gold_dir, "g19_ais_fusion.npz"))
    lam = float(g["lambda_phi"]) + float(g["lambda_nu"])
    total = 0
    for ci in range(int(g["n_cases"])):
        c = g19_case(g, ci)
        n = len(c["x"])
        flags = np.where(c["xf32"], 1, 0).astype(np.uint8)          # MHT_F_STATE_F32
        for own_of in (None, int(c["msgs"][0].mmsi)):
            own = np.zeros(n, dtype=np.int32) if own_of is None else np.full(n, own_of, dtype=np.int32)
            r = fuse_radar_and_ais(gpu_ctx, pv, float(g["eta2"]), lam, c["x"], c["P"], c["pd"], flags, own, c["msgs"], float(g[c["p"] + "t_leaf"]),
                                   float(g[c["p"] + "t_scan"]), c["eta2_ais"], c["lambda_ais"], c["z"])
            for l in range(n):
                a, b = int(r["child_ptr"][l]), int(r["child_ptr"][l + 1])
                if own_of is None:
                    check_children(c, l, r["x"][a:b], r["P"][a:b], r["radar"][a:b], r["nllr"][a:b], r["mmsi"][a:b])
                    total += b - a
                else:          # only that ship's messages: the reference's children with the others filtered out
                    ra, rb = int(c["ptr"][l]), int(c["ptr"][l + 1])
                    keep = c["out_mmsi"][ra:rb] == own_of
                    assert np.array_equal(r["radar"][a:b], c["out_radar"][ra:rb][keep]) and np.all(r["mmsi"][a:b] == own_of)
                    assert np.allclose(r["x"][a:b], c["out_x"][ra:rb][keep], rtol=1e-9, atol=1e-9)
    assert total == 590


def test_fuse_seam_edge_cases(gpu_ctx):
    from pymht_amd.device import fuse_radar_and_ais
    from pymht_amd.ais import AisMessage
    from pymht_amd.models import pv
    x = np.array([[10.0, 20.0, 1.0, -1.0]])
    P = pv.P0[None].astype(np.float32)
    z = np.zeros((0, 2), dtype=np.float32)
    # no messages: no children; no leaves: an empty CSR
    r = fuse_radar_and_ais(gpu_ctx, pv, 5.99, 1e-4, x, P, [0.9], [0], [0], [], 0.0, 2.5, 9.45, 1e-7, z)
    assert list(r["child_ptr"]) == [0, 0]
    r = fuse_radar_and_ais(gpu_ctx, pv, 5.99, 1e-4, np.zeros((0, 4)), np.zeros((0, 4, 4), dtype=np.float32), [], [], [], [AisMessage(1.0, x[0], 257000001, True)],
                           0.0, 2.5, 9.45, 1e-7, z)
    assert list(r["child_ptr"]) == [0]
    # a message on top of the leaf, no radar measurement at all: one pure-AIS child
    r = fuse_radar_and_ais(gpu_ctx, pv, 5.99, 1e-4, x, P, [0.9], [0], [0], [AisMessage(1.0, [11.0, 19.0, 1.0, -1.0], 257000001, True)], 0.0, 2.5, 9.45, 1e-7, z)
    assert list(r["child_ptr"]) == [0, 1] and list(r["radar"]) == [-1] and list(r["mmsi"]) == [257000001]
