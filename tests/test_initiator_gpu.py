"""GPU: the device M-of-N initiator (csrc/mht_init.hip) against what the REAL reference initiator returned on the
same measurement streams (tests/golden/g8_initiator.npz, oracle/gen_initiator_golden.py): birth decisions, measurement numbers and
the sizes of the preliminary-track / initiator lists, float32 states and covariances -- all exactly (np.array_equal)."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def make_initiator(ctx, M_required, N_checks, max_meas=1024):
    from scipy.stats import chi2
    from pymht_amd import _lib
    from pymht_amd.models import pv
    from pymht_amd.models.constants import sigmaQ_tracker
    cfg = _lib.MhtInitiatorConfig()
    cfg.m_required, cfg.n_checks = M_required, N_checks
    cfg.max_meas, cfg.max_prelim, cfg.max_born = max_meas, 2048, 256
    cfg.v_max = 20.0
    cfg.gamma = float(chi2(df=2).ppf(0.99))
    cfg.merge_threshold = 4 * 2.5 ** 2
    cfg.default_pd = 0.8
    cfg.C[:] = np.asarray(pv.C_RADAR, np.float32).reshape(-1).tolist()
    cfg.R[:] = np.asarray(pv.R_RADAR(), np.float32).reshape(-1).tolist()
    cfg.P0[:] = np.asarray(pv.P0, np.float32).reshape(-1).tolist()
    cfg.sigma_q = float(sigmaQ_tracker)
    h = C.c_void_p()
    _lib.check(ctx.lib.mht_initiator_create(ctx.handle, C.byref(h), C.byref(cfg)))
    return h


def test_device_initiator_matches_reference_streams(gold_dir):
    import torch
    from pymht_amd import _lib
    from pymht_amd.device import Context
    g = np.load(os.path.join(gold_dir, "g8_initiator.npz"))
    ctx = Context(0)
    born_total = 0
    for c in range(int(g["n_cases"])):
        p = "c%d_" % c
        ini = make_initiator(ctx, int(g[p + "M"]), int(g[p + "N"]))
        for k in range(int(g[p + "n_scans"])):
            q = p + "s%02d_" % k
            z = np.ascontiguousarray(g[q + "z"], dtype=np.float32).reshape(-1, 2)
            zd = torch.from_numpy(z if len(z) else np.zeros((1, 2), np.float32)).cuda()
            _lib.check(ctx.lib.mht_initiator_step(ini, zd.data_ptr(), len(z), None, float(g[p + "times"][k])))
            x = np.zeros((256, 4)); P = np.zeros((256, 16), np.float32); m = np.zeros(256, np.int32)
            nb, npre, nseed = C.c_int32(0), C.c_int32(0), C.c_int32(0)
            pp = lambda a: a.ctypes.data_as(C.c_void_p)
            _lib.check(ctx.lib.mht_initiator_born(ini, 256, pp(x), pp(P), pp(m), C.byref(nb), C.byref(npre), C.byref(nseed)))
            n = nb.value
            want_m = g[q + "meas"]
            assert n == len(want_m), (c, k, n, len(want_m))
            # a merged target carries no measurement number in the reference (-1 in the fixture); the device reports 0
            assert np.array_equal(m[:n], np.where(want_m < 0, 0, want_m)), (c, k)
            # float32 birth states and covariances BIT FOR BIT: the device orders F.dot(state), C.dot(pred), K.dot(delta) as the host
            # BLAS's gemv does (csrc/mht_math.h::gemv_row), the products of 2-D arrays as its gemm does
            assert np.array_equal(x[:n], g[q + "x"].reshape(-1, 4).astype(np.float64)), (c, k)
            assert np.array_equal(P[:n].reshape(-1, 4, 4), g[q + "P"].reshape(-1, 4, 4)), (c, k)
            assert (npre.value, nseed.value) == (int(g[q + "n_prelim"]), int(g[q + "n_seeds"])), (c, k)
            born_total += n
        _lib.check(ctx.lib.mht_initiator_destroy(ini))
    assert born_total >= 50
    ctx.close()
