"""The reference's own (smoke-level) tests for this path, pointed at this build: tests/test_models.py and tests/test_kalman.py of
erikliland/pyMHT construct the CV model for T = 1, a batch of ten zero states with unit covariance, and call predict / precalc /
numpyFilter on it without asserting values.  Here the same inputs go through the mirrored model module, the oracle (CPU) and
the gate seam on the GPU -- and the values ARE compared."""
import numpy as np
import pytest

import mht_oracle as orc
from pymht_amd.models import pv

dT = 1.0
n = 10
x_0_list = np.zeros((n, 4))
P_0_list = np.array([np.diag([1.0, 1.0, 1.0, 1.0]).astype(np.float32)] * n)
A = np.array([[1.0, 0., dT, 0.], [0., 1.0, 0., dT], [0., 0., 1.0, 0.], [0., 0., 0., 1.0]], dtype=np.float32)
C = np.array([[1.0, 0., 0., 0.], [0., 1.0, 0., 0.]], dtype=np.float32)
Q = pv.Q(dT)
R = (np.eye(2) * 1.0).astype(np.float32)


def test_Q():
    assert pv.Q(1).shape == pv.Q(1, 2).shape == (4, 4) and pv.Q(1).dtype == np.float32
    assert np.array_equal(pv.Q(1, 2), pv.Q(1, 1) * 2)


def test_R():
    assert pv.R_RADAR().shape == pv.R_RADAR(2).shape == (2, 2)
    assert np.array_equal(pv.R_RADAR(2), np.eye(2, dtype=np.float32) * 4)


def test_Phi():
    assert pv.Phi(1).shape == pv.Phi(2.0).shape == (4, 4)
    assert np.array_equal(pv.Phi(dT), A)


def test_predict_precalc_filter_oracle():
    x_bar, P_bar = orc.kf_predict(A, Q, x_0_list, P_0_list)
    assert x_bar.shape == (n, 4) and P_bar.shape == (n, 4, 4)
    assert np.array_equal(x_bar, np.zeros((n, 4)))
    assert np.allclose(P_bar[0], A.dot(np.eye(4)).dot(A.T) + Q)
    z_hat, S, S_inv, K, P_hat = orc.kf_precalc(C, R, x_bar, P_bar)
    assert z_hat.shape == (n, 2) and S.shape == S_inv.shape == (n, 2, 2) and K.shape == (n, 4, 2) and P_hat.shape == (n, 4, 4)
    assert np.allclose(np.matmul(S, S_inv), np.eye(2), atol=1e-6)
    assert np.allclose(P_hat, np.transpose(P_hat, (0, 2, 1)), atol=1e-6)            # covariance stays symmetric


@pytest.mark.gpu
def test_predict_precalc_filter_device(gpu_ctx):
    from pymht_amd.device import make_model, process_leaf_nodes
    from util import flags_for
    z = np.array([[0.3, -0.2], [40.0, 40.0], [-1.0, 1.5]], dtype=np.float32)
    r = process_leaf_nodes(gpu_ctx, make_model(A, Q, C, R, 5.99, 1e-4, 0.8), x_0_list, P_0_list, np.zeros(n), np.full(n, 0.8),
                           flags_for(x_0_list), z)
    o = orc.process_leaves(A, Q, C, R, 5.99, 1e-4, x_0_list, P_0_list, [0.8] * n, z)
    assert np.array_equal(r["x_bar"], o["x_bar"]) and np.array_equal(r["P_bar"], o["P_bar"]) and np.array_equal(r["P_hat"], o["P_hat"])
    assert np.array_equal(r["col_idx"], np.concatenate(o["idx"])) and len(r["col_idx"]) == 2 * n       # two of the three measurements gate
    assert np.array_equal(r["x_hat"], np.concatenate(o["x_hat"], axis=0))
