"""Constant-turn model, f32 matrices, state = [x, y, vx, vy, w, a]: position, velocity, turn rate w (rad/s) and its rate a -- the SIX-state
"CT" model BASELINE config 5 names.  The transition depends on the hypothesis: Phi(T, w) rotates the velocity by w*T and integrates the
arc, so every leaf carries its own A (and its own covariance chain: nothing is shared by value).

NOT part of the reference (its tracker is hard-wired to models/pv): what the reference does offer for such a model is
its dimension-generic per-hypothesis form -- kalman.predict_single(A, Q, x, P) + kalman.precalc(C, R, x_bar[None], P_bar[None]),
kalman.py:67-70, :82-101 -- and that is what the known-answer vectors tests/golden/g21_ct6.npz were made with (oracle/gen_golden.py:
gen_g21) and what the seam `mht_gate_scan_x` runs per leaf when mht_model_x.transition = 1.  Same public names as models/pv.py."""
import numpy as np
from .constants import defaultType, sigmaQ_tracker, sigmaR_RADAR_tracker


def _selector():
    c = np.zeros((2, 6), dtype=defaultType)
    c[0, 0] = c[1, 1] = 1.0
    return c


transition = "ct"      # Tracker: the transition is state dependent -- Phi(T, x[4]) per hypothesis (mht_forest_create_ex with MHT_FOREST_CT)
C_RADAR = _selector()
p = 2.5 ** 2
P0 = np.diag(np.array([p, p, 0.3 * p, 0.3 * p, 1e-4, 1e-6])).astype(defaultType)


def Phi(T, w=0.0):
    """Transition over T seconds at turn rate w: computed in float64, returned as float32 like pv.Phi.  |w| below 1e-9: the straight-line
    limits sin(wT)/w -> T, (1 - cos(wT))/w -> 0."""
    w, T = float(w), float(T)
    s, c = np.sin(w * T), np.cos(w * T)
    if abs(w) < 1e-9:
        sw, cw = T, 0.0
    else:
        sw, cw = s / w, (1.0 - c) / w
    a = np.identity(6, dtype=np.float64)
    a[0, 2], a[0, 3] = sw, -cw
    a[1, 2], a[1, 3] = cw, sw
    a[2, 2], a[2, 3] = c, -s
    a[3, 2], a[3, 3] = s, c
    a[4, 5] = T
    return a.astype(defaultType)


def Q(T, sigmaQ=sigmaQ_tracker):
    """White-noise acceleration on the velocity and white-noise jerk of the turn rate.  NOT pv.Q's entries: the position / velocity cross
    terms are T^3/2 here (G G^T with G = [T^2/2, T]) where pv.Q, following the reference, has T^3/3, and like pv.Q the whole matrix is
    scaled by sigmaQ, not its square.  The covariance prediction uses Phi(T, w) as A without the Jacobian with respect to w (not an EKF):
    the model is this build's own, validated against the reference's kalman functions fed with these matrices (G21, G23), not against
    a reference model -- the reference has none."""
    q = np.zeros((6, 6), dtype=np.float64)
    for axis in (0, 1):
        i, j = axis, 2 + axis
        q[i, i], q[i, j], q[j, i], q[j, j] = T ** 4 / 4.0, T ** 3 / 2.0, T ** 3 / 2.0, T ** 2
    q[4, 4], q[4, 5], q[5, 4], q[5, 5] = 1e-6 * T ** 4 / 4.0, 1e-6 * T ** 3 / 2.0, 1e-6 * T ** 3 / 2.0, 1e-6 * T ** 2
    return q.astype(defaultType) * sigmaQ


def R_RADAR(sigmaR=sigmaR_RADAR_tracker):
    return (np.identity(2) * np.power(sigmaR, 2)).astype(defaultType)
