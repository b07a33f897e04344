"""Constant-velocity ("position-velocity") model matrices, f32, state = [x, y, vx, vy].

Same public names as the reference's pymht/models/pv.py (C_RADAR :7, P0 :13, Q :17, R_RADAR :26,
Phi :29) because `Tracker(model, ...)` reads them off the model object (tracker.py:54-65).
"""
import numpy as np
from .constants import defaultType, sigmaQ_tracker, sigmaR_RADAR_tracker, sigmaR_RADAR_true, sigmaQ_true  # noqa: F401


def _selector():
    c = np.zeros((2, 4), dtype=defaultType)
    c[0, 0] = c[1, 1] = 1.0
    return c


C_RADAR = _selector()
H_radar = C_RADAR
p = 2.5 ** 2
P0 = np.diag(np.array([p, p, 0.3 * p, 0.3 * p])).astype(defaultType)


def Phi(T):
    """State transition for a step of T seconds."""
    a = np.identity(4, dtype=np.float64)
    a[0, 2] = a[1, 3] = T
    return a.astype(defaultType)


def Q(T, sigmaQ=sigmaQ_tracker):
    """Process noise. As in the reference (pv.py:17-23) this is scaled by sigmaQ (not its square)
    and the velocity block is T^2 -- reproduced as written, not 'fixed'."""
    q = np.zeros((4, 4), dtype=np.float64)
    q[0, 0] = q[1, 1] = T ** 4. / 4.
    q[0, 2] = q[2, 0] = q[1, 3] = q[3, 1] = T ** 3. / 3.
    q[2, 2] = q[3, 3] = T ** 2.
    return q.astype(defaultType) * sigmaQ


def R_RADAR(sigmaR=sigmaR_RADAR_tracker):
    return (np.identity(2) * np.power(sigmaR, 2)).astype(defaultType)
