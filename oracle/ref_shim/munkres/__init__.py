"""Shim: `munkres.munkres(cost) -> bool assignment matrix` (the cython-munkres-wrapper API the
reference imports at pymht/initiators/m_of_n.py:7), backed by scipy's Hungarian solver.
Used ONLY by oracle/gen_golden.py in the development container."""
import numpy as np
from scipy.optimize import linear_sum_assignment


def munkres(cost):
    cost = np.asarray(cost, dtype=np.float64)
    rows, cols = linear_sum_assignment(cost)
    out = np.zeros(cost.shape, dtype=bool)
    out[rows, cols] = True
    return out
