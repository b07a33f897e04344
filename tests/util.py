"""Helpers shared by the tests (not part of the product)."""
import numpy as np


def gate_sets(row_ptr, col_idx):
    return [tuple(col_idx[row_ptr[i]:row_ptr[i + 1]].tolist()) for i in range(len(row_ptr) - 1)]


# |delta| bound on one NLLR increment: the reference's per-leaf constant ln(lambda_ex*sqrt(det(2 pi S))/P_d) is a
# float32 value produced by NumPy's (CPU-dispatch dependent, not correctly rounded) SIMD log; the HIP path rounds a
# double-precision log to float32 instead: at most 1 ulp(f32) apart at |value| < 8  -> 4.8e-7.
NLLR_ATOL = 5e-7


def flags_for(x):
    return np.full(x.shape[0], 3 if x.dtype == np.float32 else 0, dtype=np.uint8)
