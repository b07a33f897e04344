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


def live_numpy_f64_is_pinned():
    """The float64 results of a LIVE oracle on this host are the fixtures' bit for bit only where the numpy wheel's OpenBLAS runs the kernel
    set the fixtures were recorded with (SkylakeX family: the development container and the MI355X box's EPYC 9575F): np.linalg.inv of a
    float64 matrix -- dgesv's trsm solve -- differs in the last place under the Haswell / Zen set (tools/probe/lapack_order_probe.py).
    The recorded traces (tests/golden) are compared exactly everywhere; the live-oracle AIS fuzz falls back to 1e-12 relative elsewhere."""
    try:
        from threadpoolctl import threadpool_info
        arch = [d.get("architecture", "") for d in threadpool_info() if d.get("user_api") == "blas"]
    except Exception:
        return False
    return bool(arch) and all(a in ("SkylakeX", "Cooperlake", "Sapphirerapids") for a in arch)
