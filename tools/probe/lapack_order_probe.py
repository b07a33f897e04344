"""Development probe (any host): is np.linalg.inv of a float64 2x2 / 4x4 -- LAPACK dgesv inside the numpy wheel's OpenBLAS -- what
csrc/mht_la64.h::inv_lapack restates?  Random symmetric positive definite, nearly symmetric and general matrices against the host build
of the header (tests/hostmath/libhostmath.so: `python -c "import __graft_entry__ as g; g.build()"` or the g++ line in it).
usage: [OPENBLAS_CORETYPE=Haswell|SkylakeX|Zen] python tools/probe/lapack_order_probe.py
SkylakeX kernel set (the development container, the MI355X box's EPYC 9575F): 0 of 10 000 differ.  Haswell / Zen set: ~12 % (2x2) / ~96 % (4x4)
differ in the last place -- its dtrsm solve is not fused."""
import ctypes as C
import os
import numpy as np
from threadpoolctl import threadpool_info

print([d.get("architecture") for d in threadpool_info() if d.get("user_api") == "blas"])
lib = C.CDLL(os.path.join(os.path.dirname(__file__), "..", "..", "tests", "hostmath", "libhostmath.so"))
lib.mht_host_inv_lapack.restype = C.c_double
rng = np.random.default_rng(2)
for n in (2, 4):
    mats = [rng.normal(size=(n, n)) * rng.uniform(0.1, 10) + np.eye(n) * rng.uniform(0, 5) for _ in range(3000)]
    mats += [(lambda a: a @ a.T + np.eye(n) + rng.normal(size=(n, n)) * 1e-9)(rng.normal(size=(n, n))) for _ in range(2000)]
    ref = np.linalg.inv(np.array(mats))
    out, bad = np.zeros((n, n)), 0
    for m, r in zip(mats, ref):
        lib.mht_host_inv_lapack(n, np.ascontiguousarray(m).ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
        bad += not np.array_equal(out, r)
    print("%d x %d: %d of %d differ" % (n, n, bad, len(mats)))
