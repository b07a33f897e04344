"""ctypes binding of libmht_amd.so (include/mht_amd.h).  The library is the product: there is no
Python/NumPy fallback -- if it cannot be loaded, or no GPU is visible when a ctx is created, callers get
an exception."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libmht_amd.so")

MHT_OK, MHT_E_INVALID, MHT_E_HIP, MHT_E_CAPACITY, MHT_E_INFEASIBLE, MHT_E_LIMIT, MHT_E_STATE = 0, -1, -2, -3, -4, -5, -6
F_STATE_F32, F_SCORE_F32 = 1, 2


class MhtModel(C.Structure):
    _fields_ = [("A", C.c_float * 16), ("Q", C.c_float * 16), ("C", C.c_float * 8), ("R", C.c_float * 4),
                ("eta2", C.c_double), ("lambda_ex", C.c_double), ("default_pd", C.c_double),
                ("default_miss_nllr", C.c_double)]


class MhtNodes(C.Structure):
    _fields_ = [("x", C.c_void_p), ("cnllr", C.c_void_p), ("pd", C.c_void_p), ("parent", C.c_void_p),
                ("meas", C.c_void_p), ("cov", C.c_void_p), ("flags", C.c_void_p), ("P", C.c_void_p),
                ("cap", C.c_int32), ("cap_cov", C.c_int32)]


class MhtError(RuntimeError):
    def __init__(self, code, message):
        RuntimeError.__init__(self, "libmht_amd error %d: %s" % (code, message))
        self.code = code


_lib = None


def load(build_if_missing=True):
    """dlopen libmht_amd.so (building it in-tree with hipcc first if the sources are newer)."""
    global _lib
    if _lib is not None:
        return _lib
    if build_if_missing and os.environ.get("MHT_AMD_NO_BUILD", "0") != "1":
        from . import build
        try:
            build.build_library(verbose=False)
        except Exception:
            if not os.path.exists(LIB_PATH):
                raise
    if not os.path.exists(LIB_PATH):
        raise ImportError("libmht_amd.so is missing (%s); run `python -m pymht_amd.build`" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    lib.mht_last_error.restype = C.c_char_p
    lib.mht_abi_version.restype = C.c_int
    _lib = lib
    _declare(lib)
    return lib


def _declare(lib):
    vp, i32, i64, dbl = C.c_void_p, C.c_int32, C.c_int64, C.c_double
    sig = {
        "mht_create": [C.POINTER(vp), C.c_int, vp],
        "mht_destroy": [vp],
        "mht_synchronize": [vp],
        "mht_gate_scan": [vp, C.POINTER(MhtModel), C.POINTER(MhtNodes), vp, i32, vp, i32, C.POINTER(MhtNodes), vp,
                          vp, vp, C.POINTER(i32)],
    }
    for name, args in sig.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = C.c_int


def check(rc):
    if rc != MHT_OK:
        raise MhtError(rc, load().mht_last_error().decode("utf-8", "replace"))


def exported_symbols():
    """Symbols include/mht_amd.h declares (parsed from the header) -- used by the CPU test-suite."""
    import re
    hdr = open(os.path.join(HERE, "..", "include", "mht_amd.h")).read()
    return sorted(set(re.findall(r"\b(mht_[a-z_0-9]+)\s*\(", hdr)))
