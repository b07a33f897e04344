"""ctypes binding of libmht_amd.so (include/mht_amd.h).  The library is the product: there is no
Python/NumPy fallback -- if it cannot be loaded, or no GPU is visible when a ctx is created, callers get
an exception."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libmht_amd.so")

MHT_OK, MHT_E_INVALID, MHT_E_HIP, MHT_E_CAPACITY, MHT_E_INFEASIBLE, MHT_E_LIMIT, MHT_E_STATE = 0, -1, -2, -3, -4, -5, -6
F_STATE_F32, F_SCORE_F32 = 1, 2


def _model_type(nx):
    class _M(C.Structure):
        _fields_ = [("A", C.c_float * (nx * nx)), ("Q", C.c_float * (nx * nx)), ("C", C.c_float * (2 * nx)), ("R", C.c_float * 4),
                    ("eta2", C.c_double), ("lambda_ex", C.c_double), ("default_pd", C.c_double),
                    ("default_miss_nllr", C.c_double)]
    _M.__name__ = "MhtModel%d" % nx
    return _M


MhtModel = _model_type(4)       # mht_model of libmht_amd.so (MHT_NX = 4)
MhtModel6 = _model_type(6)      # ... of libmht_amd6.so (MHT_NX = 6)


def model_type(nx):
    return MhtModel if nx == 4 else MhtModel6


class MhtModelX(C.Structure):      # mht_model_x: dimension-generic model (nx = 4 or 6 states, 2 measurements)
    _fields_ = [("nx", C.c_int32), ("A", C.POINTER(C.c_float)), ("Q", C.POINTER(C.c_float)), ("C", C.POINTER(C.c_float)),
                ("R", C.POINTER(C.c_float)), ("eta2", C.c_double), ("lambda_ex", C.c_double),
                ("transition", C.c_int32), ("period", C.c_double)]      # transition = 1: constant turn, A per leaf (models/ct.py)


class MhtNodes(C.Structure):
    _fields_ = [("x", C.c_void_p), ("cnllr", C.c_void_p), ("pd", C.c_void_p), ("parent", C.c_void_p),
                ("meas", C.c_void_p), ("cov", C.c_void_p), ("flags", C.c_void_p), ("P", C.c_void_p),
                ("cap", C.c_int32), ("cap_cov", C.c_int32)]


class MhtForestConfig(C.Structure):
    _fields_ = [("max_targets", C.c_int32), ("max_nodes", C.c_int32), ("max_meas", C.c_int32), ("n_scan", C.c_int32),
                ("blp_max_iter", C.c_int32), ("blp_node_limit", C.c_int32), ("score_limit", C.c_double),
                ("cnllr_limit", C.c_double), ("radar_x", C.c_double), ("radar_y", C.c_double),
                ("radar_range", C.c_double), ("merge_threshold", C.c_double)]


def _report_type(nx):
    class _R(C.Structure):
        _fields_ = [("id", C.c_int32), ("status", C.c_int32), ("sel_node", C.c_int32), ("sel_meas", C.c_int32),
                    ("new_index", C.c_int32), ("root_scan", C.c_int32), ("root_node", C.c_int32), ("n_leaves", C.c_int32),
                    ("sel_x", C.c_double * nx), ("sel_cnllr", C.c_double), ("score", C.c_double), ("root_cnllr", C.c_double),
                    ("root_x", C.c_double * nx), ("root_meas", C.c_int32), ("cluster", C.c_int32)]
    return _R


MhtTargetReport = _report_type(4)
MhtTargetReport6 = _report_type(6)


class MhtScanReport(C.Structure):
    _fields_ = [("scan", C.c_int32), ("n_targets", C.c_int32), ("n_alive", C.c_int32), ("n_leaves_in", C.c_int32),
                ("n_children", C.c_int32), ("n_leaves_out", C.c_int32), ("n_clusters", C.c_int32), ("n_ilp", C.c_int32),
                ("n_branched", C.c_int32), ("n_limit", C.c_int32), ("blp_iters_max", C.c_int32), ("error", C.c_int32),
                ("used_words", C.c_int32), ("n_births", C.c_int32), ("pad", C.c_int32 * 2),
                ("t_process", C.c_int32), ("t_cluster", C.c_int32), ("t_optim", C.c_int32), ("t_scan", C.c_int32), ("used", C.c_void_p),
                ("targets", C.c_void_p), ("births", C.c_void_p)]


class MhtBirthReport(C.Structure):
    _fields_ = [("id", C.c_int32), ("meas", C.c_int32), ("x0", C.c_double * 4), ("P0", C.c_float * 16)]


class MhtBirthReport6(C.Structure):
    _fields_ = [("id", C.c_int32), ("meas", C.c_int32), ("x0", C.c_double * 6), ("P0", C.c_float * 36)]


class MhtInitiatorConfig(C.Structure):
    _fields_ = [("m_required", C.c_int32), ("n_checks", C.c_int32), ("max_meas", C.c_int32), ("max_prelim", C.c_int32),
                ("max_born", C.c_int32), ("v_max", C.c_double), ("gamma", C.c_double), ("merge_threshold", C.c_double),
                ("default_pd", C.c_double), ("C", C.c_float * 8), ("R", C.c_float * 4), ("P0", C.c_float * 16),
                ("sigma_q", C.c_float)]


class MhtError(RuntimeError):
    def __init__(self, code, message):
        RuntimeError.__init__(self, "libmht_amd error %d: %s" % (code, message))
        self.code = code


_libs = {}


def load(build_if_missing=True, nx=4):
    """dlopen libmht_amd.so (nx = 4) or libmht_amd6.so (nx = 6: the same sources compiled with -DMHT_NX=6), building it in-tree with
    hipcc first if the sources are newer."""
    if nx in _libs:
        return _libs[nx]
    from . import build
    LIB_PATH = build.lib_path(nx)
    if build_if_missing and os.environ.get("MHT_AMD_NO_BUILD", "0") != "1":
        try:
            build.build_library(verbose=False, nx=nx)
        except Exception:
            if not os.path.exists(LIB_PATH):
                raise
    if not os.path.exists(LIB_PATH):
        raise ImportError("%s is missing (%s); run `python -m pymht_amd.build`" % (os.path.basename(LIB_PATH), LIB_PATH))
    # PyTorch-ROCm bundles its own HIP runtime (torch/lib/libamdhip64.so).  If libmht_amd.so is loaded first it pulls in
    # /opt/rocm's copy, a later `import torch` then brings a second runtime into the process and one of the two reports
    # "no ROCm-capable device".  Importing torch first makes both resolve to the same runtime.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    lib.mht_last_error.restype = C.c_char_p
    lib.mht_abi_version.restype = C.c_int
    lib.nx = nx
    _libs[nx] = lib
    _declare(lib, nx)
    return lib


def _declare(lib, nx=4):
    vp, i32, i64, dbl = C.c_void_p, C.c_int32, C.c_int64, C.c_double
    MhtModel = model_type(nx)
    sig = {
        "mht_create": [C.POINTER(vp), C.c_int, vp],
        "mht_destroy": [vp],
        "mht_synchronize": [vp],
        "mht_gate_scan": [vp, C.POINTER(MhtModel), C.POINTER(MhtNodes), vp, i32, vp, i32, C.POINTER(MhtNodes), vp,
                          vp, vp, C.POINTER(i32)],
        "mht_gate_scan_x": [vp, C.POINTER(MhtModelX), i32, vp, vp, vp, vp, vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, C.POINTER(i32)],
        "mht_cluster": [vp, i32, i32, vp, vp],
        "mht_solve_blp": [vp, i32, i32, i32, i32, vp, vp, vp, i32, i32, vp, C.POINTER(dbl), C.POINTER(i32),
                          C.POINTER(i32), C.POINTER(i32)],
        "mht_prune": [vp, i32, vp, i32, vp, vp, vp, vp],
        "mht_forest_create_ex": [vp, vp, vp, C.c_uint32],
        "mht_forest_set_ais": [vp, vp, i32, vp, i32, dbl, dbl],
        "mht_forest_read_mmsi": [vp, i32, i32, i32, vp, vp],
        "mht_forest_read_mmsi_nodes": [vp, i32, i32, vp, vp, vp],
        "mht_initiator_set_ais": [vp, vp, i32, vp],
        "mht_fuse_ais": [vp, vp, i32, vp, vp, vp, vp, vp, vp, i32, vp, i32, dbl, dbl, vp, i32, vp, vp, vp, vp, vp, vp, i32, vp],
        "mht_fuse_ais_f64": [vp, vp, i32, vp, vp, vp, vp, vp, vp, i32, vp, i32, dbl, dbl, vp, i32, vp, vp, vp, vp, vp, vp, i32, vp],
        "mht_forest_create": [vp, C.POINTER(MhtModel), C.POINTER(MhtForestConfig)],
        "mht_forest_add_targets": [vp, i32, vp, vp, vp, vp, vp, i32, vp, vp],
        "mht_forest_add_targets_dev": [vp, i32, vp, vp, vp, vp, vp, i32, vp, vp],
        "mht_forest_step": [vp, vp, i32],
        "mht_forest_step_host": [vp, vp, i32],
        "mht_forest_report": [vp, C.POINTER(MhtScanReport)],
        "mht_forest_report_begin": [vp],
        "mht_forest_report_get": [vp, i32, C.POINTER(MhtScanReport)],
        "mht_forest_leaves": [vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, C.POINTER(i32)],
        "mht_forest_leaves_f64": [vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, C.POINTER(i32)],
        "mht_forest_set_prune_similar": [vp, dbl],
        "mht_forest_set_blp_time_limit": [vp, dbl],
        "mht_forest_chain": [vp, i32, i32, i32, vp, vp, vp, vp, vp, C.POINTER(i32)],
        "mht_forest_chain_f64": [vp, i32, i32, i32, vp, vp, vp, vp, vp, vp, C.POINTER(i32)],
        "mht_forest_chains_begin": [vp, i32, vp, i32, i32, i32, C.POINTER(C.c_int64)],
        "mht_forest_chains_fetch": [vp, C.c_int64, i32, vp, vp, vp, vp, vp, vp, C.POINTER(i32)],
        "mht_forest_debug_read": [vp, C.c_char_p, vp, i64],
        "mht_forest_set_timing": [vp, i32],
        "mht_forest_stage_times": [vp, C.POINTER(C.c_float * 5), C.POINTER(i32)],
        "mht_initiator_create": [vp, C.POINTER(vp), C.POINTER(MhtInitiatorConfig)],
        "mht_initiator_destroy": [vp],
        "mht_initiator_step": [vp, vp, i32, vp, dbl],
        "mht_initiator_born": [vp, i32, vp, vp, vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)],
        "mht_forest_initiate": [vp, vp, vp, i32, dbl],
        "mht_forest_scan": [vp, vp, vp, i32, dbl],
        "mht_forest_step_sharded_begin": [vp, vp, i32, i32, i32, vp],
        "mht_forest_step_sharded_end": [vp, vp],
        "mht_forest_sharded_words": [vp, i32, C.POINTER(i32)],
        "mht_forest_step_sharded_begin2": [vp, vp, i32, i32, i32, vp, i32],
        "mht_group_create": [C.POINTER(vp), i32, C.POINTER(vp)],
        "mht_group_step": [vp, C.POINTER(vp), C.POINTER(i32)],
        "mht_group_destroy": [vp],
    }
    for name, args in sig.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = C.c_int


def check(rc, lib=None):
    if rc != MHT_OK:
        libs = [lib] if lib is not None else (list(_libs.values()) or [load()])
        msgs = [l.mht_last_error().decode("utf-8", "replace") for l in libs]
        raise MhtError(rc, msgs[0] if len(msgs) == 1 else " | ".join("[%d-state build] %s" % (l.nx, m) for l, m in zip(libs, msgs) if m))


def exported_symbols():
    """Symbols include/mht_amd.h declares (parsed from the header) -- used by the CPU test-suite."""
    import re
    hdr = open(os.path.join(HERE, "..", "include", "mht_amd.h")).read()
    return sorted(set(re.findall(r"\b(mht_[a-z_0-9]+)\s*\(", hdr)))
