"""TEST INFRASTRUCTURE (development container only).

Imports the *real* reference (`/root/reference/pymht`, read-only) through the shims in
oracle/ref_shim so that golden vectors can be generated from it.
Nothing here travels usefully to the GPU box: /root/reference does not exist there and
`load()` raises.  Only oracle/gen_golden.py calls this.
"""
import os
import sys

REFERENCE_ROOT = "/root/reference"


def available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "pymht"))


def load():
    """Return the dict of reference modules {tracker, pyTarget, kalman, pv, classDefinitions, m_of_n, pywraplp}."""
    if not available():
        raise RuntimeError("reference not present at %s (expected on the GPU box)" % REFERENCE_ROOT)
    os.environ.setdefault("MPLBACKEND", "Agg")
    import numpy as np
    for alias, target in (("bool", bool), ("int", int), ("float", float), ("Inf", np.inf), ("NaN", np.nan)):
        if alias not in np.__dict__:
            setattr(np, alias, target)
    shim = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_shim")
    for p in (shim, REFERENCE_ROOT):
        if p not in sys.path:
            sys.path.insert(0, p)
    real_version = np.__version__
    np.__version__ = "1.99.0"          # tracker.py:29-30 asserts minor >= 12
    try:
        import pymht.tracker as tracker
    finally:
        np.__version__ = real_version
    import pymht.pyTarget as pyTarget
    import pymht.utils.kalman as kalman
    import pymht.models.pv as pv
    import pymht.utils.classDefinitions as classDefinitions
    import pymht.initiators.m_of_n as m_of_n
    from ortools.linear_solver import pywraplp
    return dict(tracker=tracker, pyTarget=pyTarget, kalman=kalman, pv=pv,
                classDefinitions=classDefinitions, m_of_n=m_of_n, pywraplp=pywraplp)
