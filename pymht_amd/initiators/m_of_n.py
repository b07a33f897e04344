"""`Initiator`: the reference's M-of-N track initiator (pymht/initiators/m_of_n.py:215-478) on the MI355X.

Same constructor and `processMeasurements(radar_measurement_list, ais_measurement_list)` as the reference class (tracker.py:66-72,
:264-278), but the preliminary tracks, the initiators and the two global-nearest-neighbour assignments live in HBM and one scan is
one launch of `initiator_kernel` (csrc/mht_init.hip).  Inside `pymht_amd.tracker.Tracker` the same device object runs directly
behind the scan's commit (`mht_forest_initiate`): no host round trip; `processMeasurements` is the stand-alone entry.
There is no host implementation in the product: the NumPy restatement lives in oracle/m_of_n_oracle.py (test infrastructure).
"""
import ctypes as C

import numpy as np
from scipy.stats import chi2

from .. import _lib
from ..models import pv
from ..models.constants import sigmaQ_tracker
from ..pyTarget import Target

GATE_PROBABILITY = 0.99
GAMMA = float(chi2(df=2).ppf(GATE_PROBABILITY))      # m_of_n.py:12-16
import os
MAX_BORN = int(os.environ.get("MHT_MAX_BORN", "256"))      # confirmed tracks of one scan, before merging (more: MHT_E_CAPACITY); 256 = the report's BIRTH_CAP


class Initiator:
    def __init__(self, M, N, v_max, C_mat, R, mergeThreshold=5, ctx=None, maxMeasurements=2048, maxPreliminary=4096, default_pd=0.8, **kwargs):
        assert ctx is not None, "the device initiator needs a pymht_amd.device.Context (there is no host implementation)"
        self.M, self.N, self.v_max = M, N, v_max
        self.merge_threshold = mergeThreshold
        self._ctx, self._lib = ctx, ctx.lib
        cfg = _lib.MhtInitiatorConfig()
        cfg.m_required, cfg.n_checks = int(M), int(N)
        cfg.max_meas, cfg.max_prelim, cfg.max_born = int(maxMeasurements), int(maxPreliminary), MAX_BORN
        cfg.v_max, cfg.gamma, cfg.merge_threshold, cfg.default_pd = float(v_max), GAMMA, float(mergeThreshold), float(default_pd)
        cfg.C[:] = np.asarray(C_mat, np.float32).reshape(-1).tolist()
        cfg.R[:] = np.asarray(R, np.float32).reshape(-1).tolist()
        cfg.P0[:] = np.asarray(pv.P0, np.float32).reshape(-1).tolist()
        cfg.sigma_q = float(sigmaQ_tracker)
        self.handle = C.c_void_p()
        _lib.check(self._lib.mht_initiator_create(ctx.handle, C.byref(self.handle), C.byref(cfg)))
        self.n_preliminary = self.n_initiators = 0

    def processMeasurements(self, radar_measurement_list, ais_measurement_list=()):
        """m_of_n.py:233-244: the list holds the measurements no track gated; returns the new `Target`s."""
        import torch
        # (tracks started from AIS messages, m_of_n.py:262-280, go through the device initiator inside the forest: Tracker(aisAided=True)
        # hands the messages over with mht_initiator_set_ais; this host-driven seam of the initiator takes radar measurements only)
        assert len(ais_measurement_list) == 0, "Initiator.processMeasurements takes radar measurements; AIS-started tracks run through Tracker(aisAided=True)"
        z = np.ascontiguousarray(np.asarray(radar_measurement_list.measurements, dtype=np.float32)).reshape(-1, 2)
        zd = torch.from_numpy(z if len(z) else np.zeros((1, 2), np.float32)).to(self._ctx.device)
        _lib.check(self._lib.mht_initiator_step(self.handle, zd.data_ptr(), len(z), None, float(radar_measurement_list.time)))
        x = np.zeros((MAX_BORN, 4)); P = np.zeros((MAX_BORN, 16), np.float32); m = np.zeros(MAX_BORN, np.int32)
        nb, npre, nseed = C.c_int32(0), C.c_int32(0), C.c_int32(0)
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        _lib.check(self._lib.mht_initiator_born(self.handle, MAX_BORN, p(x), p(P), p(m), C.byref(nb), C.byref(npre), C.byref(nseed)))
        self.n_preliminary, self.n_initiators = npre.value, nseed.value
        out = []
        for i in range(nb.value):
            mi = int(m[i])
            out.append(Target(radar_measurement_list.time, None, x[i].astype(np.float32), P[i].reshape(4, 4).copy(),
                              measurementNumber=(mi if mi > 0 else None), measurement=(z[mi - 1] if mi > 0 else None)))
        return out

    def close(self):
        if self.handle:
            self._lib.mht_initiator_destroy(self.handle)
            self.handle = C.c_void_p()
