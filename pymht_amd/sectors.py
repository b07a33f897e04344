"""Several independent sensor sectors on ONE MI355X (BASELINE config 4 on a single device).

The reference has no notion of a sector: config 4 is simply four `Tracker` objects fed with four disjoint scan streams (nothing in
pymht/tracker.py:162-307 couples two instances).  `SectorGroup` keeps exactly that -- one drop-in `Tracker` per sector -- but steps
their device forests with ONE launch per stage for the whole group (`mht_group_step`, include/mht_amd.h): a single sector is a chain
of dependent memory round trips that leaves most of the chip idle, so S sectors cost about as much as one.
"""
import ctypes as C

import numpy as np

from . import _lib


class SectorGroup:
    """trackers: `pymht_amd.tracker.Tracker` objects on the same device with the same maxTargets / maxNodes / maxMeasurements / N."""

    def __init__(self, trackers):
        self.trackers = list(trackers)
        n = len(self.trackers)
        assert n >= 1
        self._lib = self.trackers[0]._lib
        handles = (C.c_void_p * n)(*[t._ctx.handle for t in self.trackers])
        self._h = C.c_void_p()
        _lib.check(self._lib.mht_group_create(C.byref(self._h), n, handles))
        self._zp = (C.c_void_p * n)()
        self._M = (C.c_int32 * n)()

    def step_dev(self, z_ptrs, Ms):
        """Raw replay: device pointers (ints) and measurement counts, one per sector; asynchronous, nothing is fetched."""
        for i, (p, m) in enumerate(zip(z_ptrs, Ms)):
            self._zp[i] = p
            self._M[i] = m
        rc = self._lib.mht_group_step(self._h, self._zp, self._M)
        if rc:
            _lib.check(rc)

    def addMeasurementLists(self, scanLists, pruneSimilar=False):
        """One `MeasurementList` per sector: all sectors' steps 1-6 in one batched launch set, then every tracker folds its own
        report and runs its own step 7 -- the result for every sector is what `Tracker.addMeasurementList` gives.
        pruneSimilar: bool or one bool per sector (tracker.py:230)."""
        assert len(scanLists) == len(self.trackers)
        ps = list(pruneSimilar) if hasattr(pruneSimilar, "__len__") else [bool(pruneSimilar)] * len(self.trackers)
        zs = [trk._stage_scan(sl, pruneSimilar=p) for trk, sl, p in zip(self.trackers, scanLists, ps)]
        self.step_dev([z.data_ptr() for z in zs], [int(z.shape[0]) for z in zs])
        for trk, sl in zip(self.trackers, scanLists):
            trk._after_step(sl, trk._staged_np, None)

    def close(self):
        if self._h:
            _lib.check(self._lib.mht_group_destroy(self._h))
            self._h = C.c_void_p()
