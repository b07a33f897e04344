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
    """trackers: `pymht_amd.tracker.Tracker` objects on the same device; the radar-only ones (which share the batched launches) with
    the same maxTargets / maxNodes / maxMeasurements / N."""

    def __init__(self, trackers):
        self.trackers = list(trackers)
        assert len(self.trackers) >= 1
        # AIS-aided trackers (path records in two halves, identities per node: `fgrow_ais_kernel`) and constant-turn ones (per-hypothesis
        # transitions) have launches of their own: they are members of the group for the caller, and stepped one by one behind the
        # batched launch set of the others (nothing couples two sectors: the order does not matter)
        self._own = [bool(getattr(t, "_ais", False)) or getattr(getattr(t, "_model_mod", None), "transition", None) == "ct" for t in self.trackers]
        self._batched = [i for i, o in enumerate(self._own) if not o]
        n = len(self._batched)
        self._lib = self.trackers[self._batched[0] if n else 0]._lib      # (four- and six-state members load different libraries)
        self._h = C.c_void_p()
        if n:
            handles = (C.c_void_p * n)(*[self.trackers[i]._ctx.handle for i in self._batched])
            _lib.check(self._lib.mht_group_create(C.byref(self._h), n, handles))
        self._zp = (C.c_void_p * max(n, 1))()
        self._M = (C.c_int32 * max(n, 1))()

    def step_dev(self, z_ptrs, Ms):
        """Raw replay: device pointers (ints) and measurement counts, one per sector; asynchronous, nothing is fetched.  Only for groups
        whose members all share the batched launches (no AIS-aided or constant-turn member: those have launches of their own and are
        stepped through `addMeasurementLists`)."""
        if any(self._own) and len(z_ptrs) != len(self._batched):
            raise ValueError("SectorGroup.step_dev: %d of the %d members have launches of their own (AIS-aided / constant-turn): use addMeasurementLists" % (sum(self._own), len(self.trackers)))
        for i, (p, m) in enumerate(zip(z_ptrs, Ms)):
            self._zp[i] = p
            self._M[i] = m
        rc = self._lib.mht_group_step(self._h, self._zp, self._M)
        if rc:
            _lib.check(rc)

    def addMeasurementLists(self, scanLists, aisLists=None, pruneSimilar=False, **kwargs):
        """One `MeasurementList` per sector: all sectors' steps 1-6 in one batched launch set, then every tracker folds its own
        report and runs its own step 7 -- the result for every sector is what `Tracker.addMeasurementList` gives.
        aisLists: None, or one AIS message list (or None) per sector -- only for members made with aisAided=True (tracker.py:417-552).
        pruneSimilar: bool or one bool per sector (tracker.py:230).  Further keyword arguments (e.g. aisInitialization) go to the
        AIS-aided members' `addMeasurementList`."""
        assert len(scanLists) == len(self.trackers)
        ps = list(pruneSimilar) if hasattr(pruneSimilar, "__len__") else [bool(pruneSimilar)] * len(self.trackers)
        ais = list(aisLists) if aisLists is not None else [None] * len(self.trackers)
        assert len(ais) == len(self.trackers)
        for i in self._batched:
            if ais[i] is not None and len(ais[i]) > 0:
                raise NotImplementedError("sector %d was not made for AIS messages: pass aisAided=True to its Tracker" % i)
        if self._batched:
            zs = [self.trackers[i]._stage_scan(scanLists[i], pruneSimilar=ps[i]) for i in self._batched]
            self.step_dev([z.data_ptr() for z in zs], [int(z.shape[0]) for z in zs])
        try:
            for i, trk in enumerate(self.trackers):      # (the members with launches of their own queue theirs behind the group's)
                if self._own[i]:
                    if getattr(trk, "_ais", False):
                        trk.addMeasurementList(scanLists[i], ais[i], pruneSimilar=ps[i], **kwargs)
                    else:
                        trk.addMeasurementList(scanLists[i], pruneSimilar=ps[i])
        finally:      # (a member of its own that refuses its scan must not leave the batched members stepped on the device and not folded on the host)
            for i in self._batched:
                trk = self.trackers[i]
                trk._after_step(scanLists[i], trk._staged_np, None)

    def close(self):
        if self._h:
            _lib.check(self._lib.mht_group_destroy(self._h))
            self._h = C.c_void_p()
