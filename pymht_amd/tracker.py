"""`Tracker`: drop-in for `pymht.tracker.Tracker`, running on one MI355X.

Same constructor, `preInitialize`, `initiateTarget`, `addMeasurementList(scanList, aisList=AisMessageList(), **kw)`,
`getTrackNodes`, `getRuntimeAverage`, `runtimeLog`/`toc` keys and double-underscore attributes as the reference
(pymht/tracker.py:39-307).  Steps 1-6 of a scan (grow, cluster, optimise, terminate, N-scan
prune) run as two HIP launches (grow incl. the clustering's union-find, ILP; the report rides in the next scan's grow launch) on the
device-resident hypothesis forest of libmht_amd.so (include/mht_amd.h); the host sees one report per scan.  Step 7 (M-of-N initiation) runs on the device as well (`useInitiator`).

XML result export: `getScenarioElement` / `_storeTrackerArgs` / `_storeRun` (tracker.py:1469-1545).
AIS-aided tracking (tracker.py:394-396, :417-552; tracks started from AIS messages m_of_n.py:262-280): construct with
`aisAided=True` (and a finite `radarRange`), then `addMeasurementList(scan, aisList)`.  Not supported (raise): `dynamicWindow`.
There is no CPU fallback: without the HIP library or without a GPU the constructor raises.
"""
import ctypes as C
import logging
import os
import time

import numpy as np

from . import _lib
from .device import Context, make_model
from .initiators import m_of_n
from .pyTarget import Target
from .utils.classDefinitions import AisMessageList, MeasurementList  # noqa: F401
from .utils import xmlDefinitions as xmltags
from .utils.xmlDefinitions import activeTag, outofrangeTag, preinitializedTag, toolowscoreTag

log = logging.getLogger(__name__)

def _report_dtypes(nx):
    """NumPy views of mht_target_report / mht_birth_report for the nx-state build of the library (include/mht_amd.h: MHT_NX)."""
    rep = np.dtype([("id", "<i4"), ("status", "<i4"), ("sel_node", "<i4"), ("sel_meas", "<i4"),
                    ("new_index", "<i4"), ("root_scan", "<i4"), ("root_node", "<i4"), ("n_leaves", "<i4"),
                    ("sel_x", "<f8", (nx,)), ("sel_cnllr", "<f8"), ("score", "<f8"), ("root_cnllr", "<f8"),
                    ("root_x", "<f8", (nx,)), ("root_meas", "<i4"), ("cluster", "<i4")])
    birth = np.dtype([("id", "<i4"), ("meas", "<i4"), ("x0", "<f8", (nx,)), ("P0", "<f4", (nx * nx,))])
    return rep, birth


_REPORT_DTYPE, _BIRTH_DTYPE = _report_dtypes(4)
assert _REPORT_DTYPE.itemsize == C.sizeof(_lib.MhtTargetReport) and _BIRTH_DTYPE.itemsize == C.sizeof(_lib.MhtBirthReport)
assert _report_dtypes(6)[0].itemsize == C.sizeof(_lib.MhtTargetReport6) and _report_dtypes(6)[1].itemsize == C.sizeof(_lib.MhtBirthReport6)
_F_COV_F64 = 16      # include/mht_amd.h: MHT_F_COV_F64
_STATUS_TAG = {0: activeTag, 1: outofrangeTag, 2: toolowscoreTag, 3: toolowscoreTag}


class DeviceTarget(Target):
    """A `Target` that mirrors one node of the device forest (`scanNumber`, `_node` locate it)."""
    _tracker = None
    _node = -1
    _hist_mmsi = None      # AIS forest: the identity the node's track is bound to, as the device keeps it per node
    _P_lazy = False        # the covariance handed to the constructor is a placeholder: the node's own is fetched from the device when somebody reads it
    _P = None

    @property
    def P_0(self):
        if self._P_lazy:
            self._P_lazy = False
            t = self._tracker
            # (the constructor's placeholder is the tracker-wide default P_0, not this node's covariance: a node the device cannot answer for
            # any more -- it has left the ring, or reports are still pending -- has NO covariance to show, not a wrong one)
            self._P = None
            if t is not None and self._node >= 0 and self.scanNumber is not None and 0 <= len(t.__scanHistory__) - self.scanNumber < t._cfg.n_scan + 4:
                try:
                    P = t._window_chain(self.scanNumber, self._node)[4]
                    if len(P):
                        self._P = np.array(P[0]).reshape(t.nx, t.nx)
                except _lib.MhtError:
                    pass      # (the node has left the device ring)
        return self._P

    @P_0.setter
    def P_0(self, value):
        self._P, self._P_lazy = value, False

    def _getHistoricalMmsi(self):
        """pyTarget.py:297-302; the device carries the answer with every node (own identity, else the nearest AIS-updated ancestor's)."""
        if self._hist_mmsi is not None:
            return self._hist_mmsi
        return Target._getHistoricalMmsi(self)

    def getLeafNodes(self):
        if self._children is None and self.isRoot and self._tracker is not None:
            return self._tracker._leaf_views(self)
        return Target.getLeafNodes(self)


class Tracker():

    def __init__(self, model, radarPeriod, lambda_phi, lambda_nu, **kwargs):
        # Radar parameters (tracker.py:43-51)
        self.position = kwargs.get('position', np.array([0., 0.]))
        self.radarRange = kwargs.get('radarRange', float('inf'))
        self.radarPeriod = radarPeriod
        self.fixedPeriod = True
        self.default_P_d = kwargs.get('P_d', 0.8)
        assert self.default_P_d < 1 and self.default_P_d > 0, "Invalid P_d"
        # State space model (tracker.py:53-59)
        self.A = model.Phi(radarPeriod)
        self.C = model.C_RADAR
        self.P_0 = model.P0
        self.R_RADAR = model.R_RADAR()
        self.Q = model.Q(radarPeriod)
        # Target initiator (tracker.py:61-72)
        self.maxSpeedMS = kwargs.get('maxSpeedMS', 20)
        self.M_required = kwargs.get('M_required', 2)
        self.N_checks = kwargs.get('N_checks', 3)
        self.mergeThreshold = 4 * (model.sigmaR_RADAR_tracker ** 2)
        # Tracker storage (tracker.py:74-84): the forest lives on the device; the host keeps flat NumPy tables and
        # builds `Target` views only when somebody looks (properties __targetList__, __trackNodes__, ... below)
        self.__scanHistory__ = []
        self._scan_times = np.zeros(256)     # time of scan k at index k (index 0 = before the first scan); grows by doubling
        self.__aisHistory__ = []
        self.trackIdCounter = 0
        # Timing and logging (tracker.py:86-101)
        self._runtimeLog_ = {k: [] for k in ('Total', 'Process', 'Cluster', 'Optim', 'ILP-Prune', 'DynN', 'N-Prune',
                                           'Terminate', 'Init')}
        self.tic = {}
        self._toc_ = {}
        self._nOptimSolved_ = 0
        # Tracker parameters (tracker.py:103-118)
        self.pruneSimilar = kwargs.get('pruneSimilar', False)
        self.lambda_phi = lambda_phi
        self.lambda_nu = lambda_nu
        self.lambda_ex = lambda_phi + lambda_nu
        self.eta2 = kwargs.get('eta2', 5.99)
        self.eta2_ais = kwargs.get('eta2_ais', 9.45)
        N = kwargs.get('N', 5)
        self.N_max = N
        self.N = N
        self.scoreUpperLimit = -np.log(1 - self.default_P_d) * 0.8
        self.clnnrUpperLimit = 3.0
        self.pruneThreshold = kwargs.get("pruneThreshold", 4)
        self.targetSizeLimit = 3000      # (tracker.py:118: only used by the reference's dynamic window; kept for the XML settings block)
        self._prune_similar_on = False      # (what the device forest is currently set to; decided per scan like the reference does)
        # MI355X side
        # state dimension: 4 (the reference's models/pv) or 6 (e.g. pymht_amd.models.ca -- BASELINE config 5 names a six-state model; the
        # reference's kalman module is dimension-generic, its tracker and initiator are not): the six-state build of the library
        self.nx = int(np.asarray(self.C).shape[1])
        assert self.nx in (4, 6), "pymht_amd is built for 4- and 6-state models"
        self._REPORT_DTYPE, self._BIRTH_DTYPE = _report_dtypes(self.nx)
        self.useInitiator = kwargs.get('useInitiator', self.nx == 4)
        if self.useInitiator and self.nx != 4:
            raise NotImplementedError("the M-of-N initiator is the reference's 4-state one (m_of_n.py imports models/pv): useInitiator=False for a six-state model")
        self._ctx = Context(kwargs.get('device', 0), nx=self.nx)
        self._lib = self._ctx.lib
        self._model = make_model(self.A, self.Q, self.C, self.R_RADAR, self.eta2, self.lambda_ex, self.default_P_d)
        cfg = _lib.MhtForestConfig()
        cfg.max_targets = int(kwargs.get('maxTargets', 2048))
        cfg.max_nodes = int(kwargs.get('maxNodes', 1 << 19))
        cfg.max_meas = int(kwargs.get('maxMeasurements', 2048))
        cfg.n_scan = int(N)
        cfg.blp_max_iter = int(kwargs.get('blpMaxIter', 200))
        cfg.blp_node_limit = int(kwargs.get('blpNodeLimit', 1 << 20))
        self._blp_time_limit = kwargs.get('blpTimeLimit', None)      # seconds of branch and bound per cluster (None: exact, however long)
        cfg.score_limit = float(self.scoreUpperLimit)
        cfg.cnllr_limit = float(self.clnnrUpperLimit)
        cfg.radar_x, cfg.radar_y = float(self.position[0]), float(self.position[1])
        cfg.radar_range = float(self.radarRange)
        cfg.merge_threshold = float(self.mergeThreshold)
        self._cfg = cfg
        # AIS-aided tracking (tracker.py:394-396, :417-552): the forest has to be made for it (identities per node, AIS rows in the ILP);
        # `aisAided=True` is the switch -- the reference decides per call by looking at its aisList, a device forest cannot
        self._ais = bool(kwargs.get('aisAided', False))
        self._model_mod = model
        self.P_ais = 0.5                                    # tracker.py:109
        self._leaf_time = None                              # time of the current leaves (the last scan's, or the initial targets')
        if self._ais:
            if self.nx != 4:
                raise NotImplementedError("AIS messages report four states (models/ais.py): aisAided needs a 4-state model")
            if int(N) > 7:
                raise NotImplementedError("aisAided: N-scan window of at most 7")
            _lib.check(self._lib.mht_forest_create_ex(self._ctx.handle, C.byref(self._model), C.byref(cfg), 1))      # MHT_FOREST_AIS
        elif getattr(model, "transition", None) == "ct":
            # a state-dependent transition (pymht_amd/models/ct.py, BASELINE config 5's constant-turn model): Phi(T, w) per hypothesis, the
            # reference's per-hypothesis form kalman.predict_single + kalman.precalc (kalman.py:67-70, :82-101); self.A = Phi(T, 0) carries T
            if self.nx != 6:
                raise NotImplementedError("the constant-turn model has six states")
            _lib.check(self._lib.mht_forest_create_ex(self._ctx.handle, C.byref(self._model), C.byref(cfg), 2))      # MHT_FOREST_CT
        else:
            _lib.check(self._lib.mht_forest_create(self._ctx.handle, C.byref(self._model), C.byref(cfg)))
        if self._blp_time_limit is not None:
            _lib.check(self._lib.mht_forest_set_blp_time_limit(self._ctx.handle, 1e3 * float(self._blp_time_limit)))
        # Target initiator (tracker.py:61-72): on the device, behind every scan's commit (mht_forest_initiate)
        self.initiator = m_of_n.Initiator(self.M_required, self.N_checks, self.maxSpeedMS, self.C, self.R_RADAR, self.mergeThreshold,
                                          ctx=self._ctx, maxMeasurements=cfg.max_meas, default_pd=self.default_P_d) \
            if self.useInitiator else None
        # per-stage device times (toc['Process'], ['Cluster'], ['Optim'], ['N-Prune']: five HIP events per scan and a stream
        # synchronisation to read them): off by default, it keeps the host from running ahead of the device
        self._timing = bool(kwargs.get('deviceTiming', False))
        _lib.check(self._lib.mht_forest_set_timing(self._ctx.handle, int(self._timing)))
        # host mirror of the target list (one row per target, target-list order)
        self._tbl_ = np.zeros(0, dtype=self._REPORT_DTYPE)      # (only id, root_scan, root_node, root_meas, root_x, root_cnllr are read)
        self._sel_ = None            # report records of the live targets after the last scan (selected leaves)
        self._labels = np.zeros(0, np.int64)
        self._labels_src = None      # the report rows of the last scan: their `cluster` column is read when somebody asks for the clusters
        self._history = []          # chunks of committed roots: dict(id, scan, node, meas, x, cnllr, time) arrays
        self._history_mmsi = []     # AIS forest: the identities of those roots, chunk by chunk (0 = none)
        self._last_ais_scan = -(1 << 30)      # last scan that carried AIS messages
        self._dead_chunks = []      # [records, scan time, scan number, z, window chains or None, ticket, fold number] of terminated tracks
        self._chain_pending = []    # the entries of _dead_chunks whose chains are still on their way (mht_forest_chains_begin)
        self._fold_no = 0
        self._birth = {}            # Target.ID -> (time, scan, x, P, meas, measurement, status)
        self._views = {}            # cache of lazily built views, dropped at every scan
        self._scanStatsLog = [] if kwargs.get('logScanStats', False) else None      # lastScanStats (+ nTargets) of every scan (property scanStatsLog)
        self._pendq = []            # the scans whose reports are still on their way, oldest first (folded by later scans or by the first look)
        self._dead = False          # a device step failed: the forest cannot go on
        self._staged = self._staged_prev = self._staged_np = None
        self._stats_ = {}
        self._scanrecs = []          # per-scan numbers of the folded reports that `toc` / `runtimeLog` / `lastScanStats` have not been built from yet
        self._rep = _lib.MhtScanReport(); self._rep_ref = C.byref(self._rep)
        self._rec_size = self._REPORT_DTYPE.itemsize
        self._no_recs = np.zeros(0, dtype=self._REPORT_DTYPE)

    # ------------------------------------------------------------------------------------------------
    def preInitialize(self, simList):
        """tracker.py:139-145: one root per ground-truth object of simList[0]."""
        # one batch: the device admits the candidates sequentially (each is tested against the leaves and against the
        # candidates admitted before it), exactly like calling initiateTarget one by one
        self._add_targets([Target(initialTarget.time, None, initialTarget.cartesianState(), self.P_0, status=preinitializedTag)
                           for initialTarget in simList[0]])

    def initiateTarget(self, newTarget):
        """tracker.py:147-160.  The neighbour test (pyTarget.py:181-189) runs on the device against the current
        leaves."""
        self._add_targets([newTarget])

    def _add_targets(self, targets):
        self._drain()
        n = len(targets)
        if n == 0:
            return []
        if self._ais:
            # the AIS-aided update steps every leaf from ITS time to the message's (tracker.py:448: node.time); the device keeps one time for
            # all leaves (the last scan's, or the initial targets'): a target of another time would get a wrong first time step
            lt = self._leaf_time if self._leaf_time is not None else float(targets[0].time)
            if any(float(t.time) != lt for t in targets):
                raise ValueError("aisAided tracker: an initiated target's time differs from the time of the current leaves (%r): "
                                 "the AIS time steps are built from one leaf time" % lt)
        x0 = np.ascontiguousarray(np.array([np.asarray(t.x_0, dtype=np.float64) for t in targets]).reshape(n, self.nx))
        P0 = np.ascontiguousarray(np.array([np.asarray(t.P_0, dtype=np.float32) for t in targets]).reshape(n, self.nx * self.nx))
        f32 = np.array([np.asarray(t.x_0).dtype == np.float32 for t in targets])
        flags = np.where(f32, _lib.F_STATE_F32 | _lib.F_SCORE_F32, 0).astype(np.uint8)
        pd = np.full(n, self.default_P_d, dtype=np.float64)
        meas = np.array([0 if t.measurementNumber is None else int(t.measurementNumber) for t in targets], dtype=np.int32)
        acc = np.zeros(n, dtype=np.uint8)
        ids = np.zeros(n, dtype=np.int32)
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        _lib.check(self._lib.mht_forest_add_targets(self._ctx.handle, n, p(x0), p(P0), p(flags), p(pd), p(meas), 1,
                                                    p(acc), p(ids)))
        scan = len(self.__scanHistory__)
        ok = acc.astype(bool)
        out = [t for t, a in zip(targets, ok) if a]
        if not out:
            return out
        rows = np.zeros(len(out), dtype=self._REPORT_DTYPE)
        rows["id"], rows["root_scan"], rows["root_node"], rows["root_meas"], rows["root_x"] = ids[ok], scan, -1, meas[ok], x0[ok]
        self._tbl_ = np.concatenate([self._tbl_, rows])
        for t, i in zip(out, ids[ok]):
            self._birth[int(i)] = (t.time, scan, t.x_0, t.P_0, t.measurementNumber, t.measurement, t.status)
            t.ID = int(i)
        if scan == 0 and self._leaf_time is None:
            self._leaf_time = float(out[0].time)
        self.trackIdCounter = int(ids[ok].max()) + 1
        self._views.clear()
        return out

    # ------------------------------------------------------------------------------------------------
    def addMeasurementList(self, scanList, aisList=AisMessageList(), **kwargs):
        """tracker.py:162-307 for the radar-only case.

        Returns as soon as the scan is queued on the device: steps 1-7 run there without a host round trip, and the scan's report
        travels to pinned host memory behind them.  It is folded into the host mirror by the next call or by the first look at
        the tracker's state (every attribute / method that exposes results waits for it), so a host that streams scans in
        overlaps its own bookkeeping with the device's work."""
        tic = {'Total': time.time(), '_call': time.perf_counter()}
        z = self._accept_scan(scanList, aisList, kwargs)
        tic['_print'] = {k: v for k, v in kwargs.items() if k in ("printTime", "printCluster", "printInfo", "on_color") and v}
        self._set_prune_similar(bool(kwargs.get('pruneSimilar', False)))      # tracker.py:230: a per-scan switch (the constructor's copy is never read)
        if aisList is not None and len(aisList) > 0:
            self._arm_ais(scanList, aisList, z.shape[0], bool(kwargs.get('aisInitialization', True)))
            self._last_ais_scan = len(self.__scanHistory__) + len(self._pendq) + 1
        # The host mirror is only touched once the device has accepted the scan: a rejected step (too many measurements, dead
        # forest) leaves the tracker exactly as it was.
        try:
            self._staged, self._staged_np = None, z      # (the library stages the scan in pinned memory and copies it itself)
            _lib.check(self._lib.mht_forest_scan(self._ctx.handle, self.initiator.handle if self.useInitiator else None,
                                                 z.__array_interface__['data'][0], z.shape[0], float(scanList.time)))
        except _lib.MhtError as e:
            if e.code != _lib.MHT_E_INVALID:
                self._dead = True
            elif self._ais and self.useInitiator:      # (a refused scan must not leave its messages with the initiator for the next one)
                self._lib.mht_initiator_set_ais(self.initiator.handle, None, 0, None)
            raise
        tic['_call'] = time.perf_counter() - tic['_call']      # host time of this call up to here (the scan is queued; nothing waited)
        self._leaf_time = float(scanList.time)
        self._queue_report(scanList, z, aisList, tic)

    def _arm_ais(self, scanList, aisList, nRadarMeas, ais_init):
        """The AIS messages of this scan, handed to the device in the order the reference walks them (pymht_amd/ais.py).  Needs the
        folded state of the scan before (lambda_ais counts the targets, tracker.py:438), so a scan with messages does not overlap
        with its predecessor's report."""
        from .ais import group_messages, initiator_messages
        self._drain()
        scanTime = float(scanList.time)
        assert all(float(m.time) < scanTime for m in aisList)                                   # tracker.py:180-182
        assert all(float(m.time) > scanTime - self.radarPeriod for m in aisList), str(scanTime) + str([m.time for m in aisList])
        mmsi = [m.mmsi for m in aisList]
        assert len(mmsi) == len(set(mmsi)), "Duplicate MMSI in aisList"                          # tracker.py:183-185
        if not np.isfinite(self.radarRange):
            raise ValueError("AIS-aided tracking needs a finite radarRange (lambda_ais = nTargets * P_ais / (pi * radarRange^2), tracker.py:438; "
                             "with the default inf the reference itself fails in kalman.nllr)")
        if ais_init and self.useInitiator:      # the messages no track takes start preliminary tracks (tracker.py:267-273, m_of_n.py:262-280)
            msgs_i = initiator_messages(aisList, scanTime)
            _lib.check(self._lib.mht_initiator_set_ais(self.initiator.handle, C.byref(msgs_i), len(aisList), None))
        nT = len(self._tbl_)
        if nT == 0 or self._leaf_time is None:
            return                                                                               # (no leaves: nothing to fuse)
        lambda_ais = (nT * self.P_ais) / (np.pi * self.radarRange ** 2)
        groups, nG, msgs, order = group_messages(aisList, self._leaf_time, scanTime, self._model_mod)
        _lib.check(self._lib.mht_forest_set_ais(self._ctx.handle, C.byref(groups), nG, C.byref(msgs), len(order), float(self.eta2_ais), float(lambda_ais)))

    def _set_prune_similar(self, want):
        if want != self._prune_similar_on:
            _lib.check(self._lib.mht_forest_set_prune_similar(self._ctx.handle, float(self.pruneThreshold) if want else 0.0))
            self._prune_similar_on = want

    def _after_step(self, scanList, z, aisList, tic=None):
        """Behind the device step: step 7 on the device, then the report starts its way to the host."""
        if self.useInitiator:
            # 7 -- Initiate new tracks (tracker.py:264-278): on the device, right behind the scan's commit; what it gave birth to
            # comes back with the scan's report
            _lib.check(self._lib.mht_forest_initiate(self._ctx.handle, self.initiator.handle,
                                                     self._staged.data_ptr() if self._staged is not None else None, z.shape[0],
                                                     float(scanList.time)))
        _lib.check(self._lib.mht_forest_report_begin(self._ctx.handle))
        self._queue_report(scanList, z, aisList, tic)

    def _queue_report(self, scanList, z, aisList, tic):
        self._pendq.append((scanList, z, aisList, tic if tic is not None else {'Total': time.time()}))
        # With the device initiator the report of scan k leaves the device inside the grow launch of scan k + 1 (queued by the NEXT
        # call): folding it in that next call would wait for that launch -- report -> fold -> queue -> launch -> report is a latency
        # loop through the host, 85 us per scan whatever the device needs.  The call for scan k + 2 folds it instead, behind its own
        # queueing: it arrived a scan ago and nobody waits.  (Without the initiator the report travels right behind its own scan.)
        lag = 2 if (self.useInitiator and not self._timing) else 1
        if len(self._pendq) > lag:
            prev = self._pendq.pop(0)
            self._finish_scan(*prev, which=len(self._pendq))
        if self._timing:          # (the stage times are read scan by scan)
            self._drain()

    def _drain(self):
        """Fold the reports of the scans that are still in flight, oldest first (waits for them)."""
        if len(self._pendq) > 1:      # (the last scan's report would only start its way to the host when it is asked for: start it now, fold the older one meanwhile)
            _lib.check(self._lib.mht_forest_report_begin(self._ctx.handle))
        while self._pendq:
            prev = self._pendq.pop(0)
            self._finish_scan(*prev, which=len(self._pendq))

    def _accept_scan(self, scanList, aisList, kwargs):
        """Argument checks of addMeasurementList; returns the scan as the (M,2) float32 array the device gates."""
        if self._dead:
            raise RuntimeError("this Tracker's device forest is dead (an earlier scan failed); create a new Tracker")
        if aisList is not None and len(aisList) > 0:
            if not self._ais:
                raise NotImplementedError("this Tracker was not made for AIS messages: pass aisAided=True to the constructor (tracker.py:417-552)")
            if kwargs.get('aisInitialization', True) and not self.useInitiator:
                pass      # (no initiator: nothing starts tracks, like a reference tracker whose initiator was taken out)
        if kwargs.get('dynamicWindow', False):
            raise NotImplementedError("dynamicWindow is not supported by pymht_amd")
        m = np.asarray(scanList.measurements)
        z = np.ascontiguousarray(m, dtype=np.float32).reshape(-1, 2)
        if m.dtype != np.float32 and m.size and not np.array_equal(z.astype(np.float64).reshape(m.shape), m.astype(np.float64)):
            # the reference gates in the dtype it is given (its simulator produces float32); the device takes float32
            log.warning("addMeasurementList: measurements are %s and not exactly representable as float32; they are rounded "
                        "(the reference would gate them in %s)", m.dtype, m.dtype)
        return z

    def _upload_scan(self, z):
        """The scan in device memory (a torch tensor of the tracker, kept alive until the scan after the next): device pointer."""
        import torch
        self._staged_np = z
        t = torch.from_numpy(z if z.size else np.zeros((1, 2), np.float32)).to(self._ctx.device, non_blocking=True)
        self._staged_prev, self._staged = self._staged, t
        return t.data_ptr()

    def _stage_scan(self, scanList, aisList=None, **kwargs):
        """SectorGroup: checks + the scan in device memory (a torch tensor kept alive until the next scan)."""
        z = self._accept_scan(scanList, aisList, kwargs)
        self._set_prune_similar(bool(kwargs.get('pruneSimilar', False)))
        self._upload_scan(z)
        return self._staged[:z.shape[0]] if z.size else self._staged[:0]

    def _finish_scan(self, scanList, z, aisList, tic, which=0):
        """The host side of a scan: wait for its report, fold it (and the device initiator's births) into the host mirror.

        Only what later scans need is done here -- the report's rows become the table of the live targets, births and terminated tracks
        are taken in -- and the scan's numbers are kept as one tuple; the dictionaries the reference exposes (`toc`, `runtimeLog`,
        `lastScanStats`, the scan log) are built from those tuples when somebody reads them (`_materialise`): a host that streams scans
        in pays a few microseconds per scan here, not thirty."""
        t_fold = time.perf_counter()
        t_init = time.time()
        self._fold_no += 1
        if self._chain_pending:      # (chains begun two folds ago were gathered in front of the scan this fold's report belongs to)
            self._resolve_chains(self._fold_no - 2)
        rep = self._rep
        rc = self._lib.mht_forest_report_get(self._ctx.handle, which, self._rep_ref)
        if rc == _lib.MHT_E_LIMIT:
            # soft: an ILP ran into the branch-and-bound node limit.  The selection it returned is feasible (not proven optimal)
            # and the device forest HAS advanced with it: fold the report like any other (the reference logs "Optim result NOT
            # optimal", tracker.py:1201, and carries on as well)
            log.warning("scan %d: %d ILP(s) hit the branch-and-bound node limit; their selection is feasible, not proven optimal",
                        rep.scan, rep.n_limit)
        elif rc:
            self._dead = True
            _lib.check(rc)
        hist = self.__scanHistory__
        hist.append(scanList)
        scanNumber = len(hist)
        if scanNumber >= len(self._scan_times):
            self._scan_times = np.concatenate([self._scan_times, np.zeros(len(self._scan_times))])
        scanTime = scanList.time
        self._scan_times[scanNumber] = scanTime
        self.__aisHistory__.append(aisList)
        nRadarMeas = z.shape[0]
        assert rep.scan == scanNumber
        nT = rep.n_targets
        # (one memcpy of the rows out of the pinned block, viewed as records: NumPy copies a structured array field by field, 12x slower)
        recs = np.frombuffer(C.string_at(rep.targets, nT * self._rec_size), dtype=self._REPORT_DTYPE) if nT else self._no_recs
        used_raw = C.string_at(rep.used, 8 * max(rep.used_words, 1))
        self._apply_report(recs, scanTime, scanNumber, z, rep.n_alive == nT)
        if rep.n_births:
            births = np.frombuffer(C.string_at(rep.births, rep.n_births * self._BIRTH_DTYPE.itemsize), dtype=self._BIRTH_DTYPE)
            self._apply_births(births, scanTime, scanNumber, z[self._unused_of(used_raw, nRadarMeas)])
        # toc['Total'] = what THIS scan cost: the host time of its addMeasurementList call + the device time of its stages + the host
        # time of folding its report.  NOT the wall time between the call and the fold: the report of scan k is folded by the call for
        # scan k+1 (or by the first look at the results), so that interval is the host's idle time between scans -- a real-time host
        # feeding one scan per radarPeriod would see Total ~ radarPeriod and a spurious "Did not pass real time demand".
        call = tic.get('_call', 0.0)
        stage = None
        if self._timing:
            ms = (C.c_float * 5)()
            _lib.check(self._lib.mht_forest_stage_times(self._ctx.handle, C.byref(ms), None))
            stage = (ms[0] * 1e-3, ms[1] * 1e-3, ms[2] * 1e-3, ms[3] * 1e-3, ms[4] * 1e-3)
        device = stage[4] if stage is not None else rep.t_scan * 1e-8
        now = time.perf_counter()
        total = call + device + (now - t_fold)
        # (rep is reused by the next fold: the numbers are taken out now)
        self._scanrecs.append((tic, rep.t_process, rep.t_cluster, rep.t_optim, rep.t_scan, stage, rep.n_ilp, rep.n_leaves_in, rep.n_children,
                               nRadarMeas, rep.n_leaves_out, rep.n_clusters, rep.n_branched, rep.blp_iters_max, rep.n_limit, used_raw,
                               len(self._tbl_), time.time() - t_init, total))
        if len(self._scanrecs) >= 256:      # (a host that streams for hours without reading toc / runtimeLog: the tuples -- each holds its scan's tic and used-measurement words -- do not pile up)
            self._materialise()
        if total > self.radarPeriod * 0.6 or tic.get('_print'):
            self._materialise()
            if total > 0.1 and os.environ.get("MHT_STALL_DEBUG") == "1":      # (development: which part of a scan was slow)
                self._stall_debug(scanNumber, call, now - t_fold)
            if total > self.radarPeriod:
                log.critical("Did not pass real time demand! Used {0:.0f}ms of {1:.0f}ms".format(total * 1000, self.radarPeriod * 1000))
            elif total > self.radarPeriod * 0.6:      # tracker.py:285-287
                log.warning("Did almost not pass real time demand! Used {0:.0f}ms of {1:.0f}ms".format(total * 1000, self.radarPeriod * 1000))
            # the reference's per-scan console output (tracker.py:222-223, :296-301), when the scan is folded
            kw = tic.get('_print')
            if kw:
                if kw.get("printCluster", False):
                    self.printClusterList(self.__clusterList__)
                if kw.get("printInfo", False):
                    print("Added scan number:", len(self.__scanHistory__), " \tnRadarMeas ", nRadarMeas, sep="")
                if kw.get("printTime", False):
                    self.printTimeLog(**kw)

    @staticmethod
    def _unused_of(used_raw, nRadarMeas):
        """bit j of the report's used-measurement words -> unused[j]"""
        return ~np.unpackbits(np.frombuffer(used_raw, dtype=np.uint8), bitorder="little")[:nRadarMeas].astype(bool)

    def _materialise(self):
        """The per-scan tuples `_finish_scan` left -> the reference's `toc` / `runtimeLog` / `lastScanStats` (and the scan log)."""
        recs, self._scanrecs = self._scanrecs, []
        if len(recs) > 1:
            # all but the newest scan only add a row to `runtimeLog` (and to the scan log when it is kept); `toc`, `tic`, `lastScanStats` are the newest
            # scan's.  One pass per column instead of a dictionary per scan: a streaming host pays 0.5 us per scan for this, not 2.7 (it runs every
            # 256 scans, in the middle of the stream)
            older = recs[:-1]
            rl = self._runtimeLog_
            rl['Total'].extend(r[18] for r in older)
            rl['Init'].extend(r[17] for r in older)
            rl['Process'].extend(r[1] * 1e-8 if r[5] is None else r[5][0] for r in older)
            rl['Cluster'].extend(r[2] * 1e-8 if r[5] is None else r[5][1] for r in older)
            rl['Optim'].extend(r[3] * 1e-8 if r[5] is None else r[5][2] for r in older)
            rl['N-Prune'].extend(0.0 if r[5] is None else r[5][3] for r in older)
            zeros = [0.0] * len(older)
            for k in ('ILP-Prune', 'DynN', 'Terminate'):
                rl[k].extend(zeros)
            if self._scanStatsLog is not None:
                for r in older:
                    self._scanStatsLog.append(dict(L=r[7], G=r[8] - r[7], M=r[9], leaves_out=r[10], clusters=r[11], ilp=r[6], branched=r[12], blp_iters_max=r[13],
                                                   limit=r[14], unused=self._unused_of(r[15], r[9]), nTargets=r[16]))
            recs = recs[-1:]
        for (tic, t_process, t_cluster, t_optim, t_scan, stage, n_ilp, n_leaves_in, n_children, nRadarMeas, n_leaves_out, n_clusters, n_branched,
             blp_iters_max, n_limit, used_raw, n_tbl, t_init, total) in recs:
            self.tic = tic
            # Per-stage times (tracker.py:192-294: the reference's own per-scan metric).  The kernels stamp the GPU's wall clock (10 ns
            # ticks) at the start of every stage and the commit files the differences in the report: no HIP events, no host cost, present
            # on every scan.  Track termination and the N-scan prune decision are part of the ILP launch (toc['Optim']); the target-side
            # commit rides in the next scan's grow launch (toc['N-Prune'] = 0 unless deviceTiming measures it with HIP events).
            toc = {'Process': t_process * 1e-8, 'Cluster': t_cluster * 1e-8, 'Optim': t_optim * 1e-8, 'Terminate': 0.0, 'N-Prune': 0.0,
                   'Device': t_scan * 1e-8, 'ILP-Prune': 0.0, 'DynN': 0.0, 'Init': t_init, 'Total': total}
            if stage is not None:
                toc['Process'], toc['Cluster'], toc['Optim'], toc['N-Prune'], toc['Device'] = stage
            self._toc_ = toc
            self._nOptimSolved_ = n_ilp
            for k, v in self._runtimeLog_.items():
                if k in toc:
                    v.append(toc[k])
            self._stats_ = dict(L=n_leaves_in, G=n_children - n_leaves_in, M=nRadarMeas, leaves_out=n_leaves_out, clusters=n_clusters, ilp=n_ilp,
                                branched=n_branched, blp_iters_max=blp_iters_max, limit=n_limit, unused=self._unused_of(used_raw, nRadarMeas))
            if self._scanStatsLog is not None:
                self._scanStatsLog.append(dict(self._stats_, nTargets=n_tbl))

    def _stall_debug(self, scanNumber, call, fold):
        self._materialise()
        t = self._toc_
        log.critical("scan %d slow: call %.1f ms, device %.1f ms (process %.1f, cluster %.1f, optim %.1f), fold + wait %.1f ms", scanNumber,
                     1e3 * call, 1e3 * t['Device'], 1e3 * t['Process'], 1e3 * t['Cluster'], 1e3 * t['Optim'], 1e3 * fold)
        dbg = np.zeros(8, dtype=np.uint64)
        self._lib.mht_forest_debug_read(self._ctx.handle, b"init_dbg", dbg.ctypes.data_as(C.c_void_p), 64)
        log.critical("init_dbg %s at t = %.1f s (process clock), pid %d", dbg.tolist(), time.process_time(), os.getpid())
        st2 = np.zeros(16, dtype=np.uint64)
        self._lib.mht_forest_debug_read(self._ctx.handle, b"status2", st2.ctypes.data_as(C.c_void_p), 128)
        log.critical("status2 words (overflow | n_children << 32, n_dead | timeout bits << 32): %s %s", [hex(int(x)) for x in st2[:2]], [hex(int(x)) for x in st2[8:10]])

    def _apply_births(self, births, scanTime, scanNumber, z_unused):
        """The device initiator's candidates that Tracker.initiateTarget's device twin admitted: append them to the host mirror."""
        ok = births["id"] >= 0
        if not ok.any():
            return
        b = births[ok]
        n = len(b)
        x0 = b["x0"].astype(np.float32)
        rows = np.zeros(n, dtype=self._REPORT_DTYPE)
        rows["id"], rows["root_scan"], rows["root_node"], rows["root_meas"], rows["root_x"] = b["id"], scanNumber, -1, b["meas"], b["x0"]
        self._tbl_ = np.concatenate([self._tbl_, rows])
        for i in range(n):
            m = int(b["meas"][i])
            self._birth[int(b["id"][i])] = (scanTime, scanNumber, x0[i], b["P0"][i].reshape(self.nx, self.nx).copy(), (m if m > 0 else None),
                                            (z_unused[m - 1] if m > 0 else None), activeTag)
        self.trackIdCounter = int(b["id"].max()) + 1
        self._views.clear()

    def _apply_report(self, recs, scanTime, scanNumber, z, all_alive=None):
        """Fold the scan report into the host tables (no per-target Python objects, no per-field copies: the table of the live
        targets IS the report's rows)."""
        prev = self._tbl_
        if self._views:
            self._views.clear()
        self._labels_src = recs
        alive = None
        if all_alive is None:
            alive = recs["status"] == 0
            all_alive = bool(alive.all())
        if all_alive:
            live = recs
            moved = live["root_scan"] != prev["root_scan"]
        else:
            if alive is None:
                alive = recs["status"] == 0
            # terminated tracks keep their whole history (the reference's _pruneEverythingExceptHistory): the window ancestors of the
            # last selected node are fetched now, while they are still in the device ring
            dead = recs[~alive]
            # (ONE gather launch queued behind the scans in flight, fetched two folds later -- it has long arrived then -- or by whoever
            # looks at the terminated tracks first: a fetch per track stopped the stream for a round trip through the device, 130 us each)
            ticket = C.c_int64(0)
            start = np.ascontiguousarray(dead["sel_node"], dtype=np.int32)
            _lib.check(self._lib.mht_forest_chains_begin(self._ctx.handle, scanNumber, start.ctypes.data_as(C.c_void_p), len(start), self._cfg.n_scan + 2,
                                                         1 if self._ais else 0, C.byref(ticket)))
            entry = [dead, scanTime, scanNumber, z, None, ticket.value, self._fold_no]
            self._dead_chunks.append(entry)
            self._chain_pending.append(entry)
            live = recs[alive]
            moved = live["root_scan"] != prev["root_scan"][alive]
        if moved.any():      # the root of these targets advanced: the new roots join the committed history
            chunk = live if moved.all() else live[moved]
            self._history.append(chunk)
            if self._ais:        # their identities (pyTarget.py:34: Target.mmsi), read while their layers are still in the device ring
                mm = np.zeros(len(chunk), dtype=np.int64)
                # (no message in the last n_scan + 4 scans: no node of the window was updated with one -- nothing to read)
                if scanNumber - self._last_ais_scan <= self._cfg.n_scan + 4:
                    for sc in np.unique(chunk["root_scan"]):
                        if sc >= 1 and scanNumber - int(sc) < self._cfg.n_scan + 3:
                            rows = np.where((chunk["root_scan"] == sc) & (chunk["root_node"] >= 0))[0]
                            if len(rows):      # a gather of these nodes on the device (they sit all over the layer's index space)
                                nodes = np.ascontiguousarray(chunk["root_node"][rows], dtype=np.int32)
                                got = np.zeros(len(rows), dtype=np.int32)
                                _lib.check(self._lib.mht_forest_read_mmsi_nodes(self._ctx.handle, int(sc), len(rows), nodes.ctypes.data_as(C.c_void_p),
                                                                                got.ctypes.data_as(C.c_void_p), None))
                                mm[rows] = got
                self._history_mmsi.append(mm)
        self._tbl_ = live
        self._sel_ = (live, scanTime, scanNumber, z)

    # ---- results: every look at them first folds the report that is still in flight ---------------------------------------
    @property
    def lastScanStats(self):
        self._drain()
        if self._scanrecs:
            self._materialise()
        return self._stats_

    @property
    def scanStatsLog(self):
        self._drain()
        if self._scanrecs:
            self._materialise()
        return self._scanStatsLog

    @property
    def toc(self):
        self._drain()
        if self._scanrecs:
            self._materialise()
        return self._toc_

    @property
    def runtimeLog(self):
        self._drain()
        if self._scanrecs:
            self._materialise()
        return self._runtimeLog_

    @property
    def nOptimSolved(self):
        self._drain()
        if self._scanrecs:
            self._materialise()
        return self._nOptimSolved_

    @property
    def _sel(self):
        self._drain()
        return self._sel_

    @property
    def _tbl(self):
        self._drain()
        return self._tbl_

    # ---- lazily built views (the reference's attributes, tracker.py:74-84) -----------------------------------------
    def _mmsi_layer(self, scanNumber):
        """AIS forest: (mmsi, bound identity) of every node of a layer of the window, read once per scan and cached with the views."""
        key = ("_mmsi", scanNumber)
        v = self._views.get(key)
        if v is None:
            n = int(self._cfg.max_nodes)
            mm, hist = np.zeros(n, dtype=np.int32), np.zeros(n, dtype=np.int32)
            _lib.check(self._lib.mht_forest_read_mmsi(self._ctx.handle, int(scanNumber), 0, n, mm.ctypes.data_as(C.c_void_p), hist.ctypes.data_as(C.c_void_p)))
            v = self._views[key] = (mm, hist)
        return v

    def _node_view(self, r, scanTime, scanNumber, z):
        m = int(r["sel_meas"])
        mmsi = hist = None
        if self._ais and int(r["sel_node"]) >= 0 and scanNumber == len(self.__scanHistory__):
            layer = self._mmsi_layer(scanNumber)
            mm, hist = int(layer[0][int(r["sel_node"])]), (int(layer[1][int(r["sel_node"])]) or None)
            if mm:
                mmsi, m = mm, (m if m > 0 else None)      # (an AIS-updated node without a radar measurement: measurementNumber None, tracker.py:520)
        node = DeviceTarget(scanTime, scanNumber, np.array(r["sel_x"]), self.P_0, ID=int(r["id"]), P_d=self.default_P_d,
                            measurementNumber=m, measurement=(z[m - 1] if m else None), mmsi=mmsi,
                            cumulativeNLLR=float(r["sel_cnllr"]), status=_STATUS_TAG[int(r["status"])])
        node._tracker, node._node, node._hist_mmsi = self, int(r["sel_node"]), hist
        node._P_lazy = int(r["sel_node"]) >= 0
        node._lazy_parent = self._make_parent_loader(int(r["id"]))
        return node

    @property
    def __trackNodes__(self):
        self._drain()
        v = self._views.get("track")
        if v is None:
            v = np.empty(len(self._tbl_["id"]), dtype=np.dtype(object))
            if self._sel_ is not None and len(self._sel_[0]) == len(v):
                live, scanTime, scanNumber, z = self._sel_
                for i in range(len(live)):
                    v[i] = self._node_view(live[i], scanTime, scanNumber, z)
                n0 = len(live)
            else:
                n0 = 0 if self._sel_ is None else len(self._sel_[0])
                if self._sel_ is not None:
                    live, scanTime, scanNumber, z = self._sel_
                    for i in range(n0):
                        v[i] = self._node_view(live[i], scanTime, scanNumber, z)
            roots = self.__targetList__
            for i in range(n0, len(v)):      # targets born after the last scan: their node is the root itself
                v[i] = roots[i]
            self._views["track"] = v
        return v

    @property
    def __targetList__(self):
        self._drain()
        v = self._views.get("roots")
        if v is None:
            tb = self._tbl_
            v = []
            for i in range(len(tb["id"])):
                tid = int(tb["id"][i])
                b = self._birth.get(tid)
                if b is not None and int(tb["root_scan"][i]) == b[1]:      # still the root it was born with
                    root = DeviceTarget(b[0], b[1], b[2], b[3], ID=tid, P_d=self.default_P_d, measurementNumber=b[4],
                                        measurement=b[5], status=b[6])
                else:
                    rs, rm, r_mmsi, r_hist = int(tb["root_scan"][i]), int(tb["root_meas"][i]), None, None
                    if self._ais and rs >= 1 and int(tb["root_node"][i]) >= 0 and len(self.__scanHistory__) - rs < self._cfg.n_scan + 4:
                        layer = self._mmsi_layer(rs)      # (the root's own identity and the one its track is bound to: per node on the device)
                        mm, r_hist = int(layer[0][int(tb["root_node"][i])]), (int(layer[1][int(tb["root_node"][i])]) or None)
                        if mm:
                            r_mmsi, rm = mm, (rm if rm > 0 else None)
                    root = DeviceTarget(float(self._scan_times[max(rs, 0)]), rs, tb["root_x"][i].copy(), self.P_0,
                                        ID=tid, P_d=self.default_P_d, measurementNumber=rm, mmsi=r_mmsi,
                                        cumulativeNLLR=float(tb["root_cnllr"][i]))
                    root._hist_mmsi = r_hist
                    root._lazy_parent = self._history_parent_loader(tid)
                root.isRoot, root._tracker, root._node = True, self, int(tb["root_node"][i])
                v.append(root)
            self._views["roots"] = v
        return v

    @property
    def __clusterList__(self):
        self._drain()
        labels = self._labels_src["cluster"] if self._labels_src is not None else self._labels
        return [np.where(labels == lab)[0] for lab in np.unique(labels)]

    @property
    def __terminatedTargets__(self):
        self._drain()
        self._resolve_chains()
        out = []
        for recs, scanTime, scanNumber, z, chains, _, _ in self._dead_chunks:
            for r, chain in zip(recs, chains):
                v = self._node_view(r, scanTime, scanNumber, z)
                tid, rs = int(r["id"]), int(r["root_scan"])
                hist = [c for c in self._history_chain(tid) if c.scanNumber <= rs]      # committed roots up to the root it died with
                self._link_chain(v, tid, chain, rs, int(r["root_node"]), hist[-1] if hist else None)
                out.append(v)
        return out

    @property
    def __targetWindowSize__(self):
        self._drain()
        return [self.N] * len(self._tbl_["id"])

    @property
    def __associatedMeasurements__(self):
        self._drain()
        """The association sets live on the device as (target, measurement node) edges; the host view is rebuilt from
        the leaves' ancestor chains on demand (slow path, for inspection only)."""
        return [root.getMeasurementSet() if root.trackHypotheses is not None else set() for root in self.__targetList__]

    # ------------------------------------------------------------------------------------------------
    def _history_chain(self, target_id):
        """Committed roots of one track, oldest first, as Target views linked through `parent`."""
        b = self._birth.get(target_id)
        chain = []
        if b is not None:
            first = DeviceTarget(b[0], b[1], b[2], b[3], ID=target_id, P_d=self.default_P_d, measurementNumber=b[4],
                                 measurement=b[5], status=b[6])
            first._tracker = self
            chain.append(first)
        for ci, ch in enumerate(self._history):
            for k in np.where(ch["id"] == target_id)[0]:
                sc = int(ch["root_scan"][k])
                zz = self.__scanHistory__[sc - 1].measurements if sc >= 1 else None
                m = int(ch["root_meas"][k])
                mmsi = None
                if self._ais and ci < len(self._history_mmsi) and int(self._history_mmsi[ci][k]):
                    mmsi, m = int(self._history_mmsi[ci][k]), (m if m > 0 else None)      # (an AIS-updated node without a radar measurement: measurementNumber None)
                v = DeviceTarget(float(self._scan_times[max(sc, 0)]), sc, ch["root_x"][k].copy(), self.P_0, ID=target_id, P_d=self.default_P_d,
                                 measurementNumber=m, measurement=(np.asarray(zz)[m - 1] if (zz is not None and m) else None), mmsi=mmsi,
                                 cumulativeNLLR=float(ch["root_cnllr"][k]))
                v._tracker, v._node = self, int(ch["root_node"][k])
                v.parent = chain[-1] if chain else None
                chain.append(v)
        return chain

    def _history_parent_loader(self, target_id):
        def load(view):
            chain = self._history_chain(target_id)
            older = [c for c in chain if c.scanNumber < view.scanNumber]
            return older[-1] if older else None
        return load

    def _window_chain(self, scan, node):
        """The ancestors of (scan, node) that are still in the device ring: arrays nodes, meas, x, cnllr, P (index 0 = the node)."""
        n_max = self._cfg.n_scan + 2
        nodes = np.zeros(n_max, dtype=np.int32)
        meas = np.zeros(n_max, dtype=np.int32)
        x = np.zeros((n_max, self.nx))
        cn = np.zeros(n_max)
        n = C.c_int32(0)
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        if self._ais:      # covariances in the dtype the reference gives every node: float64 behind an AIS update (MHT_F_COV_F64), float32 otherwise
            P64 = np.zeros((n_max, self.nx * self.nx))
            fl = np.zeros(n_max, dtype=np.uint8)
            _lib.check(self._lib.mht_forest_chain_f64(self._ctx.handle, scan, node, n_max, p(nodes), p(meas), p(x), p(cn), p(P64), p(fl), C.byref(n)))
            k = n.value
            return nodes[:k], meas[:k], x[:k], cn[:k], [P64[i] if (fl[i] & _F_COV_F64) else P64[i].astype(np.float32) for i in range(k)]
        P = np.zeros((n_max, self.nx * self.nx), dtype=np.float32)
        _lib.check(self._lib.mht_forest_chain(self._ctx.handle, scan, node, n_max, p(nodes), p(meas), p(x), p(cn), p(P), C.byref(n)))
        k = n.value
        return nodes[:k], meas[:k], x[:k], cn[:k], P[:k]

    def _resolve_chains(self, older_than=None):
        """Take the window chains of terminated tracks out of the library's pinned blocks (mht_forest_chains_fetch): all that are pending,
        or those begun before fold `older_than` (their gather launch ran long ago: no wait)."""
        while self._chain_pending and (older_than is None or self._chain_pending[0][6] < older_than):
            entry = self._chain_pending.pop(0)
            n_max = self._cfg.n_scan + 2
            p = lambda a: a.ctypes.data_as(C.c_void_p)
            chains = []
            for i in range(len(entry[0])):
                nodes = np.zeros(n_max, dtype=np.int32)
                meas = np.zeros(n_max, dtype=np.int32)
                x = np.zeros((n_max, self.nx))
                cn = np.zeros(n_max)
                n = C.c_int32(0)
                if self._ais:      # (dtypes per node as _window_chain)
                    P64 = np.zeros((n_max, self.nx * self.nx))
                    fl = np.zeros(n_max, dtype=np.uint8)
                    _lib.check(self._lib.mht_forest_chains_fetch(self._ctx.handle, entry[5], i, p(nodes), p(meas), p(x), p(cn), p(P64), p(fl), C.byref(n)))
                    k = n.value
                    chains.append((nodes[:k], meas[:k], x[:k], cn[:k], [P64[j] if (fl[j] & _F_COV_F64) else P64[j].astype(np.float32) for j in range(k)]))
                else:
                    P = np.zeros((n_max, self.nx * self.nx), dtype=np.float32)
                    _lib.check(self._lib.mht_forest_chains_fetch(self._ctx.handle, entry[5], i, p(nodes), p(meas), p(x), p(cn), p(P), None, C.byref(n)))
                    k = n.value
                    chains.append((nodes[:k], meas[:k], x[:k], cn[:k], P[:k]))
            entry[4] = chains

    def _link_chain(self, view, target_id, chain, root_scan, root_node, root_view):
        """Hang the window ancestors `chain` under `view`; where the chain reaches (root_scan, root_node) it continues with
        `root_view` (the committed history).  An ancestor at the root's scan that is NOT the root means the view was taken
        before a prune that cut its branch off: the chain ends there (parent None) instead of being joined to a foreign root."""
        nodes, meas, x, cn, P = chain
        if len(P):
            view.P_0 = P[0].reshape(self.nx, self.nx).copy()
        prev = view
        for k in range(1, len(nodes)):
            sc = view.scanNumber - k
            if sc == root_scan:
                prev._parent, prev._lazy_parent = (root_view if int(nodes[k]) == root_node or root_node < 0 else None), None
                return view._parent
            if sc < root_scan:
                break
            zz = self.__scanHistory__[sc - 1].measurements if sc >= 1 else None
            m = int(meas[k])
            mmsi = hist = None
            if self._ais and sc >= 1:      # the ancestor's own identity and the one its track is bound to (pyTarget.py:34, :297-302), per node on the device
                layer = self._mmsi_layer(sc)
                mm, hist = int(layer[0][int(nodes[k])]), (int(layer[1][int(nodes[k])]) or None)
                if mm:
                    mmsi, m = mm, (m if m > 0 else None)
            a = DeviceTarget(self.__scanHistory__[sc - 1].time if sc >= 1 else view.time, sc, x[k].copy(),
                             P[k].reshape(self.nx, self.nx).copy(), ID=target_id, P_d=self.default_P_d, measurementNumber=m,
                             measurement=(np.asarray(zz)[m - 1] if (zz is not None and m) else None), mmsi=mmsi,
                             cumulativeNLLR=float(cn[k]))
            a._tracker, a._node, a._hist_mmsi = self, int(nodes[k]), hist
            prev._parent, prev._lazy_parent = a, None
            prev = a
        prev._lazy_parent = None
        return view._parent

    def _make_parent_loader(self, target_id):
        def load(view):
            # ancestors inside the device window, then the committed root history kept on the host
            if len(self.__scanHistory__) - view.scanNumber >= self._cfg.n_scan + 4:      # (the device ring holds N+4 scans: mht_forest.hip RING_EXTRA)
                return None
            chain = self._window_chain(view.scanNumber, view._node)
            hit = np.where(self._tbl_["id"] == target_id)[0]
            if len(hit) == 0:
                return self._link_chain(view, target_id, chain, -1, -1, None)
            i = int(hit[0])
            return self._link_chain(view, target_id, chain, int(self._tbl_["root_scan"][i]), int(self._tbl_["root_node"][i]),
                                    self.__targetList__[i])
        return load

    def _leaf_snapshot(self):
        """All leaves of the forest, exported once per scan: cached with the other lazily built views (dropped by the next scan or
        birth), sized by the report's leaf count rather than by max_nodes (`Target.getLeafNodes` of every root goes through here)."""
        self._drain()
        snap = self._views.get("_leaves")
        if snap is not None:
            return snap
        leaves_out = self._scanrecs[-1][10] if self._scanrecs else self._stats_.get("leaves_out", 0)
        cap = int(min(self._cfg.max_nodes, max(4096, 2 * int(leaves_out) + 4 * len(self._tbl_) + 1024)))
        while True:
            snap = self._leaf_export(cap)
            if snap is not None:
                break
            cap = self._cfg.max_nodes
        self._views["_leaves"] = snap
        return snap

    def _leaf_export(self, cap):
        n = C.c_int32(0)
        x = np.zeros((cap, self.nx)); cn = np.zeros(cap)
        P = np.zeros((cap, self.nx * self.nx), dtype=np.float64 if self._ais else np.float32)
        meas = np.zeros(cap, dtype=np.int32); tgt = np.zeros(cap, dtype=np.int32); ids = np.zeros(cap, dtype=np.int32)
        node = np.zeros(cap, dtype=np.int32); fl = np.zeros(cap, dtype=np.uint8)
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        # (AIS forest: every covariance as float64 -- exact for the float32 ones; "Pf64" tells which leaves the reference carries in float64)
        export = self._lib.mht_forest_leaves_f64 if self._ais else self._lib.mht_forest_leaves
        _lib.check(export(self._ctx.handle, cap, p(x), p(P), p(cn), p(meas), p(tgt), p(ids), p(node), p(fl), C.byref(n)))
        if n.value > cap and cap < self._cfg.max_nodes:
            return None      # (more leaves than the estimate -- the export is truncated to the capacity: the caller retries with the full one)
        k = min(n.value, cap)
        out = dict(x=x[:k], P=P[:k].reshape(k, self.nx, self.nx), cnllr=cn[:k], meas=meas[:k], target=tgt[:k], ID=ids[:k],
                   node=node[:k], flags=fl[:k])
        if self._ais:      # identities; a node with one and no radar measurement gets meas = -1 (measurementNumber None)
            out["Pf64"] = (out["flags"] & _F_COV_F64) != 0
            scan = len(self.__scanHistory__)
            out["mmsi"] = self._mmsi_layer(scan)[0][out["node"]].astype(np.int64) if scan > 0 else np.zeros(k, dtype=np.int64)
            out["meas"] = np.where((out["meas"] == 0) & (out["mmsi"] != 0), -1, out["meas"]).astype(np.int32)
        return out

    def leafBatch(self):
        """All current leaves in target-list / DFS order (what the next scan will gate): dict of arrays."""
        return dict(self._leaf_snapshot())

    def _leaf_views(self, root):
        snap = self._leaf_snapshot()
        scan = len(self.__scanHistory__)
        t = self.__scanHistory__[-1].time if scan else root.time
        out = []
        for i in np.where(snap["ID"] == root.ID)[0]:
            if int(snap["node"][i]) == root._node and root.scanNumber == scan:
                out.append(root)
                continue
            m, mmsi = int(snap["meas"][i]), None
            if self._ais and int(snap["mmsi"][i]):
                mmsi, m = int(snap["mmsi"][i]), (m if m > 0 else None)
            Pi = snap["P"][i].copy() if (not self._ais or snap["Pf64"][i]) else snap["P"][i].astype(np.float32)
            v = DeviceTarget(t, scan, snap["x"][i].copy(), Pi, ID=root.ID, P_d=self.default_P_d,
                             measurementNumber=m, mmsi=mmsi, cumulativeNLLR=float(snap["cnllr"][i]))
            v._tracker, v._node = self, int(snap["node"][i])
            v._lazy_parent = self._make_parent_loader(root.ID)
            out.append(v)
        return out

    # ------------------------------------------------------------------------------------------------
    @property
    def nTargets(self):
        """len(__targetList__) without building the views."""
        self._drain()
        return len(self._tbl_["id"])

    def getTrackNodes(self):
        return self.__trackNodes__

    def synchronize(self):
        """Wait for everything queued on the device and fold it (reports are folded lazily otherwise)."""
        self._drain()
        self._ctx.synchronize()

    # ---- the reference's console log (tracker.py:129-137, :1402-1467): same columns, fed from the device's scan report ---------------
    def setHighPriority(self):
        import platform
        import psutil
        proc = psutil.Process()
        proc.nice(psutil.HIGH_PRIORITY_CLASS if platform.system() == "Windows" else 5)

    def getTimeLogHeader(self):
        cols = (('{:3} ', "Nr"), ('{:9} ', "Num Targets"), ('{:12} ', "Iteration (ms)"),
                ('({0:23} ', "(nMeasurements + nAisUpdates / nNodes) Process time ms"))
        return ("".join(f.format(v) for f, v in cols) + '({0:2}) {1:5}'.format("nClusters", 'Cluster') +
                '({0:3}) {1:6}'.format("nOptimSolved", 'Optim') + '{:4}'.format('DynN') + '{:5}'.format('N-Prune') +
                '{:3}'.format('Terminate') + '{:5}'.format('Init'))

    def getTimeLogString(self):
        """One line per scan like the reference's.  Stage times are the device's (with `deviceTiming=True`; otherwise only Total and
        Init are known); the node count of the reference's line is the number of hypotheses the scan produced (children of the grow
        kernel): the forest keeps no per-target node totals."""
        st = self.lastScanStats
        ms = {k: 1e3 * self.toc.get(k, 0.0) for k in ('Total', 'Process', 'Cluster', 'Optim', 'DynN', 'N-Prune', 'Terminate', 'Init')}
        return ('{:<3.0f} '.format(len(self.__scanHistory__)) + 'nTrack {:2.0f} '.format(len(self.__targetList__)) +
                'Total {0:6.0f} '.format(ms['Total']) +
                'Process({0:4.0f}+{1:<3.0f}/{2:6.0f}) {3:6.1f} '.format(st["M"], 0, st["L"] + st["G"], ms['Process']) +
                'Cluster({0:2.0f}) {1:5.1f} '.format(st["clusters"], ms['Cluster']) +
                'Optim({0:g}) {1:6.1f} '.format(self.nOptimSolved, ms['Optim']) + 'DynN {:4.1f} '.format(ms['DynN']) +
                'N-Prune {:5.1f} '.format(ms['N-Prune']) + 'Kill {:3.1f} '.format(ms['Terminate']) + 'Init {:5.1f}'.format(ms['Init']))

    def printTimeLog(self, **kwargs):
        late = self.toc['Total'] > self.radarPeriod
        print(("!! " if late else "   ") + self.getTimeLogString())      # (the reference colours the line with termcolor)

    def printTimeLogHeader(self):
        print(self.getTimeLogHeader())

    @staticmethod
    def printClusterList(clusterList):
        print("Clusters:")
        for i, cluster in enumerate(clusterList):
            print("Cluster ", i, " contains target(s):\t", cluster, sep="")

    def printTargetList(self, **kwargs):
        np.set_printoptions(precision=2, suppress=True)
        print("TargetList:")
        for i, target in enumerate(self.__targetList__):
            print("%3d: %s" % (i, target))
        print()

    # ---- XML result export (tracker.py:1469-1545): what the reference's evaluation scripts read ----------------------------------
    def getScenarioElement(self, **kwargs):
        import xml.etree.ElementTree as ET
        return ET.Element(xmltags.scenarioTag)

    def _storeTrackerArgs(self, scenarioElement, **kwargs):
        import xml.etree.ElementTree as ET
        for key, value in kwargs.items():
            scenarioElement.attrib[str(key)] = str(value)
        settings = ET.SubElement(scenarioElement, xmltags.trackerSettingsTag)
        for name, value in (("M_required", self.M_required), ("N_checks", self.N_checks), ("mergeThreshold", self.mergeThreshold),
                            ("ownPosition", self.position), ("radarRange", self.radarRange), ("radarPeriod", self.radarPeriod),
                            ("lambdaPhi", self.lambda_phi), ("lambdaNu", self.lambda_nu), ("lambdaEx", self.lambda_ex), ("eta2", self.eta2),
                            ("N_max", self.N_max), ("NLLR_upperLimit", self.scoreUpperLimit), ("pruneThreshold", self.pruneThreshold),
                            ("targetSizeLimit", self.targetSizeLimit), ("maxSpeedMS", self.maxSpeedMS)):
            ET.SubElement(settings, name).text = str(value)

    def _storeRun(self, scenarioElement, preInitialized=True, **kwargs):
        """One <Run>: the per-stage run times of every scan and one <Track> per live and per terminated target -- all states of
        the selected hypothesis' chain (preInitialized=True; without the reference's pykalman-smoothed copy) or its first and last."""
        import xml.etree.ElementTree as ET
        run = ET.SubElement(scenarioElement, xmltags.runTag)
        run.attrib[xmltags.iterationTag] = str(kwargs[xmltags.iterationTag] if xmltags.iterationTag in kwargs
                                               else len(scenarioElement.findall(xmltags.runTag)))
        if xmltags.seedTag in kwargs:
            run.attrib[xmltags.seedTag] = str(kwargs[xmltags.seedTag])
        prec = xmltags.timeLogPrecision
        runtime = ET.SubElement(run, xmltags.runtimeTag, attrib={xmltags.descriptionTag: "Per iteration", xmltags.precisionTag: str(prec)})
        for stage, values in self.runtimeLog.items():
            if not values:
                continue
            v = np.array(values)
            ET.SubElement(runtime, str(stage), attrib={xmltags.meanTag: str(round(np.mean(v), prec)), xmltags.minTag: str(round(np.min(v), prec)),
                                                       xmltags.maxTag: str(round(np.max(v), prec))}
                          ).text = np.array_str(v, precision=prec, max_line_width=999999)
        for nodes, extra in ((self.__trackNodes__, {}), (self.__terminatedTargets__, {xmltags.terminatedTag: True})):
            for node in nodes:
                if preInitialized:
                    node._storeNode(run, self.radarPeriod, **extra)
                else:
                    node._storeNodeSparse(run, **extra)
        return run

    def getRuntimeAverage(self):
        return {k: np.mean(np.array(v)) for k, v in self.runtimeLog.items() if len(v)}

    def _findClustersFromSets(self):
        return self.__clusterList__

    def close(self):
        self._pendq = []
        if self.initiator is not None:
            self.initiator.close()
            self.initiator = None
        self._ctx.close()
