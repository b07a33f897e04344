"""`Tracker`: drop-in for `pymht.tracker.Tracker` on the radar-only path, running on one MI355X.

Same constructor, `preInitialize`, `initiateTarget`, `addMeasurementList(scanList, aisList=AisMessageList(), **kw)`,
`getTrackNodes`, `getRuntimeAverage`, `runtimeLog`/`toc` keys and double-underscore attributes as the reference
(pymht/tracker.py:39-307).  Steps 1-6 of a scan (grow, cluster, optimise, terminate, N-scan
prune) run as five HIP launches on the device-resident hypothesis forest of libmht_amd.so (include/mht_amd.h);
the host sees one report per scan.  Step 7 (M-of-N initiation, off the hot path) stays on the host.

Not supported (raise): AIS fusion (`aisList` non-empty; tracker.py:417-552), `dynamicWindow`, `pruneSimilar`.
There is no CPU fallback: without the HIP library or without a GPU the constructor raises.
"""
import ctypes as C
import logging
import time

import numpy as np

from . import _lib
from .device import Context, make_model
from .initiators import m_of_n
from .pyTarget import Target
from .utils.classDefinitions import AisMessageList, MeasurementList  # noqa: F401
from .utils.xmlDefinitions import activeTag, outofrangeTag, preinitializedTag, toolowscoreTag

log = logging.getLogger(__name__)

_REPORT_DTYPE = np.dtype([("id", "<i4"), ("status", "<i4"), ("sel_node", "<i4"), ("sel_meas", "<i4"),
                          ("new_index", "<i4"), ("root_scan", "<i4"), ("root_node", "<i4"), ("n_leaves", "<i4"),
                          ("sel_x", "<f8", (4,)), ("sel_cnllr", "<f8"), ("score", "<f8"), ("root_cnllr", "<f8"),
                          ("root_x", "<f8", (4,)), ("root_meas", "<i4"), ("cluster", "<i4")])
assert _REPORT_DTYPE.itemsize == C.sizeof(_lib.MhtTargetReport)
_STATUS_TAG = {0: activeTag, 1: outofrangeTag, 2: toolowscoreTag, 3: toolowscoreTag}


class DeviceTarget(Target):
    """A `Target` that mirrors one node of the device forest (`scanNumber`, `_node` locate it)."""
    _tracker = None
    _node = -1

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
        self.initiator = m_of_n.Initiator(self.M_required, self.N_checks, self.maxSpeedMS, self.C, self.R_RADAR,
                                          self.mergeThreshold)
        # Tracker storage (tracker.py:74-84)
        self.__targetList__ = []
        self.__targetWindowSize__ = []
        self.__scanHistory__ = []
        self.__associatedMeasurements__ = []
        self.__trackNodes__ = np.empty(0, dtype=np.dtype(object))
        self.__terminatedTargets__ = []
        self.__clusterList__ = []
        self.__aisHistory__ = []
        self.trackIdCounter = 0
        # Timing and logging (tracker.py:86-101)
        self.runtimeLog = {k: [] for k in ('Total', 'Process', 'Cluster', 'Optim', 'ILP-Prune', 'DynN', 'N-Prune',
                                           'Terminate', 'Init')}
        self.tic = {}
        self.toc = {}
        self.nOptimSolved = 0
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
        assert not self.pruneSimilar, "pruneSimilar is not supported by pymht_amd"
        # MI355X side
        self.useInitiator = kwargs.get('useInitiator', True)
        self._ctx = Context(kwargs.get('device', 0))
        self._lib = self._ctx.lib
        self._model = make_model(self.A, self.Q, self.C, self.R_RADAR, self.eta2, self.lambda_ex, self.default_P_d)
        cfg = _lib.MhtForestConfig()
        cfg.max_targets = int(kwargs.get('maxTargets', 2048))
        cfg.max_nodes = int(kwargs.get('maxNodes', 1 << 17))
        cfg.max_meas = int(kwargs.get('maxMeasurements', 2048))
        cfg.n_scan = int(N)
        cfg.blp_max_iter = int(kwargs.get('blpMaxIter', 200))
        cfg.blp_node_limit = int(kwargs.get('blpNodeLimit', 1 << 20))
        cfg.score_limit = float(self.scoreUpperLimit)
        cfg.cnllr_limit = float(self.clnnrUpperLimit)
        cfg.radar_x, cfg.radar_y = float(self.position[0]), float(self.position[1])
        cfg.radar_range = float(self.radarRange)
        cfg.merge_threshold = float(self.mergeThreshold)
        self._cfg = cfg
        _lib.check(self._lib.mht_forest_create(self._ctx.handle, C.byref(self._model), C.byref(cfg)))
        self._timing = bool(kwargs.get('deviceTiming', True))
        _lib.check(self._lib.mht_forest_set_timing(self._ctx.handle, int(self._timing)))
        self._history = {}          # Target.ID -> list of committed root views (oldest first)
        self._roots = []            # root views, target-list order
        self._last_report = None
        self.lastScanStats = {}

    # ------------------------------------------------------------------------------------------------
    def preInitialize(self, simList):
        """tracker.py:139-145: one root per ground-truth object of simList[0]."""
        for initialTarget in simList[0]:
            self.initiateTarget(Target(initialTarget.time, None, initialTarget.cartesianState(), self.P_0,
                                       status=preinitializedTag))

    def initiateTarget(self, newTarget):
        """tracker.py:147-160.  The neighbour test (pyTarget.py:181-189) runs on the device against the current
        leaves."""
        self._add_targets([newTarget])

    def _add_targets(self, targets):
        n = len(targets)
        if n == 0:
            return []
        x0 = np.ascontiguousarray(np.array([np.asarray(t.x_0, dtype=np.float64) for t in targets]).reshape(n, 4))
        P0 = np.ascontiguousarray(np.array([np.asarray(t.P_0, dtype=np.float32) for t in targets]).reshape(n, 16))
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
        out = []
        for q, t in enumerate(targets):
            if not acc[q]:
                log.debug("Discarded an initial target: " + str(t))
                continue
            v = DeviceTarget(t.time, scan, t.x_0, t.P_0, ID=int(ids[q]), P_d=self.default_P_d,
                             measurementNumber=t.measurementNumber, measurement=t.measurement, status=t.status,
                             cumulativeNLLR=t.cumulativeNLLR)
            v.isRoot = True
            v._tracker = self
            self.trackIdCounter = int(ids[q]) + 1
            self.__targetList__.append(v)
            self._roots.append(v)
            self.__associatedMeasurements__.append(set())
            self.__trackNodes__ = np.append(self.__trackNodes__, v)
            self.__targetWindowSize__.append(self.N)
            self._history[v.ID] = [v]
            out.append(v)
        return out

    # ------------------------------------------------------------------------------------------------
    def addMeasurementList(self, scanList, aisList=AisMessageList(), **kwargs):
        """tracker.py:162-307 for the radar-only case."""
        if aisList is not None and len(aisList) > 0:
            raise NotImplementedError("AIS fusion (tracker.py:417-552) is outside the MI355X hot path")
        if kwargs.get('dynamicWindow', False) or kwargs.get('pruneSimilar', False):
            raise NotImplementedError("dynamicWindow / pruneSimilar are not supported by pymht_amd")
        self.tic.clear()
        self.toc.clear()
        self.__scanHistory__.append(scanList)
        self.__aisHistory__.append(aisList)
        scanTime = scanList.time
        scanNumber = len(self.__scanHistory__)
        self.tic['Total'] = time.time()
        z = np.ascontiguousarray(np.asarray(scanList.measurements, dtype=np.float32).reshape(-1, 2))
        nRadarMeas = z.shape[0]
        # steps 1-6 on the device
        _lib.check(self._lib.mht_forest_step_host(self._ctx.handle, z.ctypes.data_as(C.c_void_p), nRadarMeas))
        rep = _lib.MhtScanReport()
        _lib.check(self._lib.mht_forest_report(self._ctx.handle, C.byref(rep)))
        assert rep.scan == scanNumber
        nT = rep.n_targets
        recs = np.ctypeslib.as_array(C.cast(rep.targets, C.POINTER(C.c_uint8)), shape=(nT * _REPORT_DTYPE.itemsize,)) \
            .view(_REPORT_DTYPE).copy() if nT else np.zeros(0, dtype=_REPORT_DTYPE)
        used_words = np.ctypeslib.as_array(C.cast(rep.used, C.POINTER(C.c_uint64)), shape=(max(rep.used_words, 1),)).copy()
        used = np.unpackbits(used_words.view(np.uint8), bitorder="little")[:nRadarMeas].astype(bool)
        unusedRadarMeasurementIndices = ~used
        if self._timing:
            ms = (C.c_float * 5)()
            _lib.check(self._lib.mht_forest_stage_times(self._ctx.handle, C.byref(ms), None))
            self.toc['Process'], self.toc['Cluster'], self.toc['Optim'] = ms[0] * 1e-3, ms[1] * 1e-3, ms[2] * 1e-3
            self.toc['Terminate'] = 0.0
            self.toc['N-Prune'] = ms[3] * 1e-3
            self.toc['Device'] = ms[4] * 1e-3
        self.toc['ILP-Prune'] = 0.0
        self.toc['DynN'] = 0.0
        self.nOptimSolved = rep.n_ilp
        self._apply_report(recs, rep, scanTime, scanNumber, z)
        # 7 -- Initiate new tracks (tracker.py:264-278), host side
        self.tic['Init'] = time.time()
        if self.useInitiator:
            unused = scanList.filterUnused(unusedRadarMeasurementIndices) if hasattr(scanList, "filterUnused") else \
                MeasurementList(scanTime, z[unusedRadarMeasurementIndices])
            new_initial_targets = self.initiator.processMeasurements(unused, [])
            self._add_targets(new_initial_targets)
        self.toc['Init'] = time.time() - self.tic['Init']
        self.toc['Total'] = time.time() - self.tic['Total']
        if self.toc['Total'] > self.radarPeriod:
            log.critical("Did not pass real time demand! Used {0:.0f}ms of {1:.0f}ms".format(
                self.toc['Total'] * 1000, self.radarPeriod * 1000))
        for k, v in self.runtimeLog.items():
            if k in self.toc:
                v.append(self.toc[k])
        self.lastScanStats = dict(L=rep.n_leaves_in, G=rep.n_children - rep.n_leaves_in, M=nRadarMeas,
                                  leaves_out=rep.n_leaves_out, clusters=rep.n_clusters, ilp=rep.n_ilp,
                                  branched=rep.n_branched, blp_iters_max=rep.blp_iters_max,
                                  unused=unusedRadarMeasurementIndices)

    def _apply_report(self, recs, rep, scanTime, scanNumber, z):
        """Rebuild the host-side views (__trackNodes__, __targetList__, clusters, terminated list) from the report."""
        nT = len(recs)
        # clusters: group by label, ordered by label, members ascending (tracker.py:972-974)
        labels = recs["cluster"] if nT else np.zeros(0, dtype=np.int32)
        self.__clusterList__ = [np.where(labels == lab)[0] for lab in np.unique(labels)]
        track_nodes, roots, windows, assoc = [], [], [], []
        for t in range(nT):
            r = recs[t]
            old_root = self._roots[t]
            m = int(r["sel_meas"])
            node = DeviceTarget(scanTime, scanNumber, np.array(r["sel_x"]), self.P_0, ID=int(r["id"]),
                                P_d=self.default_P_d, measurementNumber=m,
                                measurement=(z[m - 1] if m > 0 else None), cumulativeNLLR=float(r["sel_cnllr"]),
                                status=_STATUS_TAG[int(r["status"])])
            node._tracker, node._node = self, int(r["sel_node"])
            node._lazy_parent = self._make_parent_loader(int(r["id"]))
            node._score = float(r["score"])
            if int(r["status"]) != 0:
                log.info("Terminating track %d (%s)" % (int(r["id"]), node.status))
                self.__terminatedTargets__.append(node)
                continue
            if int(r["root_scan"]) != old_root.scanNumber:
                old_root.isRoot = False
                rm = int(r["root_meas"])
                root = DeviceTarget(self.__scanHistory__[int(r["root_scan"]) - 1].time if r["root_scan"] > 0 else old_root.time,
                                    int(r["root_scan"]), np.array(r["root_x"]), self.P_0, ID=int(r["id"]),
                                    P_d=self.default_P_d, measurementNumber=rm, cumulativeNLLR=float(r["root_cnllr"]))
                root._tracker, root._node, root.isRoot = self, int(r["root_node"]), True
                root.parent = old_root
                self._history[root.ID].append(root)
            else:
                root = old_root
            track_nodes.append(node)
            roots.append(root)
            windows.append(self.N)
            assoc.append(None)
        self._roots = roots
        self.__targetList__ = list(roots)
        self.__targetWindowSize__ = windows
        self.__associatedMeasurements__ = assoc
        arr = np.empty(len(track_nodes), dtype=np.dtype(object))
        for i, n in enumerate(track_nodes):
            arr[i] = n
        self.__trackNodes__ = arr
        self._last_report = recs

    # ------------------------------------------------------------------------------------------------
    def _make_parent_loader(self, target_id):
        def load(view):
            # ancestors inside the device window, then the committed root history kept on the host
            n_max = self._cfg.n_scan + 2
            nodes = np.zeros(n_max, dtype=np.int32)
            meas = np.zeros(n_max, dtype=np.int32)
            x = np.zeros((n_max, 4))
            cn = np.zeros(n_max)
            P = np.zeros((n_max, 16), dtype=np.float32)
            n = C.c_int32(0)
            p = lambda a: a.ctypes.data_as(C.c_void_p)
            if len(self.__scanHistory__) - view.scanNumber >= self._cfg.n_scan + 2:
                return None
            _lib.check(self._lib.mht_forest_chain(self._ctx.handle, view.scanNumber, view._node, n_max, p(nodes), p(meas),
                                                  p(x), p(cn), p(P), C.byref(n)))
            hist = self._history.get(target_id, [])
            root = hist[-1] if hist else None
            view.P_0 = P[0].reshape(4, 4).copy()
            prev = view
            for k in range(1, n.value):
                sc = view.scanNumber - k
                if root is not None and sc == root.scanNumber and int(nodes[k]) == root._node:
                    prev._parent, prev._lazy_parent = root, None
                    return view._parent
                zz = self.__scanHistory__[sc - 1].measurements if sc >= 1 else None
                m = int(meas[k])
                a = DeviceTarget(self.__scanHistory__[sc - 1].time if sc >= 1 else view.time, sc, x[k].copy(),
                                 P[k].reshape(4, 4).copy(), ID=target_id, P_d=self.default_P_d, measurementNumber=m,
                                 measurement=(np.asarray(zz)[m - 1] if (zz is not None and m > 0) else None),
                                 cumulativeNLLR=float(cn[k]))
                a._tracker, a._node = self, int(nodes[k])
                prev._parent, prev._lazy_parent = a, None
                prev = a
            return view._parent
        return load

    def _leaf_snapshot(self):
        cap = self._cfg.max_nodes
        n = C.c_int32(0)
        x = np.zeros((cap, 4)); P = np.zeros((cap, 16), dtype=np.float32); cn = np.zeros(cap)
        meas = np.zeros(cap, dtype=np.int32); tgt = np.zeros(cap, dtype=np.int32); ids = np.zeros(cap, dtype=np.int32)
        node = np.zeros(cap, dtype=np.int32); fl = np.zeros(cap, dtype=np.uint8)
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        _lib.check(self._lib.mht_forest_leaves(self._ctx.handle, cap, p(x), p(P), p(cn), p(meas), p(tgt), p(ids), p(node),
                                               p(fl), C.byref(n)))
        k = n.value
        return dict(x=x[:k], P=P[:k].reshape(k, 4, 4), cnllr=cn[:k], meas=meas[:k], target=tgt[:k], ID=ids[:k],
                    node=node[:k], flags=fl[:k])

    def leafBatch(self):
        """All current leaves in target-list / DFS order (what the next scan will gate): dict of arrays."""
        return self._leaf_snapshot()

    def _leaf_views(self, root):
        snap = self._leaf_snapshot()
        scan = len(self.__scanHistory__)
        t = self.__scanHistory__[-1].time if scan else root.time
        out = []
        for i in np.where(snap["ID"] == root.ID)[0]:
            v = DeviceTarget(t, scan, snap["x"][i].copy(), snap["P"][i].copy(), ID=root.ID, P_d=self.default_P_d,
                             measurementNumber=int(snap["meas"][i]), cumulativeNLLR=float(snap["cnllr"][i]))
            v._tracker, v._node = self, int(snap["node"][i])
            v._lazy_parent = self._make_parent_loader(root.ID)
            out.append(v)
        return out

    # ------------------------------------------------------------------------------------------------
    def getTrackNodes(self):
        return self.__trackNodes__

    def getRuntimeAverage(self):
        return {k: np.mean(np.array(v)) for k, v in self.runtimeLog.items() if len(v)}

    def _findClustersFromSets(self):
        return self.__clusterList__

    def close(self):
        self._ctx.close()
