"""Host side of the AIS-aided path: the scan's AIS messages in the order the reference walks them, with the per-group
transition matrices the device needs (csrc/mht_ais_math.h: AisGroup, AisMsg).

Reference: Tracker.__fuseRadarAndAis, pymht/tracker.py:429 (`aisTimeSet = {m.time for m in aisMeasurements}`), :447 (`for aisTime in
aisTimeSet`), :452 (`for highAccuracy in [True, False]`), :460-461 (the group's messages in list order); models/pv.py:12-24 (Phi, Q
for the two time steps), models/ais.py:9-13 (measurement noise of the two accuracy classes)."""
import ctypes as C

import numpy as np


class AisMessage:
    """classDefinitions.py:428-434: time, state [x, y, vx, vy], mmsi (> 1e8), highAccuracy."""
    __slots__ = ("time", "state", "mmsi", "highAccuracy")

    def __init__(self, time, state, mmsi, highAccuracy=False):
        self.time, self.state, self.mmsi, self.highAccuracy = time, np.asarray(state, dtype=np.float64), int(mmsi), bool(highAccuracy)


class AisMessageList(list):
    """classDefinitions.py:597-617: of several messages of one ship only the latest is kept."""

    def __init__(self, *args):
        list.__init__(self, *args)
        latest = {}
        for i, m in enumerate(self):
            if m.mmsi not in latest or self[latest[m.mmsi]].time <= m.time:
                latest[m.mmsi] = i
        keep = sorted(latest.values())
        self[:] = [self[i] for i in keep]

    def filterUnused(self, usedMmsiSet):
        return [m for m in self if m.mmsi not in usedMmsiSet]


class MhtAisGroup(C.Structure):
    _fields_ = [("A1", C.c_float * 16), ("Q1", C.c_float * 16), ("A2", C.c_float * 16), ("Q2", C.c_float * 16),
                ("r_diag", C.c_float), ("first", C.c_int32), ("count", C.c_int32), ("pad", C.c_int32)]


class MhtAisMsg(C.Structure):
    _fields_ = [("state", C.c_double * 4), ("mmsi", C.c_int32), ("pad", C.c_int32)]


SIGMA_HIGH, SIGMA_LOW = 1.0, 3.0          # models/ais.py:6-7


def group_messages(ais_list, leaf_time, scan_time, model):
    """-> (groups: ctypes array of MhtAisGroup, msgs: ctypes array of MhtAisMsg, order: list of indices into ais_list).
    `order[i]` is the message behind device message i; the AIS measurement node of device message i is M + i."""
    times = {m.time for m in ais_list}            # (a SET, iterated in its own order: what the reference does)
    groups, order = [], []
    for t in times:
        dT1, dT2 = float(t) - leaf_time, scan_time - float(t)
        A1, Q1, A2, Q2 = model.Phi(dT1), model.Q(dT1), model.Phi(dT2), model.Q(dT2)
        for high in (True, False):
            idx = [i for i, m in enumerate(ais_list) if m.time == t and bool(m.highAccuracy) == high]
            if not idx:
                continue
            g = MhtAisGroup()
            for name, mat in (("A1", A1), ("Q1", Q1), ("A2", A2), ("Q2", Q2)):
                getattr(g, name)[:] = np.asarray(mat, dtype=np.float32).reshape(-1).tolist()
            g.r_diag = float(np.float32(np.power(SIGMA_HIGH if high else SIGMA_LOW, 2)))
            g.first, g.count = len(order), len(idx)
            groups.append(g)
            order.extend(idx)
    garr = (MhtAisGroup * max(len(groups), 1))(*groups)
    marr = (MhtAisMsg * max(len(order), 1))()
    for i, src in enumerate(order):
        m = ais_list[src]
        marr[i].state[:] = np.asarray(m.state, dtype=np.float64).tolist()
        marr[i].mmsi = int(m.mmsi)
    return garr, len(groups), marr, order


class MhtAisInitMsg(C.Structure):
    _fields_ = [("state", C.c_double * 4), ("dT", C.c_double), ("mmsi", C.c_int32), ("pad", C.c_int32)]


def initiator_messages(ais_list, scan_time):
    """The scan's messages for the initiator (m_of_n.py:265-280), in LIST order: ctypes array of MhtAisInitMsg."""
    arr = (MhtAisInitMsg * max(len(ais_list), 1))()
    for i, m in enumerate(ais_list):
        arr[i].state[:] = np.asarray(m.state, dtype=np.float64).tolist()
        arr[i].dT = float(scan_time) - float(m.time)
        arr[i].mmsi = int(m.mmsi)
    return arr
