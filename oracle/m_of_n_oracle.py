"""TEST INFRASTRUCTURE (oracle): NumPy restatement of the reference's M-of-N track initiation, step 7 of every scan
(tracker.py:264-278 -> pymht/initiators/m_of_n.py:233-244).  It is bit-identical to the real reference initiator on the streams of
tests/golden/g8_initiator.npz (tests/test_initiator_golden.py) and is what the oracle tracker and the CPU baseline use; the product's
initiator runs on the device (pymht_amd/csrc/mht_init.hip, pymht_amd/initiators/m_of_n.py).  Arithmetic follows the reference's
dtypes (f32 states/covariances) so that scan traces -- which depend on when new targets are born -- stay comparable.

Pipeline per scan: (1) predict preliminary tracks, gate (chi2 0.99), global-nearest-neighbour
assignment, M/N bookkeeping -> confirmed tracks become new `Target`s; (2) leftover measurements are
paired with last scan's leftovers under v_max*dt (GNN again) -> new preliminary tracks; (3) what is
still left becomes next scan's initiators.
"""
import logging
import numpy as np
from scipy.optimize import linear_sum_assignment
from scipy.stats import chi2

from pymht_amd.models import pv
from pymht_amd.pyTarget import Target

log = logging.getLogger(__name__)

GATE_PROBABILITY = 0.99
GAMMA = chi2(df=2).ppf(GATE_PROBABILITY)
CONFIRMED, PRELIMINARY, DEAD = 1, 0, -1
# Covariance added in the duplicate-track similarity test: the reference uses its low-accuracy AIS
# measurement covariance there (models/ais.py:9-13, R(False) = 3.0^2 * I4, f32) even for radar tracks.
_SIM_R = (np.identity(4) * np.power(3.0, 2)).astype(np.float32)


def gnn_assign(delta, gate=np.inf):
    """Global nearest neighbour on a (tracks x measurements) distance matrix; entries above `gate`
    are forbidden.  Restates m_of_n.py:24-104: forbidden pairs get a big-M cost, all-forbidden rows /
    columns are dropped, the rest is padded square with 10*max and solved by the Hungarian method."""
    cost = np.array(delta, copy=True)
    cost[cost > gate] = np.inf
    ok = cost < np.inf
    if not ok.any():
        return []
    big_m = np.power(10., 1.0 + np.ceil(np.log10(1. + np.sum(cost[ok]))))
    cost[~ok] = big_m
    keep_c, keep_r = ok.any(axis=0), ok.any(axis=1)
    n_r, n_c = int(keep_r.sum()), int(keep_c.sum())
    n = max(n_r, n_c)
    sq = np.zeros((n, n)) + 10. * np.max(cost[ok])
    sq[:n_r, :n_c] = cost[np.ix_(keep_r, keep_c)]
    rr, cc = linear_sum_assignment(sq.astype(np.double))
    r_idx, c_idx = np.where(keep_r)[0], np.where(keep_c)[0]
    out = []
    for r, c in zip(rr, cc):
        if r < n_r and c < n_c and ok[r_idx[r], c_idx[c]]:
            out.append((r_idx[r], c_idx[c]))
    return out


class PreliminaryTrack:
    def __init__(self, state, covariance, mmsi=None):
        self.state, self.covariance = state, covariance
        self.n = self.m = 0
        self.predicted_state = None
        self.measurement_index = None
        self.mmsi = mmsi
        self.K = None

    def verdict(self, M, N):
        if self.m >= M:
            return CONFIRMED
        if self.n >= N:
            return DEAD
        return PRELIMINARY

    def similarity(self, other):
        d = self.state - other.state
        S = self.covariance + _SIM_R
        return d.T.dot(np.linalg.inv(S)).dot(d)


class _Seed:
    __slots__ = ("value", "timestamp")

    def __init__(self, value, timestamp):
        self.value, self.timestamp = value, timestamp


def merge_close_targets(cands, threshold):
    """m_of_n.py:133-154: greedily average candidates closer than `threshold` [m]."""
    out, used = [], set()
    for i, t in enumerate(cands):
        if i in used:
            continue
        d = np.array([np.linalg.norm(t.x_0[0:2] - o.x_0[0:2]) for o in cands])
        near = np.where(d < threshold)[0]
        pick = [cands[j] for j in near if j not in used]
        used.update(int(j) for j in near)
        if len(pick) == 1:
            out.append(pick[0])
        else:
            out.append(Target(pick[0].time, None, np.mean(np.array([q.x_0 for q in pick]), axis=0),
                              np.mean(np.array([q.P_0 for q in pick]), axis=0), measurement=pick[0].measurement))
    return out


class Initiator:
    def __init__(self, M, N, v_max, C, R, mergeThreshold=5, **kwargs):
        self.M, self.N, self.C, self.R = M, N, C, R
        self.v_max = v_max
        self.gamma = GAMMA
        self.merge_threshold = mergeThreshold
        self.initiators, self.preliminary_tracks = [], []
        self.last_timestamp = None

    def processMeasurements(self, radar_measurement_list, ais_measurement_list=()):
        """ais_measurement_list: the scan's AIS messages no track took (objects with time, state, mmsi): each starts a preliminary track
        unless one with its identity exists or an existing one is too similar (m_of_n.py:262-280)."""
        unused, born = self._advance_preliminary(radar_measurement_list, ais_measurement_list)
        unused = self._pair_with_seeds(unused, radar_measurement_list)
        z = radar_measurement_list.measurements
        self.initiators = [_Seed(z[i], radar_measurement_list.time) for i in unused]
        self.last_timestamp = radar_measurement_list.time
        return merge_close_targets(born, self.merge_threshold)

    # m_of_n.py:246-378
    def _advance_preliminary(self, mlist, ais=()):
        born, now = [], mlist.time
        z = np.array(mlist.measurements, dtype=np.float32)
        tracks = self.preliminary_tracks
        if self.last_timestamp is not None:
            dt = now - self.last_timestamp
            F, Qm = pv.Phi(dt), pv.Q(dt)
            for t in tracks:
                t.predicted_state = F.dot(t.state)
                t.covariance = F.dot(t.covariance).dot(F.T) + Qm
        else:
            assert not tracks
        have = {t.mmsi for t in tracks if t.mmsi is not None}             # m_of_n.py:262-264
        for m in ais:                                                      # m_of_n.py:265-280
            if m.mmsi in have:
                continue
            dT = now - m.time
            Phi = pv.Phi(dT)                                               # (models/ais.py:15-20 is the same matrix)
            state = Phi.dot(m.state)                                       # classDefinitions.py:470-475
            cov = Phi.dot(pv.P0).dot(Phi.T) + pv.Q(dT)
            cand = PreliminaryTrack(state, cov, m.mmsi)
            cand.predicted_state = state
            if not any(p.similarity(cand) <= 1.0 for p in tracks):
                tracks.append(cand)
        pred = np.array([np.copy(t.predicted_state) for t in tracks], ndmin=2, dtype=np.float32)
        for t in tracks:
            t.predicted_state = None
        n1, n2 = len(tracks), z.shape[0]
        if n1 == 0 or (len(ais) == 0 and (n2 == 0 or z.size == 0)):      # m_of_n.py:289-292
            return np.arange(n2).tolist(), born
        delta = np.ones((n1, n2), dtype=np.float32) * np.inf
        for i in range(n1):
            dz = z - self.C.dot(pred[i])
            dist = np.linalg.norm(dz, axis=1)
            P_bar = tracks[i].covariance
            S_inv = np.linalg.inv(self.C.dot(P_bar).dot(self.C.T) + self.R)
            tracks[i].K = P_bar.dot(self.C.T).dot(S_inv)
            inside = np.sum(np.matmul(dz, S_inv) * dz, axis=1) <= self.gamma
            delta[i, inside] = dist[inside]
        pairs = gnn_assign(delta)
        for ti, mi in pairs:
            t = tracks[ti]
            dz = z[mi] - self.C.dot(pred[ti])
            P_bar = t.covariance
            t.state = pred[ti] + t.K.dot(dz)
            t.covariance = P_bar - t.K.dot(self.C).dot(P_bar)
            t.m += 1
            t.measurement_index = mi
        hit = {p[0] for p in pairs}
        for ti, t in enumerate(tracks):
            if ti not in hit:
                t.state = pred[ti]
            t.n += 1
        drop = []
        for ti, t in enumerate(tracks):
            v = t.verdict(self.M, self.N)
            if np.linalg.norm(t.state[2:4]) > self.v_max * 1.5 or v == DEAD:
                drop.append(ti)
            elif v == CONFIRMED:
                born.append(Target(now, None, np.array(t.state), t.covariance,
                                   measurementNumber=t.measurement_index + 1, measurement=z[t.measurement_index]))
                drop.append(ti)
        for ti in reversed(drop):
            tracks.pop(ti)
        taken = {p[1] for p in pairs}
        return [i for i in range(n2) if i not in taken], born

    # m_of_n.py:380-413 and :425-478
    def _pair_with_seeds(self, unused, mlist):
        n1, n2 = len(self.initiators), len(unused)
        if n1 == 0 or n2 == 0:
            return unused
        now = mlist.time
        z = np.array(mlist.measurements, ndmin=2, dtype=np.float32)[unused]
        seeds = np.array([s.value for s in self.initiators], ndmin=2, dtype=np.float32)
        d = (z[None, :, :] - seeds[:, None, :]).astype(np.float64)      # float32 differences stored as float64, like the reference
        dist = np.linalg.norm(d, axis=2)
        pairs = gnn_assign(dist, self.v_max * (now - self.initiators[0].timestamp))
        taken = {unused[j] for _, j in pairs}
        left = sorted(i for i in unused if i not in taken)
        for si, mj in pairs:
            dt = now - self.initiators[si].timestamp
            vel = (z[mj] - self.initiators[si].value) / dt
            cand = PreliminaryTrack(np.hstack((z[mj], vel)), pv.P0)
            if not any(p.similarity(cand) <= 1.0 for p in self.preliminary_tracks):
                self.preliminary_tracks.append(cand)
        return left
