#!/usr/bin/env python
"""bench.py -- scans/sec of the pyMHT per-scan hot path on MI355X (BASELINE.json metric).

A "step" = one radar scan through steps 1-6 of Tracker.addMeasurementList (grow/gate/score every leaf against
every measurement, cluster, per-cluster 0-1 ILP, track termination, N-scan pruning) on the device-resident
hypothesis forest -- TWO HIP launches since round 4: the grow launch (gate + update + score + child creation; its target workgroups hook
their targets into a device-wide union-find = the clustering; the target-side commit of the previous scan rides in it) and the ILP launch
(cluster tables derived per workgroup, the ILPs, the per-target termination / prune epilogue), the grow launch of scan k + 1 launched
any-order behind the ILP launch of scan k [+ commit and add_targets launches when tracks are born]; no memsets, no host round trip.
stage_ms therefore shows the commit inside "gate", no "cluster" launch ("cluster" = two back-to-back event records), and "prune" is
what two back-to-back event records cost.  Workload = BASELINE.json configs[2] (headline):
500 targets, ~500 measurements/scan, N-scan = 5, synthetic scans from pymht_amd/utils/scenario.py.

Protocol
  1. pre-pass (untimed): the full drop-in Tracker incl. the host-side M-of-N initiator runs the W+K scans once;
     the targets it gives birth to are recorded (step 7 of the reference is off the hot path).
  2. all scans and the recorded births are staged in HBM.
  3. replay: fresh forest, W untimed warm-up scans, then EXACTLY K timed scans between barrier+synchronize pairs;
     nothing is fetched from the device inside the timed region.  Afterwards the last report is compared with the
     pre-pass (same selections => the timed run did the same work).
  4. a second identical replay with HIP events around the stages gives per-stage device time -> `roofline`.
  5. rank 0, N=1 only: the CPU oracle (NumPy restatement of the reference, 1 thread) is timed on a bounded sample
     of the same scan stream -> `cpu_baseline`.
Multi-GPU (--gpus N under torch.distributed.run): every rank tracks its own independent sensor sector
(BASELINE config 4: disjoint sectors = no data-path collective), weak scaling; value = total scans / max time.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

LAMBDA_NU = 1e-4
ETA2 = 5.99
# HBM bytes per fgrow_kernel launch from the PMC counters (FETCH_SIZE + WRITE_SIZE, separate rocprofv3 --pmc passes on this
# workload, full-size scans): profiles/r04_pmc_hbm_traffic.txt.  Only quoted when the run's own counter passes are off or fail (--pmc off),
# and labelled as not measured by this run.  Keyed by config name; None = not profiled.
PMC_TRAFFIC_BYTES = {"cfg3": (1686 + 3681) * 1024}      # (round 4's steady-state launch; the transient scans of the first N+2 are larger)
HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6290 GB/s is the measured achievable copy rate


def p(a):
    return a.ctypes.data_as(C.c_void_p)


def pmc_traffic_live(config, timeout_s=180.0):
    """HBM bytes per steady-state fgrow_kernel launch from the PMC counters, measured NOW: two rocprofv3 passes (FETCH_SIZE, then
    WRITE_SIZE -- they do not fit one pass, and --pmc is not combined with any tracing) over a short replay of this same command
    on this GPU, as MI355X_MICROARCH.md prescribes.  Counter values are KiB per dispatch (calibrated on this kernel's access
    pattern in round 1: profiles/r01_pmc_hbm_traffic.txt); the figure is the mean over the last quarter of the launches (the
    replay's steady-state scans).  Returns (bytes or None, note)."""
    import glob
    import shutil
    import sqlite3
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    if any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ):
        return None, "this run is itself being profiled (no nested counter passes)"
    tot, t0 = 0.0, time.time()
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="mht_pmc_", dir="/tmp")
        try:
            cmd = [exe, "--pmc", counter, "-d", d, "-o", "p", "--", sys.executable, os.path.abspath(__file__), "--config", config,
                   "--cpu-scans", "0", "--sectors", "0", "--steps", "40", "--warmup", "8", "--pmc", "off", "--extras", "off"]
            left = timeout_s - (time.time() - t0)
            if left < 20:
                return None, "time budget of the counter passes used up"
            r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                               timeout=left)
            dbs = glob.glob(os.path.join(d, "**", "*_results.db"), recursive=True)
            if r.returncode != 0 or not dbs:
                return None, "rocprofv3 --pmc %s failed (rc %d)" % (counter, r.returncode)
            c = sqlite3.connect(dbs[0])
            tabs = [q[0] for q in c.execute("select name from sqlite_master where type='table'")]
            t = lambda pre: next(x for x in tabs if x.startswith(pre))
            pmc, disp, sym = t("rocpd_pmc_event"), t("rocpd_kernel_dispatch"), t("rocpd_info_kernel_symbol")
            vals = [v for (v,) in c.execute(
                'select p.value from "%s" d join "%s" p on p.event_id = d.event_id join "%s" s on s.id = d.kernel_id '
                "where s.kernel_name like '%%fgrow_kernel%%' order by d.start" % (disp, pmc, sym))]
            c.close()
            if len(vals) < 16:
                return None, "only %d fgrow_kernel launches in the %s pass" % (len(vals), counter)
            tail = vals[-(len(vals) // 4):]
            tot += 1024.0 * sum(tail) / len(tail)
        except Exception as e:      # (a bench line without the counters is still a bench line)
            return None, "counter pass failed: %s" % repr(e)[:120]
        finally:
            shutil.rmtree(d, ignore_errors=True)
    return tot, "this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in two separate passes over a 64-scan replay of the same stream on this GPU, " \
                "KiB per fgrow_kernel dispatch, mean of the last quarter of the launches (steady state).  Under counter collection " \
                "(ROCPROF_COUNTER_COLLECTION=1) rocprofv3 runs one kernel at a time, so the library keeps the initiator on the ctx stream and " \
                "every launch wait is satisfied by stream order: the counted grow launch does the same loads and stores as the timed one, its " \
                "waits for the previous scan's ILP launch return at once (the counters profile the event-synchronised variant of the path)"


def model_of(sc):
    """The tracking model of a config: the reference's 4-state CV model (models/pv.py), or -- BASELINE config 5 names a six-state
    constant-turn model, the reference ships none -- pymht_amd/models/ct.py in the six-state build of the library (a forest made with
    MHT_FOREST_CT: every hypothesis its own Phi(T, w) and covariance chain).  MHT_BENCH_CFG5_MODEL=ca: the linear constant-acceleration
    stand-in of rounds 3-4 (covariances shared by value); =pv: the 4-state model at config 5's size."""
    from pymht_amd.models import pv, ca, ct
    if sc.get("name") != "cfg5":
        return pv
    return {"ct": ct, "ca": ca, "pv": pv}[os.environ.get("MHT_BENCH_CFG5_MODEL", "ct")]


def roots_of(sc, model):
    nx = model.C_RADAR.shape[1]
    return sc["x0"] if nx == 4 else np.concatenate([sc["x0"], np.zeros((len(sc["x0"]), nx - 4))], axis=1)


def make_tracker(sc, device, **kw):
    from pymht_amd.tracker import Tracker
    from pymht_amd.pyTarget import Target
    model = model_of(sc)
    big = sc.get("name") == "cfg5"
    if model.C_RADAR.shape[1] != 4:
        kw["useInitiator"] = False      # (the M-of-N initiator is the reference's 4-state one)
    trk = Tracker(model, sc["period"], sc["lambda_phi"], LAMBDA_NU, P_d=sc["P_d"], N=sc["N"], eta2=ETA2, device=device,
                  maxTargets=kw.pop("maxTargets", int(os.environ.get("MHT_BENCH_MAXT", "2304" if big else "2048"))),
                  maxNodes=kw.pop("maxNodes", int(os.environ.get("MHT_BENCH_MAXN", str(1 << (20 if big else 19))))),
                  maxMeasurements=int(os.environ.get("MHT_BENCH_MAXM", "2048" if big else "1024")), **kw)
    trk._add_targets([Target(sc["t0"], None, x.copy(), model.P0, status="preinitialized") for x in roots_of(sc, model)])
    return trk


def prepass(sc, device, n_untimed=0):
    """Full tracker through the drop-in API (device initiator); records births per scan and per-scan stats.  The API rate is taken
    over the scans behind the first `n_untimed` (pre-roll + warm-up, like the timed replay)."""
    from pymht_amd.utils.classDefinitions import MeasurementList
    trk = make_tracker(sc, device, deviceTiming=False, logScanStats=True)
    births, stats = [], []
    orig = trk._apply_births

    class Born:          # what Replay needs of a new target
        def __init__(self, x0, P0, meas):
            self.x_0, self.P_0, self.measurementNumber = x0, P0, meas

    births = [[] for _ in sc["scans"]]

    def recording(b, scanTime, scanNumber, z_unused):
        orig(b, scanTime, scanNumber, z_unused)
        for r in b[b["id"] >= 0]:
            births[scanNumber - 1].append(Born(r["x0"].astype(np.float32), r["P0"].reshape(trk.nx, trk.nx).copy(), int(r["meas"])))

    trk._apply_births = recording
    # streaming use of the drop-in API: scans go in one after the other, results are looked at after the last one (every look at
    # the tracker's state waits for the scan in flight; a host that reads after every scan serialises itself with the device)
    lists = [MeasurementList(float(t), z) for z, t in zip(sc["scans"], sc["times"])]
    for m in lists[:n_untimed]:
        trk.addMeasurementList(m)
    trk.synchronize()
    t0 = time.time()
    for m in lists[n_untimed:]:
        trk.addMeasurementList(m)
    trk.synchronize()
    api_s = (time.time() - t0) / max(1, len(lists) - n_untimed)      # seconds per scan
    stats = [(s["L"], s["G"], s["M"], s["ilp"], s["branched"], s["blp_iters_max"], s["nTargets"]) for s in trk.scanStatsLog]
    live = trk._sel[0]      # selected leaf of every target that survived the last scan (tracks born by that scan's initiator excluded)
    final = [(int(i), int(m)) for i, m in zip(live["id"], live["sel_meas"])]
    init_s = float(np.sum(trk.runtimeLog["Init"]))
    trk.close()
    return births, np.array(stats), final, api_s, init_s


_GROUP_STREAMS = []


def pick_streams(n, prios, device):
    """The HIP streams the groups of sectors run on: made ONCE per process, four of them up front, and every run takes the first n.  The
    runtime maps streams onto a few hardware queues as they are first used and later streams SHARE the queues of earlier ones: a second
    multi-sector run on streams of its own could land both of its groups on one queue -- its launch chains then run one after the other
    (measured: 16 sectors in two groups 49 k instead of 80 k scans/s whenever the four-sector run had taken four streams before it,
    profiles/r05_merge_ab.txt).  The first streams of a process sit on queues of their own."""
    want = max(n, 4)
    base = [0, -1, 1, 0]
    while len(_GROUP_STREAMS) < want:
        q = len(_GROUP_STREAMS)
        _GROUP_STREAMS.append(torch.cuda.Stream(device=device, priority=(prios[q % len(prios)] if len(prios) > 2 or q >= 4 else base[q % 4])))
    return _GROUP_STREAMS[:n]


class Replay:
    """Drives the forest through the raw C ABI with every input resident in HBM."""

    def __init__(self, sc, births, device):
        from pymht_amd import _lib
        self._lib_mod = _lib
        self.sc = sc
        if os.environ.get("MHT_BENCH_STREAM") == "1" and torch.cuda.current_stream(device).cuda_stream == 0:      # experiment: a stream of its own instead of the null stream
            self._own_stream = torch.cuda.Stream(device=device)
            with torch.cuda.stream(self._own_stream):
                self.trk = make_tracker(sc, device, useInitiator=False, deviceTiming=False)
        else:
            self.trk = make_tracker(sc, device, useInitiator=False, deviceTiming=False)
        self.lib, self.h = self.trk._lib, self.trk._ctx.handle
        dev = self.trk._ctx.device
        self.M = [int(z.shape[0]) for z in sc["scans"]]
        zall = np.concatenate([z.reshape(-1, 2) for z in sc["scans"]], axis=0).astype(np.float32)
        self.z = torch.from_numpy(zall).to(dev)
        self.zoff = np.concatenate([[0], np.cumsum(self.M)]).astype(np.int64)
        flat = [b for per in births for b in per]
        self.nb = [len(per) for per in births]
        self.boff = np.concatenate([[0], np.cumsum(self.nb)]).astype(np.int64)
        n = max(len(flat), 1)
        nx = self.nx = self.trk.nx
        x0 = np.zeros((n, nx)); P0 = np.zeros((n, nx * nx), np.float32); fl = np.zeros(n, np.uint8); me = np.zeros(n, np.int32)
        for i, b in enumerate(flat):
            x0[i] = np.asarray(b.x_0, dtype=np.float64)
            P0[i] = np.asarray(b.P_0, dtype=np.float32).reshape(nx * nx)
            fl[i] = 3 if np.asarray(b.x_0).dtype == np.float32 else 0
            me[i] = 0 if b.measurementNumber is None else int(b.measurementNumber)
        self.bx, self.bP = torch.from_numpy(x0).to(dev), torch.from_numpy(P0).to(dev)
        self.bf, self.bm = torch.from_numpy(fl).to(dev), torch.from_numpy(me).to(dev)
        self.bpd = torch.full((n,), float(sc["P_d"]), dtype=torch.float64, device=dev)
        self.k = 0
        torch.cuda.synchronize()

    def step(self):
        k, lib, h = self.k, self.lib, self.h
        rc = lib.mht_forest_step(h, self.z.data_ptr() + int(self.zoff[k]) * 8, self.M[k])
        if rc:
            self._lib_mod.check(rc)
        self.k += 1
        self._births(k)

    def births_after_step(self):
        """group replay: the scan itself went out with the group; this sector's recorded births of that scan follow"""
        k = self.k
        self.k += 1
        self._births(k)

    def _births(self, k):
        lib, h = self.lib, self.h
        nb = self.nb[k]
        if nb:
            o = int(self.boff[k])
            rc = lib.mht_forest_add_targets_dev(h, nb, self.bx.data_ptr() + o * 8 * self.nx, self.bP.data_ptr() + o * 4 * self.nx * self.nx,
                                                self.bf.data_ptr() + o, self.bpd.data_ptr() + o * 8,
                                                self.bm.data_ptr() + o * 4, 1, None, None)
            if rc:
                self._lib_mod.check(rc)

    def report(self):
        rep = self._lib_mod.MhtScanReport()
        self._lib_mod.check(self.lib.mht_forest_report(self.h, C.byref(rep)))
        _REPORT_DTYPE = self.trk._REPORT_DTYPE
        nT = rep.n_targets
        recs = np.ctypeslib.as_array(C.cast(rep.targets, C.POINTER(C.c_uint8)), shape=(nT * _REPORT_DTYPE.itemsize,)) \
            .view(_REPORT_DTYPE).copy()
        return rep, recs

    def close(self):
        self.trk.close()


def ilp_accounting(sc, births, device, n_warm, n_scans):
    """SURVEY.md 8(d) "ILP stage accounting": instances, sum of nHyp (columns) and of nnz(A1) (measurement rows on the columns' paths)
    over the multi-target clusters, how they were solved, the slowest cluster -- read from the forest's tables after every scan of a
    short separate replay (untimed: every read synchronises)."""
    rp = Replay(sc, births, device)
    trk = rp.trk

    def rd(name, k, dt=np.int32):
        a = np.zeros(max(int(k), 1), dtype=dt)
        rp._lib_mod.check(rp.lib.mht_forest_debug_read(rp.h, name.encode(), a.ctypes.data_as(C.c_void_p), a.nbytes))
        return a
    for _ in range(n_warm):
        rp.step()
    inst = n_hyp = nnz = cols_sampled = certified = branched = limit = 0
    iters_max, us_max, us_sum = 0, 0.0, 0.0
    pds = 8 if int(sc["N"]) + 1 <= 8 else 16      # (ints per path record: mht_forest_create)
    for _ in range(n_scans):
        rp.step()
        rp._lib_mod.check(rp.lib.mht_synchronize(rp.h))
        cnt = rd("cl_counts", 8)
        nC, nM = int(cnt[0]), int(cnt[1])
        if nM == 0:
            continue
        ptr, ml, it, st = rd("cl_ptr", nC + 1), rd("multi_list", nM), rd("cl_iters", nC), rd("cl_status", nC)
        tm = rd("cl_time", 8 * nC).reshape(-1, 8)
        nT0 = int(ptr[nC])
        tch, tce, mem = rd("tchild", nT0 + 1), rd("tcend", nT0 + 1), rd("cl_members", nT0)
        for c in ml:
            members = mem[ptr[c]:ptr[c + 1]]
            cols = int(sum(int(tce[m]) - int(tch[m]) for m in members))
            inst += 1
            n_hyp += cols
            certified += int(st[c] == 1); branched += int(st[c] == 2); limit += int(st[c] == 3)
            iters_max = max(iters_max, int(it[c]))
            us = tm[c, 1] / 100.0
            us_max = max(us_max, us); us_sum += us
        # nnz(A1) of this scan's ILPs: entries >= 0 of the path records of their columns (one record of pds ints per column)
        sample = [m for c in ml for m in mem[ptr[c]:ptr[c + 1]]]
        for m in sample[:64]:      # (a sample of the member targets: every read is a device round trip)
            b, e = int(tch[m]), int(tce[m])
            if e > b:
                rec = rd("path@%d" % (4 * pds * b), pds * (e - b)).reshape(-1, pds)
                nnz += int((rec >= 0).sum())
        cols_sampled += sum(int(tce[m]) - int(tch[m]) for m in sample[:64])
    rp.close()
    cs = max(1, cols_sampled)
    return {"scans": n_scans, "instances": inst, "sum_nHyp": n_hyp, "sum_nnz_A1_estimate": int(round(n_hyp * nnz / cs)),
            "nnz_per_column_sampled": nnz / cs, "certified_by_dual": certified, "branch_and_bound_or_exact_search": branched, "node_limit": limit,
            "dual_iters_max": iters_max, "cluster_us_mean": us_sum / max(1, inst), "cluster_us_max": us_max,
            "note": "multi-target clusters of %d scans behind the warm-up (separate untimed replay); status 2 = solved exactly by the pair / signature search "
                    "or the GPU branch and bound after the dual rounds did not certify; nnz(A1) from the path records of a sample of the columns" % n_scans}


def cpu_baseline(sc, n_warm, n_timed, budget_s=None):
    """The oracle (NumPy restatement of the reference algorithm, validated bitwise against the reference) on the
    host: stages Process+Cluster+Optim+Terminate+N-Prune of `n_timed` scans after `n_warm` warm-up scans."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import mht_oracle as orc
    from m_of_n_oracle import Initiator
    from pymht_amd.utils.classDefinitions import MeasurementList
    from pymht_amd.models import pv
    model = model_of(sc)
    four = model.C_RADAR.shape[1] == 4

    class Adapter:
        def __init__(self):
            self.i = Initiator(2, 3, 20, pv.C_RADAR, pv.R_RADAR(), 4 * 2.5 ** 2)

        def processMeasurements(self, time_, z):
            return [(t.x_0, t.P_0, t.measurementNumber, t.measurement)
                    for t in self.i.processMeasurements(MeasurementList(time_, z))]

    o = orc.OracleTracker(sc["period"], sc["lambda_phi"], LAMBDA_NU, P_d=sc["P_d"], N=sc["N"], eta2=ETA2, initiator=Adapter() if four else None,
                          model=None if four else model)
    for x in roots_of(sc, model):
        o.initiate_target(sc["t0"], x.copy(), orc.model_P0() if four else model.P0.copy(), status="preinitialized")
    per = []
    t_b = time.time()
    for k in range(n_warm + n_timed):
        if budget_s is not None and k >= 2 and time.time() - t_b > budget_s:      # (bounded: the sample is the last scans that were reached)
            break
        info = o.add_scan(float(sc["times"][k]), sc["scans"][k])
        per.append((sum(o.toc[s] for s in ("Process", "Cluster", "Optim", "Terminate", "N-Prune")), info["L"], info["G"], info["M"], o.toc["Process"], o.toc["Cluster"] + o.toc["Optim"]))
    n_done = len(per)
    n_timed = max(1, min(n_timed, n_done - min(n_warm, n_done - 1)))
    st = np.array(per[n_done - n_timed:])
    hot = float(st[:, 0].sum())
    return dict(value=n_timed / hot, unit="scans/s", cores=1, kind="port",
                sample="oracle/mht_oracle.py (NumPy restatement, bit-identical to the reference in the dev container), "
                       "1 thread, scans %d..%d of the same scan stream after %d warm-up scans%s; stages "
                       "Process+Cluster+Optim+Terminate+N-Prune; mean L=%d G=%d M=%d; gate %.0f ms, cluster+ILP %.0f ms per scan"
                       % (n_done - n_timed, n_done - 1, n_done - n_timed, "" if budget_s is None else " (bounded to %.0f s of wall clock)" % budget_s,
                          st[:, 1].mean(), st[:, 2].mean(), st[:, 3].mean(), 1e3 * st[:, 4].mean(), 1e3 * st[:, 5].mean()))


def steady_windows(config, seed, centre, device, warm, n_win=3, win=20, long_k=400):
    """Extras that make 2 % steps measurable (the headline's timed window is K = 20 scans, ~1 ms): the same stream replayed in a run of
    its own -- `n_win` back-to-back windows of `win` scans, then one window of `long_k` scans, each bracketed by a device synchronise --
    and checked against its own pre-pass at the end.  Returns (value_steady, windows dict)."""
    from pymht_amd.utils.scenario import make_config
    n_total = warm + n_win * win + long_k
    sc2 = make_config(config, seed=seed, n_scans=n_total, centre=centre, confine=True)
    births2, _, final2, _, _ = prepass(sc2, device, warm)
    rp = Replay(sc2, births2, device)
    for _ in range(warm):
        rp.step()
    torch.cuda.synchronize()
    rates = []
    for _ in range(n_win):
        t0 = time.perf_counter()
        for _ in range(win):
            rp.step()
        torch.cuda.synchronize()
        rates.append(win / (time.perf_counter() - t0))
    t0 = time.perf_counter()
    for _ in range(long_k):
        rp.step()
    torch.cuda.synchronize()
    steady = long_k / (time.perf_counter() - t0)
    rep, recs = rp.report()
    ok = [(int(r["id"]), int(r["sel_meas"])) for r in recs if int(r["status"]) == 0] == final2 and rep.error == 0
    rp.close()
    return steady, {"windows": n_win, "steps_per_window": win, "min": float(min(rates)), "median": float(np.median(rates)), "all": [float(v) for v in rates],
                    "steady_steps": long_k, "replay_matches_prepass": bool(ok),
                    "note": "a replay of its own of the same stream (same seed, longer): %d windows of %d scans back to back, then %d scans = value_steady; scans/s" % (n_win, win, long_k)}


def config_extra(name, device, warm, steps, cpu_warm, cpu_timed, cpu_budget_s):
    """A bounded line for another BASELINE config (cfg2: 50 targets / 200 measurements / N = 3; cfg5: constant-turn six-state model, 2 000
    targets x ~2 000 measurements, N = 6) so that the driver's record holds them: pre-pass through the drop-in API, timed raw replay of
    `steps` scans behind `warm`, the same end-state check as the headline, the grow stage's roofline from HIP-event stage times, and the
    CPU restatement on a few steady-state scans of the same stream (bounded by `cpu_budget_s` of wall clock)."""
    from pymht_amd.utils.scenario import make_config
    sc = make_config(name, seed=5446, n_scans=warm + steps, confine=True)
    nx = model_of(sc).C_RADAR.shape[1]
    births, stats, final, api_s, _ = prepass(sc, device, warm)
    rp = Replay(sc, births, device)
    for _ in range(warm):
        rp.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        rp.step()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    rep, recs = rp.report()
    ok = [(int(r["id"]), int(r["sel_meas"])) for r in recs if int(r["status"]) == 0] == final and rep.error == 0
    buf = (C.c_float * 5)()
    n = C.c_int32(0)
    nt = min(16, steps)
    rp.close()
    # (per-stage times need event-synchronised launches: a second replay of the first timed scans)
    rp = Replay(sc, births, device)
    for _ in range(warm):
        rp.step()
    rp._lib_mod.check(rp.lib.mht_forest_set_timing(rp.h, 1))
    for _ in range(nt):
        rp.step()
    rp._lib_mod.check(rp.lib.mht_forest_stage_times(rp.h, C.byref(buf), C.byref(n)))
    ms = np.array(list(buf)) / nt
    rp.close()
    tm = stats[warm:warm + steps]
    Lm, Gm, Mm = float(tm[:, 0].mean()), float(tm[:, 1].mean()), float(tm[:, 2].mean())
    per_leaf = (8 * nx + 4 * nx * nx + 16) + (8 * nx + 4 * nx * nx + 8 + 4 * nx * nx)
    per_pair = 8 * nx + 16
    b_gate = per_leaf * Lm + per_pair * Gm + 8.0 * Mm
    gbs = b_gate / (ms[0] * 1e-3) / 1e9 if ms[0] > 0 else 0.0
    out = {"name": name, "workload": "%d targets, ~%d measurements per scan, N-scan=%d, %d-state model %s" % (len(sc["x0"]), int(Mm), int(sc["N"]), nx, model_of(sc).__name__.split(".")[-1]),
           "value": steps / el, "unit": "scans/s", "steps": steps, "warmup": warm, "ms_per_step": 1e3 * el / steps, "replay_matches_prepass": bool(ok),
           "leaves_per_scan": Lm, "gated_pairs_per_scan": Gm, "meas_per_scan": Mm, "ilps_per_scan": float(tm[:, 3].mean()), "api_scans_per_sec": 1.0 / api_s,
           "stage_ms": {"gate": float(ms[0]), "cluster": float(ms[1]), "ilp": float(ms[2]), "prune": float(ms[3])},
           "roofline": {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "algorithmic_bytes": b_gate, "traffic": None,
                        "kernel": "grow stage (fgrow_ct_kernel + forest_ct_kernel for the constant-turn model, fgrow_kernel else), HIP events on the ctx stream, mean of %d scans" % nt}}
    if cpu_timed > 0:
        t_c = time.time()
        try:
            cb = cpu_baseline(sc, cpu_warm, cpu_timed, budget_s=cpu_budget_s)
        except Exception as e:      # noqa: BLE001
            cb = {"error": repr(e)[:200]}
        cb["wall_s"] = round(time.time() - t_c, 1)
        out["cpu_baseline"] = cb
    return out


T_START = time.time()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=40)
    ap.add_argument("--config", default="cfg3")
    ap.add_argument("--sectors", default="4,16,64", help="concurrent independent sectors per GPU for the multi_sector figures, comma separated (0 disables)")
    ap.add_argument("--cpu-scans", type=int, default=16, help="timed oracle scans for cpu_baseline (0 disables)")
    ap.add_argument("--cpu-warm", type=int, default=8)
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak (default): every rank tracks its own sector (BASELINE config 4); strong: ALL ranks track the same sector, "
                         "its independent clusters' ILPs spread over the ranks (one all-reduce of the selections per scan)")
    ap.add_argument("--extras", default="auto", help="auto: on one GPU with the headline config the line also carries value_steady / value_windows (a longer replay of the same "
                    "stream) and bounded lines for BASELINE configs 2 and 5 (`configs`); off: none of them")
    ap.add_argument("--pmc", choices=["auto", "off"], default="auto",
                    help="auto: roofline.traffic is measured by two rocprofv3 counter passes over a short replay (rank 0, one GPU; ~40 s); "
                         "off: the figure of profiles/ is quoted and labelled as not measured by this run")
    args = ap.parse_args()

    os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    torch.cuda.set_device(local)
    dist = None
    if world > 1 or os.environ.get("MHT_BENCH_FORCE_DIST"):      # (the env switch exercises the RCCL calls on a 1-GPU box)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(29400 + os.getpid() % 500))      # (MHT_BENCH_FORCE_DIST on a one-GPU box: no launcher has set the rendezvous)
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        # RCCL prints a version banner on STDOUT when its first communicator comes up; stdout carries the one JSON line,
        # so the banner is sent to stderr (fd level: it is written by the C library)
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group("nccl", rank=rank, world_size=world)
            dist.barrier()
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved, 1)
            os.close(saved)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    from pymht_amd.utils.scenario import make_config
    # PRE scans are always run (untimed) in front of the --warmup scans: the hypothesis trees need N+2 scans to reach their steady
    # state ("first N+2 discarded"; L is flat from scan ~6 on), so --steps 20 --warmup 5 measures the same regime as
    # the default --steps 400 --warmup 40
    PRE = 16
    W, K = PRE + args.warmup, args.steps
    # every rank = its own sensor sector (own targets, own clutter): BASELINE config 4
    from pymht_amd import parallel
    strong = args.scaling == "strong"
    srank = 0 if strong else rank      # (strong scaling: every rank is fed the same sector)
    sc = make_config(args.config, seed=parallel.sector_seed(5446, srank), n_scans=W + K, centre=parallel.sector_centre(srank), confine=True)
    nx = model_of(sc).C_RADAR.shape[1]
    births, stats, final, api_s, init_s = prepass(sc, local, W)

    # ---- timed replay ---------------------------------------------------------------------------------------------
    rp = Replay(sc, births, local)
    if strong:          # one tracker over all ranks: cluster c's ILP on rank c % world, selections all-reduced (MAX) every scan
        import ctypes as C_
        nw = C_.c_int32(0)      # (the exchange block: selections + the ranks' files for a giant component searched by teams across the ranks)
        rp._lib_mod.check(rp.lib.mht_forest_sharded_words(rp.h, world, C_.byref(nw)))
        sel_rel = torch.full((nw.value,), -1, dtype=torch.int32, device=rp.trk._ctx.device)

        def one_scan():
            k = rp.k
            rp._lib_mod.check(rp.lib.mht_forest_step_sharded_begin2(rp.h, rp.z.data_ptr() + int(rp.zoff[k]) * 8, rp.M[k], world, rank,
                                                                    sel_rel.data_ptr(), nw.value))
            parallel.merge_selections(sel_rel, dist)
            rp._lib_mod.check(rp.lib.mht_forest_step_sharded_end(rp.h, sel_rel.data_ptr()))
            rp.births_after_step()
    else:
        one_scan = rp.step
    for _ in range(W):
        one_scan()
    barrier()
    t0 = time.perf_counter()
    for _ in range(K):
        one_scan()
    t_enq = time.perf_counter()      # (the host has queued the K scans: how far ahead of the device it runs)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    barrier()
    elapsed = t1 - t0
    rep, recs = rp.report()
    got = [(int(r["id"]), int(r["sel_meas"])) for r in recs if int(r["status"]) == 0]
    same_work = (got == final) and rep.error == 0
    uf_ovl = np.zeros(2, dtype=np.int32)      # scans clustered inside the grow launch, grow launches that overlapped the previous scan's ILP launch
    rp.lib.mht_forest_debug_read(rp.h, b"uf_ovl", uf_ovl.ctypes.data_as(C.c_void_p), 8)
    rp.close()
    elapsed, same_work = parallel.reduce_clock(elapsed, same_work, dist, device="cuda")
    # one picture of all sectors (outside the timed region; KB-sized all-gather over RCCL)
    alive = [r for r in recs if int(r["status"]) == 0]
    picture = parallel.gather_tracks([int(r["id"]) for r in alive], np.array([r["sel_x"] for r in alive]).reshape(-1, nx)[:, :4],
                                     dist, device="cuda")

    # ---- stage times with HIP events on the launch stream (identical replay) ----------------------------------------
    rp = Replay(sc, births, local)
    for _ in range(W):
        rp.step()
    torch.cuda.synchronize()
    rp._lib_mod.check(rp.lib.mht_forest_set_timing(rp.h, 1))
    ms = np.zeros(5)
    done = 0
    buf = (C.c_float * 5)()
    n = C.c_int32(0)
    while done < K:
        chunk = min(32, K - done)
        for _ in range(chunk):
            rp.step()
        rp._lib_mod.check(rp.lib.mht_forest_stage_times(rp.h, C.byref(buf), C.byref(n)))
        ms += np.array(list(buf))
        done += chunk
    ms /= K
    rp.close()

    # ---- several independent sectors per GPU (BASELINE config 4 on one device): ONE batched launch set per scan for all of them
    #      (mht_group_step: blockIdx.y = sector), every sector its own forest, scan stream and births ------------------------------
    per_leaf = (8 * nx + 4 * nx * nx + 16) + (8 * nx + 4 * nx * nx + 8 + 4 * nx * nx)
    per_pair = 8 * nx + 16

    def run_multi(S):
        from pymht_amd.sectors import SectorGroup
        # (its own length, whatever --steps says: the driver's 20 timed scans are 0.6 ms of a batched replay -- start-up dominated)
        Km = 200 if S <= 4 else 100
        NTAIL = 8      # untimed scans behind the timed ones, read one by one: the mean of their grow stamps feeds the batched roofline
        if K >= Km + NTAIL:
            scs, brs, sts, fins = [sc], [births], [stats], [None]      # (sector 0 = the main sector, replayed to W + K scans there: its final state is of another scan)
        else:
            scs, brs, sts, fins = [], [], [], []
        for q in range(len(scs), S):
            sq = make_config(args.config, seed=parallel.sector_seed(5446, rank) + 17 * q, n_scans=W + Km + NTAIL,
                             centre=(parallel.sector_centre(rank)[0], 20000.0 * q), confine=True)
            bq, stq, fq, _, _ = prepass(sq, local)
            scs.append(sq)
            brs.append(bq)
            sts.append(stq)
            fins.append(fq)
        # the sectors form NG groups, each on its own HIP stream: one batched launch set per group and scan; two groups' chains of
        # dependent kernels interleave on the device (while one group's ILP kernel holds a workgroup per CU, the other's grow runs)
        # (up to four sectors: a group per sector -- four launch chains side by side, 45.6 k scans/s against 41.8 k in two groups of two;
        # sixteen sectors: two groups of eight, 75 k against 58 k in four groups and 53 k in eight -- profiles/r05_merge_ab.txt, "groups")
        # (thirty-two sectors and more: four groups -- 32: 117.8 k against 104.2 k in two groups, 64: 134.6 k against 119.7 k in two and 101.8 k in eight;
        # 128 sectors: 130 k in eight groups, 102 k in sixteen -- profiles/r06_experiments.txt, "sectors per GPU")
        NG = max(1, min(int(os.environ.get("MHT_BENCH_GROUPS", "4" if (S <= 4 or S >= 32) else "2")), S))
        # (streams of DIFFERENT priority: the runtime maps streams to a few hardware queues, and two streams of equal priority may land on
        # the same one -- the groups then run one after the other instead of side by side: 43 k instead of 74 k scans/s at 16 sectors,
        # decided by chance per process)
        prios = [int(v) for v in os.environ.get("MHT_BENCH_PRIOS", "0,-1,1,0" if NG > 2 else "0,-1").split(",")]
        streams = pick_streams(NG, prios, local)
        rps = []
        for q in range(S):
            with torch.cuda.stream(streams[q % NG]):
                rps.append(Replay(scs[q], brs[q], local))
        solo = os.environ.get("MHT_BENCH_SOLO") == "1"      # development: every sector stepped on its own (needs MHT_BENCH_GROUPS = sectors)
        grps = [] if solo else [SectorGroup([r.trk for r in rps[gi::NG]]) for gi in range(NG)]

        def group_step():
            if solo:
                for r in rps:
                    r.step()
                return
            k = rps[0].k
            for gi in range(NG):
                mem = rps[gi::NG]
                grps[gi].step_dev([r.z.data_ptr() + int(r.zoff[k]) * 8 for r in mem], [r.M[k] for r in mem])
            for r in rps:
                r.births_after_step()
        for _ in range(W):
            group_step()
        barrier()
        tm0 = time.perf_counter()
        for _ in range(Km):
            group_step()
        torch.cuda.synchronize()
        tm1 = time.perf_counter()
        barrier()
        okm = True
        t_grow = [[] for _ in rps]
        t_union = []

        def stamps(r):      # DevStatus::t of the scan just stepped: [0] grow start, [1] cluster start (absolute device wall clock, 10 ns ticks)
            a = np.zeros(2 * 8, dtype=np.uint64)
            r._lib_mod.check(r.lib.mht_forest_debug_read(r.h, b"status2", a.ctypes.data_as(C.c_void_p), a.nbytes))
            w = a.reshape(2, 8)[:, 2:].astype(np.int64)[r.k & 1]
            return int(w[0]), int(w[1])
        for tail in range(NTAIL):      # (untimed: every report read synchronises) device stamps of these scans' launches: grow start -> cluster start of the sector's group
            group_step()
            for q, r in enumerate(rps):
                repm, recm = r.report()
                okm = okm and repm.error == 0
                t_grow[q].append(repm.t_process * 1e-8)
            if not solo:      # the UNION of the groups' grow intervals of this scan (the groups run side by side on their own streams: their launches overlap)
                iv = sorted(stamps(rps[gi]) for gi in range(NG))
                tot, cur0, cur1 = 0, iv[0][0], iv[0][1]
                for a0, a1 in iv[1:]:
                    if a0 <= cur1:
                        cur1 = max(cur1, a1)
                    else:
                        tot += cur1 - cur0
                        cur0, cur1 = a0, a1
                t_union.append((tot + cur1 - cur0) * 1e-8)
        for q, r in enumerate(rps):
            if fins[q] is not None:      # the sector must end where its own single tracker ended (same scans, same births): selections of every live track
                repq, recq = r.report()
                okm = okm and [(int(x["id"]), int(x["sel_meas"])) for x in recq if int(x["status"]) == 0] == fins[q]
        t_grow = [float(np.mean(v)) for v in t_grow]
        for gq in grps:
            gq.close()
        for r in rps:
            r.close()
        tmulti, okm = parallel.reduce_clock(tm1 - tm0, okm, dist, device="cuda")
        # algorithmic bytes of ALL sectors' grow stages of one scan over the duration of their batched grow launch(es)
        bytes_scan = sum(float(per_leaf) * st[W:W + Km, 0].mean() + float(per_pair) * st[W:W + Km, 1].mean() + 8.0 * st[W:W + Km, 2].mean() for st in sts)
        # all sectors' bytes over the time during which ANY group's grow launch of the scan was running (device stamps; rounds 2-5 multiplied the mean
        # launch time by the number of groups, i.e. assumed that the groups' launches run one after the other)
        tg = float(np.mean(t_union)) if t_union else float(np.sum(t_grow))
        gbs = bytes_scan / tg / 1e9 if tg > 0 else 0.0
        return {"sectors_per_gpu": S, "steps_per_sector": Km, "scans_per_sec": world * S * Km / tmulti,
                "ms_per_scan_aggregate": 1e3 * tmulti / (S * Km), "ok": okm, "groups": NG,
                "x_single_sector": None,
                "roofline": {"bound": "hbm", "kernel": "fgrow_batch_kernel: the grow stage of all sectors of a group in one launch; time = union of the groups' grow intervals per scan (absolute device wall-clock stamps, mean over 8 scans behind the timed ones)",
                             "algorithmic_bytes_all_sectors": bytes_scan, "grow_us_per_scan_all_groups": 1e6 * tg, "grow_us_per_group_launch": 1e6 * float(np.mean(t_grow)), "achieved": gbs, "peak": HBM_PEAK_GBS,
                             "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS},
                "note": "independent sectors in %d group(s), one batched launch set per group and scan (mht_group_step, grid.y = "
                        "sector), groups on separate HIP streams; the single-sector path is a chain of dependent round trips that "
                        "leaves most of the GPU idle" % NG}

    multi_all = []
    for S in [int(v) for v in str(args.sectors).split(",") if v.strip()]:
        if S > 1:
            try:      # (an optional extra: its failure must not cost the run its headline line)
                multi_all.append(run_multi(S))
            except Exception as e:      # noqa: BLE001
                multi_all.append({"sectors_per_gpu": S, "ok": False, "error": repr(e)[:300]})
    multi = multi_all[0] if multi_all else None

    timed = stats[W:W + K]
    Lm, Gm, Mm = float(timed[:, 0].mean()), float(timed[:, 1].mean()), float(timed[:, 2].mean())
    # algorithmic bytes of the grow stage per scan: per leaf read x, P, cNLLR, P_d and write x_bar, P_bar, cNLLR, P_hat;
    # per gated pair x_hat, cNLLR, measurement index, parent index; 8 B per measurement -- 280 / 48 / 8 at four states, 552 / 64 / 8 at six
    b_gate = float(per_leaf) * Lm + float(per_pair) * Gm + 8.0 * Mm
    gate_gbs = b_gate / (ms[0] * 1e-3) / 1e9
    traffic, traffic_src = PMC_TRAFFIC_BYTES.get(args.config), \
        "from profiles/r04_pmc_hbm_traffic.txt, not this run (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, bytes per full-size launch)"
    if args.pmc == "auto" and rank == 0 and world == 1:
        if os.environ.get("MHT_STALL_DEBUG") == "1": print("[bench] counter passes start at %.1f s" % (time.time() - T_START), file=sys.stderr, flush=True)
        live, note = pmc_traffic_live(args.config)
        if live is not None:
            traffic, traffic_src = live, note
        else:
            traffic_src += "; live counter passes unavailable (%s)" % note
    for m_ in multi_all:
        if "scans_per_sec" in m_:      # (a failed extra carries its error instead)
            m_["x_single_sector"] = m_["scans_per_sec"] / ((1 if strong else world) * K / elapsed)
    ilp_acc = ilp_accounting(sc, births, local, W, min(K, 32)) if rank == 0 else None
    extras_on = args.extras != "off" and rank == 0 and world == 1 and args.config == "cfg3" and not strong
    value_steady = value_windows = None
    configs_extra = []
    if extras_on:
        try:
            value_steady, value_windows = steady_windows(args.config, parallel.sector_seed(5446, srank), parallel.sector_centre(srank), local, W)
        except Exception as e:      # noqa: BLE001  (an extra: its failure must not cost the run its headline line)
            value_windows = {"error": repr(e)[:300]}
        for nm, wm, st_, cw, ct in (("cfg2", 16 + 10, 100, 8, 8), ("cfg5", 16 + 6, 60, 9, 2)):
            try:
                configs_extra.append(config_extra(nm, local, wm, st_, cw, ct if args.cpu_scans > 0 else 0, 25.0))
            except Exception as e:      # noqa: BLE001
                configs_extra.append({"name": nm, "error": repr(e)[:300]})
    out = {
        "metric": "scans/sec at ~5k+ leaf hypotheses x 500 measurements (per-scan gate + cluster + ILP + N-scan prune)",
        "value": (1 if strong else world) * K / elapsed, "unit": "scans/s", "n_gpus": world, "steps": K, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / K, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
        "dtype": "f64 state / f32 covariance (the reference's own mix)", "data": "synthetic",
        "config": {"workload": ("BASELINE configs[2]: 500 targets, ~500 meas/scan, N-scan=5, P_d=0.9, eta2=5.99; " if args.config == "cfg3" else
                                "config %s of pymht_amd/utils/scenario.py (NOT the headline workload): %d targets, N-scan=%d, %d-state model %s; " % (args.config, len(sc["x0"]), sc["N"], nx, model_of(sc).__name__.split(".")[-1])) +
                               "one independent sector per GPU" + (" -- STRONG: the same sector on every GPU, ILPs spread by cluster" if strong else ""), "name": args.config, "targets": int(timed[:, 6].mean()),
                   "leaves_per_scan": Lm, "gated_pairs_per_scan": Gm, "meas_per_scan": Mm, "n_scan": int(sc["N"]),
                   "ilps_per_scan": float(timed[:, 3].mean()), "ilps_branched": int(timed[:, 4].sum()), "tracks_all_sectors": int(sum(len(i) for i, _ in picture)),
                   "blp_dual_iters_max": int(timed[:, 5].max()), "replay_matches_prepass": same_work, "host_us_per_scan_queued": round(1e6 * (t_enq - t0) / K, 2), "scans_clustered_in_grow_launch": int(uf_ovl[0]), "grow_launches_overlapping_ilp": int(uf_ovl[1]),
                   "pre_roll_scans": PRE},
        "stage_ms": {"gate": float(ms[0]), "cluster": float(ms[1]), "ilp": float(ms[2]), "prune": float(ms[3]),
                     "device_total": float(ms[4])},
        "value_steady": value_steady,
        "value_windows": value_windows,
        "configs": configs_extra,
        "multi_sector": multi,
        "multi_sector_all": multi_all,
        "ilp": ilp_acc,
        "api_scans_per_sec": 1.0 / api_s,
        "api_note": "drop-in Tracker.addMeasurementList, streaming (scan k+2 is queued before the report of scan k is folded; results "
                    "read after the last scan): PCIe copy of every scan, steps 1-7 on the device (M-of-N initiator included), "
                    "report D2H + host mirror per scan; same scans as `value` (pre-roll and warm-up untimed)",
        "roofline": {"bound": "hbm", "achieved": gate_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": gate_gbs / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                     "kernel": "fgrow_kernel (gate + update + score + child creation + next scan's gains, 1 launch), HIP events on the ctx stream",
                     "algorithmic_bytes": b_gate},
    }
    if strong:
        # the Amdahl split of the sharded scan, from the device wall-clock stamps of the LAST timed scan on this rank (10 ns ticks in the
        # report header): grow and clustering are replicated on every rank, only the ILP stage shrinks with the rank count
        out["strong_stage_us"] = {"grow_replicated": rep.t_process * 1e-2, "cluster_replicated": rep.t_cluster * 1e-2,
                                  "ilp_this_rank": rep.t_optim * 1e-2, "scan_on_device": rep.t_scan * 1e-2,
                                  "note": "per scan: grow + cluster are replicated, the ILPs of this rank's clusters (placed by size), then one all-reduce(MAX) of T int32 and the per-target epilogue"}
    if rank == 0 and world == 1 and args.cpu_scans > 0:
        n_warm = min(args.cpu_warm, len(sc["scans"]) - args.cpu_scans)
        out["cpu_baseline"] = cpu_baseline(sc, n_warm, args.cpu_scans)
    else:
        out["cpu_baseline"] = None
    if rank == 0:
        if not same_work:
            out["warning"] = "timed replay diverged from the pre-pass"
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
