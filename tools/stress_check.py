"""Development aid: (big) 1800 targets / ~47 k leaves per scan against the oracle, scan by scan; (long N) N scans of the headline
config through the drop-in Tracker to watch counters over a long run.  python tools/stress_check.py big | long 3000"""
import sys, os, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/oracle'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
import mht_oracle as orc
from pymht_amd.utils.scenario import make_scenario
from pymht_amd.utils.classDefinitions import MeasurementList
from pymht_amd.tracker import Tracker
from pymht_amd.pyTarget import Target
from pymht_amd.models import pv
mode = sys.argv[1]
if mode == 'big':
    sc = make_scenario(T=1800, radius=9500.0, lambda_phi=3.5e-7, n_scans=7, P_d=0.9, seed=77)
    N = 5
    trk = Tracker(pv, sc["period"], sc["lambda_phi"], 1e-4, P_d=sc["P_d"], N=N, eta2=5.99, useInitiator=False, maxTargets=2048, maxNodes=1 << 19, maxMeasurements=2048, deviceTiming=True)
    trk._add_targets([Target(sc["t0"], None, x.copy(), pv.P0) for x in sc["x0"]])
    o = orc.OracleTracker(sc["period"], sc["lambda_phi"], 1e-4, P_d=sc["P_d"], N=N, eta2=5.99)
    t0 = time.time()
    for x0 in sc["x0"]: o.initiate_target(sc["t0"], x0.copy(), orc.model_P0())
    print('oracle init %.0fs, accepted %d / tracker %d' % (time.time() - t0, len(o.targets), trk.nTargets))
    for k, (z, t) in enumerate(zip(sc["scans"], sc["times"])):
        t0 = time.time(); info = o.add_scan(float(t), z); dt = time.time() - t0
        trk.addMeasurementList(MeasurementList(float(t), z))
        st = trk.lastScanStats
        want = o.selected(); sel = trk._sel[0]
        print(k, 'L', st['L'], 'G', st['G'], 'M', st['M'], 'ilp', st['ilp'], 'branched', st['branched'], 'iters', st['blp_iters_max'], 'oracle %.1fs' % dt,
              'gating', (st["L"], st["G"]) == (info["L"], info["G"]), 'sel', sel["id"].tolist() == want["ID"].tolist() and sel["sel_meas"].tolist() == want["meas"].tolist(),
              'stage us', {k2: round(1e6 * trk.toc[k2]) for k2 in ('Process', 'Cluster', 'Optim', 'N-Prune')})
else:
    import bench
    from pymht_amd.utils.scenario import make_config
    n = int(sys.argv[2])
    sc = make_config('cfg3', seed=5446, n_scans=n)
    trk = bench.make_tracker(sc, 0, deviceTiming=False)
    t0 = time.time()
    for k, (z, t) in enumerate(zip(sc["scans"], sc["times"])):
        trk.addMeasurementList(MeasurementList(float(t), z))
        if k % 500 == 0 or k == n - 1:
            st = trk.lastScanStats
            print(k, 'targets', trk.nTargets, 'L', st['L'], 'ilp', st['ilp'], 'iters', st['blp_iters_max'], 'branched', st['branched'], '%.1fs' % (time.time() - t0), flush=True)
