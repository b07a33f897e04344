"""Development: the drop-in API path's device timeline.  Run under rocprofv3:
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/apitrace -- python tools/api_trace.py run [n_scans]
then
  python tools/api_trace.py report gpurun_out/apitrace
prints, over the steady-state scans: the scan period, per-kernel mean duration, the idle time of the stream between consecutive
kernels, and the scans with births next to the ones without."""
import csv, glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np


def run(n):
    import time
    from pymht_amd.tracker import Tracker
    from pymht_amd.pyTarget import Target
    from pymht_amd.models import pv
    from pymht_amd.utils.classDefinitions import MeasurementList
    from pymht_amd.utils.scenario import make_config
    sc = make_config("cfg3", seed=5446, n_scans=n, confine=True)
    lists = [MeasurementList(float(t), z) for t, z in zip(sc["times"], sc["scans"])]
    trk = Tracker(pv, sc["period"], sc["lambda_phi"], 1e-4, P_d=sc["P_d"], N=sc["N"], eta2=5.99, logScanStats=True)
    trk._add_targets([Target(sc["t0"], None, x.copy(), pv.P0, status="preinitialized") for x in sc["x0"]])
    nb = []
    orig = trk._apply_births

    def rec(b, *a):
        orig(b, *a)
        nb.append(int((b["id"] >= 0).sum()))
    trk._apply_births = rec
    for sl in lists[:32]:
        trk.addMeasurementList(sl)
    trk.synchronize()
    t0 = time.perf_counter()
    for sl in lists[32:]:
        trk.addMeasurementList(sl)
    trk.synchronize()
    dt = time.perf_counter() - t0
    print("%d scans streamed: %.1f us per scan = %.0f scans/s" % (n - 32, 1e6 * dt / (n - 32), (n - 32) / dt))
    nb = np.array(nb)      # (the hook only runs on scans whose report lists candidates)
    print("scans whose initiator confirmed candidates: %d of %d (%.1f %%), admitted per such scan %s" % (len(nb), n, 100.0 * len(nb) / n, nb.tolist()))
    trk.close()


def report(d):
    rows = []
    for p in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        with open(p) as fh:
            for r in csv.DictReader(fh):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    short = lambda n: n.split("(")[0].replace("void ", "").replace("mht::", "").replace("(anonymous namespace)::", "")[:44]
    rows = [(s, e, short(n)) for s, e, n in rows]
    starts = [i for i, r in enumerate(rows) if r[2].startswith("fgrow")]
    starts = starts[len(starts) // 3:]                  # steady state: the last two thirds
    per, gaps, durs, by_births = [], [], {}, {0: [], 1: []}
    for a, b in zip(starts[:-1], starts[1:]):
        seg = rows[a:b]
        per.append(rows[b][0] - rows[a][0])
        busy = sum(e - s for s, e, _ in seg)
        gaps.append(per[-1] - busy)
        for s, e, n in seg:
            durs.setdefault(n, []).append(e - s)
        ps = [e - s for s, e, n in seg if n.startswith("post_scan")]
        fg = rows[b][1] - rows[b][0]
        if ps:
            by_births[1 if ps[0] > 12000 else 0].append((ps[0], per[-1]))
    per, gaps = np.array(per), np.array(gaps)
    print("scans %d: period mean %.1f us p50 %.1f; stream idle (period - kernel time, overlapping kernels on other streams count as busy) mean %.1f us" % (
        len(per), per.mean() / 1e3, np.median(per) / 1e3, gaps.mean() / 1e3))
    for n, v in sorted(durs.items(), key=lambda kv: -np.sum(kv[1])):
        v = np.array(v)
        print("  %-46s per scan %.2f launches, mean %.2f us p50 %.2f max %.1f -> %.2f us per scan" % (n, len(v) / len(per), v.mean() / 1e3, np.median(v) / 1e3, v.max() / 1e3, v.sum() / len(per) / 1e3))
    for k in (0, 1):
        if by_births[k]:
            a = np.array(by_births[k])
            print("  post_scan %s 12 us: %d scans, post_scan mean %.1f us, period mean %.1f us" % ("<=" if k == 0 else ">", len(a), a[:, 0].mean() / 1e3, a[:, 1].mean() / 1e3))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(int(sys.argv[2]) if len(sys.argv) > 2 else 432)
    else:
        report(sys.argv[2])
