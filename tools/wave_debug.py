"""Development aid: one sector stepped solo (workgroup-per-target grow kernel) and as a group of one (wavefront-per-target kernel),
scan by scan; prints where the reports / leaf sets differ."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from test_sectors_gpu import _tracker, _sectors
from pymht_amd.sectors import SectorGroup
from pymht_amd.utils.classDefinitions import MeasurementList
n_scans = int(sys.argv[1]) if len(sys.argv) > 1 else 8
q = int(sys.argv[2]) if len(sys.argv) > 2 else 2
sc = _sectors(q + 1, n_scans)[q]
a, b = _tracker(sc), _tracker(sc)
grp = SectorGroup([b])
for k in range(n_scans):
    sl = MeasurementList(float(sc["times"][k]), sc["scans"][k])
    la0 = a.leafBatch()
    a.addMeasurementList(sl)
    grp.addMeasurementLists([sl])
    sa, sb = a._sel[0], b._sel[0]
    bad = [n for n in ("id", "status", "sel_meas", "sel_x", "sel_cnllr", "score", "root_scan", "root_meas", "root_x", "n_leaves", "cluster") if not np.array_equal(sa[n], sb[n])]
    print("scan", k, "L", a.lastScanStats["L"], b.lastScanStats["L"], "G", a.lastScanStats["G"], b.lastScanStats["G"], "diff fields", bad)
    if bad:
        idx = np.where(sa["n_leaves"] != sb["n_leaves"])[0]
        cnt0 = np.bincount(la0["target"], minlength=len(sa))
        for i in idx[:10]:
            print("   target row", i, "id", sa["id"][i], "leaves in", cnt0[i] if i < len(cnt0) else -1, "n_leaves solo/wave", sa["n_leaves"][i], sb["n_leaves"][i], "root_scan", sa["root_scan"][i], sb["root_scan"][i])
        la, lb = a.leafBatch(), b.leafBatch()
        print("   leaves after:", len(la["ID"]), len(lb["ID"]))
        break
grp.close(); a.close(); b.close()
