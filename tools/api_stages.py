import sys
sys.path.insert(0, "/root/repo")
import numpy as np
from pymht_amd.tracker import Tracker
from pymht_amd.pyTarget import Target
from pymht_amd.models import pv
from pymht_amd.utils.classDefinitions import MeasurementList
from pymht_amd.utils.scenario import make_config
sc = make_config("cfg3", seed=5446, n_scans=432, confine=True)
for ui in (True, False):
    trk = Tracker(pv, sc["period"], sc["lambda_phi"], 1e-4, P_d=sc["P_d"], N=sc["N"], eta2=5.99, useInitiator=ui)
    trk._add_targets([Target(sc["t0"], None, x.copy(), pv.P0, status="preinitialized") for x in sc["x0"]])
    for z, t in zip(sc["scans"], sc["times"]):
        trk.addMeasurementList(MeasurementList(float(t), z))
    trk.synchronize()
    lg = trk._runtimeLog_
    f = lambda k: 1e6 * float(np.median(np.array(lg[k][100:])))
    print("useInitiator", ui, "median stage times (device stamps, start to start): Process %.1f Cluster %.1f Optim %.1f us; keys %s" % (f("Process"), f("Cluster"), f("Optim"), sorted(lg)))
    trk.close()
