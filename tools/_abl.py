import os, sys
sys.path.insert(0, '/root/repo')
import numpy as np, bench
from pymht_amd.utils.scenario import make_config
from pymht_amd.utils.classDefinitions import MeasurementList
sc = make_config('cfg3', seed=5446, n_scans=20)
trk = bench.make_tracker(sc, 0)
pr = []
for k, (z, t) in enumerate(zip(sc['scans'], sc['times'])):
    trk.addMeasurementList(MeasurementList(float(t), z))
    if k >= 8 and k < 14: pr.append(1e6 * trk.toc['N-Prune'])
print('ablate', os.environ.get('MHT_PRUNE_ABLATE'), 'prune stage us (scans 8-13):', np.round(pr, 1))
