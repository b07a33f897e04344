"""Development: where the G20 instance's time goes -- the HBM dual phase alone (node_limit 1) against the whole solve, by team width (MHT_BLP_NO_TEAMS=1: one workgroup)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from test_cluster_blp_gpu import gpu_blp, load_instances
from pymht_amd.device import Context
ctx = Context(0)
inst = load_instances(os.path.join(ROOT, "tests", "golden", "g20_ilp_hbm_team.npz"))[0]
for mi, nl in ((200, 1 << 22), (200, 1), (64, 1), (24, 1), (0, 1)):
    for rep in range(2):
        try:
            sel, obj, status, iters, nodes = gpu_blp(ctx, inst, max_iter=mi, node_limit=nl)
            print("max_iter %d node_limit %d: status %d iters %d nodes %d  %.1f ms" % (mi, nl, status, iters, nodes, 1e3 * gpu_blp.last_call_s))
        except Exception as e:
            print("max_iter %d node_limit %d: %s  %.1f ms" % (mi, nl, repr(e)[:80], 1e3 * gpu_blp.last_call_s))
