"""Development: the G20 giant-cluster instance (HBM team) through mht_solve_blp: time, nodes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from test_cluster_blp_gpu import gpu_blp, load_instances
from pymht_amd.device import Context
ctx = Context(0)
inst = load_instances(os.path.join(ROOT, "tests", "golden", "g20_ilp_hbm_team.npz"))[0]
for rep in range(2):
    sel, obj, status, iters, nodes = gpu_blp(ctx, inst, max_iter=200, node_limit=1 << 22)
    print("variant %r: status %d iters %d nodes %d  %.1f ms  ok=%s" % (os.environ.get("MHT_LIB_VARIANT", ""), status, iters, nodes, 1e3 * gpu_blp.last_call_s, sel == inst["sel"].tolist()))
