"""Development aid: the branch-and-bound team on the recorded giant clusters (tests/golden g9 / g20) -- wall time and nodes; run it with
MHT_LIB_VARIANT=.w64 (a build with -DMHT_TEAM_W=64) to see what twice the members buy (= what a second device's team would add)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
from test_cluster_blp_gpu import gpu_blp, load_instances
from pymht_amd.device import Context
ctx = Context(0)
gold = os.path.join(ROOT, "tests", "golden")
for f in ("g9_ilp_giant.npz", "g20_ilp_hbm_team.npz"):
    for i, inst in enumerate(load_instances(os.path.join(gold, f))):
        for rep in range(2):
            out = gpu_blp(ctx, inst, max_iter=200)
        print(f, i, "targets", len(inst["sizes"]), "columns", len(inst["cols"]), "status", out[2], "nodes", out[4] if len(out) > 4 else None, "%.1f ms" % (1e3 * gpu_blp.last_call_s), "same selection", out[0] == inst["sel"].tolist())
ctx.close()
