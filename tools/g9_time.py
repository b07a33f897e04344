"""Development aid: the giant ILP instances (tests/golden/g9_ilp_giant.npz) through the stateless seam, with and without the
reduced-cost fixing + LDS re-solve (MHT_BLP_NO_REDUCE=1)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
from test_cluster_blp_gpu import load_instances, gpu_blp
from pymht_amd.device import Context
ctx = Context(0)
for env in (sys.argv[1:] or ["0", "1"]):
    os.environ["MHT_BLP_NO_REDUCE"] = env
    insts = load_instances(os.path.join(ROOT, "tests/golden/g9_ilp_giant.npz"))
    if os.environ.get("G9_ONLY"): insts = [insts[int(os.environ["G9_ONLY"])]]
    for inst in insts * int(os.environ.get("G9_REPEAT", "1")):
        t0 = time.time()
        try:
            sel, obj, st, it, nd = gpu_blp(ctx, inst)
        except Exception as e:
            print("no_reduce=%s K=%d nH=%d: ERROR %s after %.1f ms" % (env, len(inst["sizes"]), len(inst["cols"]), repr(e)[:120], 1e3 * (time.time() - t0)), flush=True)
            continue
        dt = gpu_blp.last_call_s      # (the C call alone; packing 18 k columns in Python takes longer than solving them)
        used = {}
        feas = True
        for h in sel:
            for m in inst["cols"][h]:
                feas = feas and (int(m) not in used)
                used[int(m)] = 1
        print("feasible=%s true objective of the selection %.9f" % (feas, float(sum(inst["cost"][h] for h in sel))))
        print("no_reduce=%s K=%d nH=%d: status %d iters %d nodes %d  %.1f ms  obj %.9f (ref %.9f) sel ok=%s" % (
            env, len(inst["sizes"]), len(inst["cols"]), st, it, nd, 1e3 * dt, obj, inst["obj"], sel == inst["sel"].tolist()), flush=True)
