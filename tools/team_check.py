"""Development aid: dense scenarios whose clusters need a long branch and bound -- the same stream with the ILP teams on and off
(MHT_BLP_NO_TEAMS), selections compared scan by scan, device time of the optimisation stage printed.
python tools/team_check.py T RADIUS N SCANS SEED [LAM]"""
import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np


def run(T, radius, N, scans, seed, lam):
    from pymht_amd.tracker import Tracker
    from pymht_amd.pyTarget import Target
    from pymht_amd.models import pv
    from pymht_amd.utils.scenario import make_scenario
    from pymht_amd.utils.classDefinitions import MeasurementList
    sc = make_scenario(T=T, radius=radius, lambda_phi=lam, n_scans=scans, P_d=0.9, seed=seed)
    trk = Tracker(pv, sc["period"], sc["lambda_phi"], 1e-4, P_d=sc["P_d"], N=N, eta2=5.99, useInitiator=False, maxTargets=512, maxNodes=1 << 19)
    trk._add_targets([Target(sc["t0"], None, x.copy(), pv.P0, status="preinitialized") for x in sc["x0"]])
    out = []
    for z, t in zip(sc["scans"], sc["times"]):
        trk.addMeasurementList(MeasurementList(float(t), z))
        st = trk.lastScanStats
        sel = trk._sel[0]
        out.append(dict(L=int(st["L"]), ilp=int(st["ilp"]), branched=int(st["branched"]), limit=int(st["limit"]), optim_ms=1e3 * trk.toc["Optim"],
                        sel=[int(v) for v in sel["sel_meas"]], ids=[int(v) for v in sel["id"]], cost=float(np.sum(sel["sel_cnllr"]))))
    trk.close()
    return out


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        T, radius, N, scans, seed = int(sys.argv[2]), float(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
        lam = float(sys.argv[7])
        print("RESULT " + json.dumps(run(T, radius, N, scans, seed, lam)))
        sys.exit(0)
    args = sys.argv[1:6]
    lam = sys.argv[6] if len(sys.argv) > 6 else "2e-5"
    res = {}
    for teams in (1, 0):
        env = dict(os.environ, MHT_BLP_NO_TEAMS="0" if teams else "1")
        o = subprocess.run([sys.executable, __file__, "child"] + args + [lam], env=env, capture_output=True, text=True, timeout=1500)
        line = [l for l in o.stdout.splitlines() if l.startswith("RESULT ")]
        if not line:
            print("teams=%d FAILED:\n%s\n%s" % (teams, o.stdout[-2000:], o.stderr[-3000:]))
            sys.exit(1)
        res[teams] = json.loads(line[0][7:])
    same = all(a["sel"] == b["sel"] and a["ids"] == b["ids"] for a, b in zip(res[1], res[0]))
    print("T=%s r=%s N=%s scans=%s seed=%s lam=%s: selections identical with / without teams: %s" % (*args, lam, same))
    for k, (a, b) in enumerate(zip(res[1], res[0])):
        print("  scan %2d L=%6d ilp=%3d branched=%d/%d limit=%d/%d  optim ms: teams %.2f  solo %.2f" % (k, a["L"], a["ilp"], a["branched"], b["branched"], a["limit"], b["limit"], a["optim_ms"], b["optim_ms"]))
