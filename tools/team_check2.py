"""Development aid: team_hunt's scenario for a seed, with the ILP teams on and off (subprocesses): selections compared, optimisation-stage times printed."""
import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np


def run(s):
    from pymht_amd.tracker import Tracker
    from pymht_amd.pyTarget import Target
    from pymht_amd.models import pv
    from pymht_amd.utils.scenario import make_scenario
    from pymht_amd.utils.classDefinitions import MeasurementList
    rng = np.random.default_rng(s)
    T = int(rng.integers(45, 71)); radius = float(rng.uniform(150, 300)); N = int(rng.integers(4, 8)); lam = float(rng.choice([1e-5, 5e-5, 1.5e-4]))
    if os.environ.get("HUNT_SMALL"):
        N = int(rng.integers(2, 5)); T = int(rng.integers(55, 71)); radius = float(rng.uniform(180, 260))
    P_d = float(rng.uniform(0.6, 0.95)); eta2 = float(rng.choice([5.99, 9.21])); period = float(rng.choice([1.0, 2.5]))
    sc = make_scenario(T=T, radius=radius, lambda_phi=lam, n_scans=9, P_d=P_d, period=period, seed=s)
    trk = Tracker(pv, sc["period"], sc["lambda_phi"], 1e-4, P_d=sc["P_d"], N=N, eta2=eta2, useInitiator=False, maxTargets=512, maxNodes=1 << 19)
    trk._add_targets([Target(sc["t0"], None, x.copy(), pv.P0, status="preinitialized") for x in sc["x0"]])
    out = []
    for z, t in zip(sc["scans"], sc["times"]):
        trk.addMeasurementList(MeasurementList(float(t), z))
        st = trk.lastScanStats
        sel = trk._sel[0]
        out.append(dict(L=int(st["L"]), ilp=int(st["ilp"]), branched=int(st["branched"]), limit=int(st["limit"]), optim_ms=1e3 * trk.toc["Optim"],
                        sel=[int(v) for v in sel["sel_meas"]], ids=[int(v) for v in sel["id"]]))
        if st["L"] > 60000: break
    trk.close()
    return out


if __name__ == "__main__":
    if sys.argv[1] == "child":
        print("RESULT " + json.dumps(run(int(sys.argv[2]))))
        sys.exit(0)
    for seed in sys.argv[1:]:
        res = {}
        for teams in (1, 0):
            env = dict(os.environ, MHT_BLP_NO_TEAMS="0" if teams else "1")
            o = subprocess.run([sys.executable, __file__, "child", seed], env=env, capture_output=True, text=True, timeout=900)
            line = [l for l in o.stdout.splitlines() if l.startswith("RESULT ")]
            if not line:
                print("seed %s teams=%d FAILED:\n%s\n%s" % (seed, teams, o.stdout[-1500:], o.stderr[-2500:]))
                break
            res[teams] = json.loads(line[0][7:])
        if len(res) < 2: continue
        same = all(a["sel"] == b["sel"] and a["ids"] == b["ids"] for a, b in zip(res[1], res[0]))
        print("seed %s: selections identical with / without teams: %s; optim ms per scan (teams | solo):" % (seed, same))
        print("   " + "  ".join("%.2f|%.2f%s" % (a["optim_ms"], b["optim_ms"], "*" if a["branched"] else "") for a, b in zip(res[1], res[0])))
