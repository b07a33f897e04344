"""Synthetic radar scan streams for parity tests and bench.py.

Own generator (np.random.default_rng) -- the reference's simulator (pymht/utils/simulator.py)
draws from the global np.random state through an SVD-based multivariate_normal and is not
bit-stable across platforms, so scenario *inputs* are generated here and, for golden traces,
stored in the fixtures.

A scenario is a dict:
    x0      (T,4) f64   initial target states [x, y, vx, vy]
    scans   list of (M_k,2) f32 measurement arrays (detections + uniform clutter, shuffled)
    times   (K,) f64    scan time stamps, times[k] = t0 + (k+1)*period
    t0      float       time stamp of the initial states
    plus the parameters it was drawn with.
"""
import numpy as np

CONFIGS = {
    # name: T, radius [m], lambda_phi [1/m^2], N (N-scan window), scans  -- BASELINE.json configs 1-3
    "cfg1": dict(T=2, radius=1000.0, lambda_phi=2.5e-6, N=3, n_scans=20, P_d=0.9),
    "cfg2": dict(T=50, radius=700.0, lambda_phi=1e-4, N=3, n_scans=30, P_d=0.9),
    "cfg3": dict(T=500, radius=5000.0, lambda_phi=6.4e-7, N=5, n_scans=30, P_d=0.9),
    "dense": dict(T=20, radius=400.0, lambda_phi=2e-5, N=3, n_scans=12, P_d=0.9),
    # BASELINE.json config 5 by SIZE (2 k targets, ~2 k measurements per scan, N-scan 6) with the reference's own 4-state CV model:
    # the 6-state CT model it names does not exist in the reference.  Same target density as cfg3.
    "cfg5": dict(T=2000, radius=10000.0, lambda_phi=4.8e-7, N=6, n_scans=12, P_d=0.9),
}


def make_scenario(T, radius, lambda_phi, n_scans, P_d=0.9, period=2.5, sigma_r=2.5, sigma_v=8.0,
                  sigma_q=0.05, seed=1234, centre=(0.0, 0.0), t0=1000.0, confine=False, **_unused):
    """confine=True: a target outside the disc of radius 0.8 * radius (where the initial targets are drawn) is pulled back by a
    gentle acceleration towards the centre (0.15 m/s^2 = 0.5 m per scan, far inside the process noise of the tracking model: it follows
    without losing the track), so the target density -- and with it the number of clusters / ILPs per scan -- stays what it is at
    the start however many scans are generated (bench.py: a short and a long timed window then see the same workload).
    Without it the targets disperse (sigma_v = 8 m/s) and the scene thins out over a few hundred scans."""
    rng = np.random.default_rng(seed)
    centre = np.asarray(centre, dtype=np.float64)
    r = radius * 0.8 * np.sqrt(rng.uniform(size=T))
    th = rng.uniform(0.0, 2.0 * np.pi, size=T)
    x = np.empty((T, 4), dtype=np.float64)
    x[:, 0] = centre[0] + r * np.cos(th)
    x[:, 1] = centre[1] + r * np.sin(th)
    x[:, 2:] = rng.normal(0.0, sigma_v, size=(T, 2))
    x0 = x.copy()
    area = np.pi * radius * radius
    scans, times, truth = [], [], []
    for k in range(n_scans):
        # constant-velocity truth with a small white acceleration
        acc = rng.normal(0.0, sigma_q, size=(T, 2))
        if confine:
            rel = x[:, 0:2] - centre
            rr = np.sqrt((rel * rel).sum(axis=1))
            out = rr > 0.8 * radius
            if out.any():
                acc[out] -= 0.15 * rel[out] / rr[out, None]
        x[:, 0:2] += period * x[:, 2:4] + 0.5 * period * period * acc
        x[:, 2:4] += period * acc
        truth.append(x.copy())
        seen = rng.uniform(size=T) <= P_d
        det = x[seen, 0:2] + rng.normal(0.0, sigma_r, size=(int(seen.sum()), 2))
        n_cl = rng.poisson(lambda_phi * area)
        rc = radius * np.sqrt(rng.uniform(size=n_cl))
        tc = rng.uniform(0.0, 2.0 * np.pi, size=n_cl)
        clutter = np.stack([centre[0] + rc * np.cos(tc), centre[1] + rc * np.sin(tc)], axis=1)
        z = np.concatenate([det, clutter], axis=0)
        rng.shuffle(z, axis=0)
        scans.append(np.ascontiguousarray(z, dtype=np.float32).reshape(-1, 2))
        times.append(t0 + (k + 1) * period)
    return dict(x0=x0, scans=scans, times=np.asarray(times, dtype=np.float64), t0=float(t0),
                period=float(period), P_d=float(P_d), lambda_phi=float(lambda_phi), radius=float(radius),
                centre=centre, seed=int(seed), truth_final=x.copy(), truth=truth)


def make_ais(sc, seed=77, equipped=0.5, p_report=0.7, offsets=(0.25, 0.5, 0.75), first_mmsi=257000000):
    """AIS traffic for a scenario (the reference's simulator makes such lists with `simulateAIS`, simulator.py:112-172): a fixed
    share of the targets carries a transponder (mmsi = first_mmsi + target index); in every radar period each of them reports with
    probability `p_report`, at one of a few instants inside the period (so that several messages share a time and several times
    occur, tracker.py:429, :447-465), its state [x, y, vx, vy] at that instant plus noise of the accuracy class it claims
    (models/ais.py:6-13: sigma 1.0 high / 3.0 low, all four components).  Returns per scan a list of
    (time, state float64[4], mmsi, highAccuracy) in target order."""
    rng = np.random.default_rng(seed)
    T = len(sc["x0"])
    has = rng.uniform(size=T) < equipped
    period = sc["period"]
    out = []
    for k, t in enumerate(sc["times"]):
        xk = sc["truth"][k]                                     # truth AT the scan
        msgs = []
        for i in np.flatnonzero(has):
            if rng.uniform() >= p_report:
                continue
            off = float(offsets[int(rng.integers(0, len(offsets)))])
            tm = float(t) - (1.0 - off) * period                 # strictly inside (t - period, t)
            st = xk[i].copy()
            st[0:2] -= (float(t) - tm) * st[2:4]                 # back along the velocity to the message's instant
            high = bool(rng.uniform() > 0.5)
            st = st + rng.normal(0.0, 1.0 if high else 3.0, size=4)
            msgs.append((tm, st, int(first_mmsi + i), high))
        out.append(msgs)
    return out


def make_config(name, seed=1234, **overrides):
    cfg = dict(CONFIGS[name])
    cfg.update(overrides)
    sc = make_scenario(seed=seed, **cfg)
    sc["N"] = cfg["N"]
    sc["name"] = name
    return sc
