"""Prototype (host, NumPy) of the GPU 0-1 ILP solver: Lagrangian dual ascent + certificate + DFS branch and bound.
Development aid only (used to tune the kernel's constants against the recorded instances)."""
import sys, os, time
import numpy as np


def solve(cols, sizes, cost, max_iter=200, theta0=1.0, node_limit=200000, verbose=False):
    nT = len(sizes); nH = len(cols)
    starts = np.concatenate([[0], np.cumsum(sizes)]).astype(int)
    cost = np.asarray(cost, dtype=np.float64)
    nM = 1 + max((max(c) for c in cols if len(c)), default=-1)
    D = max((len(c) for c in cols), default=0)
    ent = -np.ones((nH, max(D, 1)), dtype=int)
    for h, c in enumerate(cols):
        ent[h, :len(c)] = c
    tgt = np.repeat(np.arange(nT), sizes)
    u = np.zeros(nM + 1)   # last = dummy for -1
    def rc_all(u):
        return cost + u[ent].sum(axis=1)   # u[-1] is dummy 0
    best_lb, best_u = -np.inf, u.copy()
    ub, ub_sel = np.inf, None
    theta = theta0
    stall = 0
    def greedy(u):
        rc = rc_all(u)
        order = np.argsort([rc[starts[t]:starts[t+1]].min() for t in range(nT)])
        used = np.zeros(nM + 1, bool); sel = [None]*nT; tot = 0.0
        for t in order:
            hs = np.arange(starts[t], starts[t+1])
            ok = ~(used[ent[hs]] & (ent[hs] >= 0)).any(axis=1)
            hs = hs[ok]
            h = hs[np.argmin(rc[hs])]
            sel[t] = h; tot += cost[h]
            e = ent[h]; used[e[e >= 0]] = True
        return tot, sel
    status = None
    for it in range(max_iter):
        rc = rc_all(u)
        sel = np.array([starts[t] + np.argmin(rc[starts[t]:starts[t+1]]) for t in range(nT)])
        usage = np.zeros(nM + 1, int)
        for h in sel:
            e = ent[h]; np.add.at(usage, e[e >= 0], 1)
        usage[-1] = 0
        lb = rc[sel].sum() - u[:nM].sum()
        if lb > best_lb + 1e-12:
            best_lb, best_u = lb, u.copy(); stall = 0
        else:
            stall += 1
            if stall >= 10:
                theta *= 0.5; stall = 0
        if usage.max() <= 1:
            c = cost[sel].sum()
            if c < ub: ub, ub_sel = c, list(sel)
            if not np.any((u[:nM] > 0) & (usage[:nM] == 0)):
                status = ('cert', it); break
        elif it % 5 == 0 or ub == np.inf:
            c, s = greedy(u)
            if c < ub: ub, ub_sel = c, s
        if ub - best_lb <= 1e-12 * max(1.0, abs(ub)):
            status = ('gap0', it); break
        g = (usage[:nM] - 1).astype(float)
        g[(u[:nM] <= 0) & (g < 0)] = 0
        nrm = (g * g).sum()
        if nrm == 0:
            status = ('cert', it); break
        step = theta * max(ub - lb, 1e-6) / nrm
        u[:nM] = np.maximum(0.0, u[:nM] + step * g)
    if status is not None:
        return sorted(int(h) for h in ub_sel), float(cost[ub_sel].sum()), status, 0
    # ---- branch and bound with the best prices ----
    if ub_sel is None:
        ub, ub_sel = greedy(best_u)
    u = best_u
    rc = rc_all(u)
    utot = u[:nM].sum()
    # static order: most contended first (targets whose best leaf conflicts), simple: by regret desc
    order = list(range(nT))
    nodes = 0
    best = [ub, list(ub_sel)]
    used = np.zeros(nM + 1, bool)
    chosen = [None] * nT
    def compat_mask(hs):
        e = ent[hs]
        return ~((used[e]) & (e >= 0)).any(axis=1)
    def rec(k, cost_so_far, u_used):
        nonlocal nodes
        nodes += 1
        if nodes > node_limit: return
        if k == nT:
            if cost_so_far < best[0] - 1e-12:
                best[0] = cost_so_far; best[1] = list(chosen)
            return
        # bound for remaining
        rem = 0.0
        mins = []
        for t in order[k:]:
            hs = np.arange(starts[t], starts[t+1]); ok = compat_mask(hs)
            if not ok.any(): return
            mins.append(rc[hs[ok]].min())
        lb = cost_so_far + sum(mins) - (utot - u_used)
        if lb >= best[0] - 1e-12: return
        t = order[k]
        hs = np.arange(starts[t], starts[t+1]); hs = hs[compat_mask(hs)]
        hs = hs[np.argsort(rc[hs], kind='stable')]
        others = sum(mins[1:])
        for h in hs:
            if cost_so_far + rc[h] + others - (utot - u_used) >= best[0] - 1e-12: break
            e = ent[h]; e = e[e >= 0]
            used[e] = True; chosen[t] = h
            rec(k + 1, cost_so_far + cost[h], u_used + u[e].sum())
            used[e] = False
    rec(0, 0.0, 0.0)
    st = ('bb', nodes) if nodes <= node_limit else ('limit', nodes)
    return sorted(int(h) for h in best[1]), float(best[0]), st, nodes


def load(path):
    g = np.load(path)
    out = []
    for i in range(int(g['n_inst'])):
        p = 'i%03d_' % i
        ptr, rows = g[p+'col_ptr'], g[p+'col_rows']
        cols = [rows[ptr[c]:ptr[c+1]].tolist() for c in range(len(ptr)-1)]
        out.append(dict(cols=cols, sizes=g[p+'sizes'].tolist(), cost=g[p+'cost'], sel=g[p+'sel'].tolist(), obj=float(g[p+'obj']), unique=bool(g[p+'unique'])))
    return out

if __name__ == '__main__':
    insts = load(sys.argv[1])
    stats = {}
    t0 = time.time()
    for i, I in enumerate(insts):
        sel, obj, st, nodes = solve(I['cols'], I['sizes'], I['cost'])
        ok = abs(obj - I['obj']) <= 1e-9 * max(1, abs(obj)) and (not I['unique'] or sel == I['sel'])
        stats.setdefault(st[0], []).append(st[1])
        if not ok or st[0] != 'cert':
            print(i, 'nH', len(I['cols']), 'nT', len(I['sizes']), st, 'ok' if ok else 'WRONG obj %.9f vs %.9f' % (obj, I['obj']))
    print({k: (len(v), int(np.mean(v)), int(np.max(v))) for k, v in stats.items()}, '%.1fs' % (time.time()-t0))
