"""Prototype (host, NumPy) of the exact small-cluster fallback of the ILP kernel: contested-row signatures, Pareto dominance,
enumeration.  Reads instances dumped by tools/blp_tail.py (gpurun_out/ilp/*.npz).  Development aid only."""
import glob, itertools, sys
import numpy as np


def reduce_instance(sizes, cost, cols):
    K = len(sizes)
    starts = np.concatenate([[0], np.cumsum(sizes)])
    tgt = np.repeat(np.arange(K), sizes)
    users = {}
    for h in range(len(cost)):
        for m in cols[h]:
            if m >= 0: users.setdefault(int(m), set()).add(int(tgt[h]))
    contested = sorted(m for m, u in users.items() if len(u) >= 2)
    cid = {m: i for i, m in enumerate(contested)}
    sig = np.zeros(len(cost), dtype=object)
    for h in range(len(cost)):
        s = 0
        for m in cols[h]:
            if int(m) in cid: s |= 1 << cid[int(m)]
        sig[h] = s
    surv = []
    for k in range(K):
        hs = list(range(starts[k], starts[k + 1]))
        keep = []
        for h in hs:
            dom = False
            for g in hs:
                if g == h: continue
                if (sig[g] & ~sig[h]) == 0 and (cost[g] < cost[h] or (cost[g] == cost[h] and (sig[g] != sig[h] or g < h))):
                    dom = True; break
            if not dom: keep.append(h)
        keep.sort(key=lambda h: cost[h])
        surv.append(keep)
    return len(contested), sig, surv


def dfs(cost, sig, surv):
    K = len(surv)
    order = sorted(range(K), key=lambda k: len(surv[k]))
    rest = [0.0] * (K + 1)
    for i in range(K - 1, -1, -1): rest[i] = rest[i + 1] + cost[surv[order[i]][0]]
    best = [np.inf, None]
    nodes = [0]
    def rec(i, mask, acc, pick):
        if i == K:
            if acc < best[0]: best[0], best[1] = acc, list(pick)
            return
        for h in surv[order[i]]:
            if acc + cost[h] + rest[i + 1] >= best[0]: break
            if sig[h] & mask: continue
            nodes[0] += 1
            pick.append(h)
            rec(i + 1, mask | sig[h], acc + cost[h], pick)
            pick.pop()
    rec(0, 0, 0.0, [])
    return best[0], best[1], nodes[0]


for p in sorted(glob.glob(sys.argv[1] + '/*.npz')):
    d = np.load(p)
    sizes, cost, cols = d['sizes'], d['cost'], d['cols']
    nc, sig, surv = reduce_instance(sizes, cost, cols)
    prod = float(np.prod([float(len(s)) for s in surv]))
    obj, pick, nodes = dfs(cost, sig, surv)
    print('%-28s K=%d nH=%4d gpu %6.1f us it=%2d st=%d nodes=%2d | contested rows %3d, survivors %s, product %.3g, dfs nodes %d, obj %.6f' % (
        p.split('/')[-1], len(sizes), len(cost), float(d['us']), int(d['iters']), int(d['status']), int(d['nodes']), nc, [len(s) for s in surv], prod, nodes, obj))

print('--- no dominance: sort by cost only; DFS nodes sequential / with prefix split (2 levels) and optimal incumbent')
for p in sorted(glob.glob(sys.argv[1] + '/*.npz')):
    d = np.load(p)
    sizes, cost, cols = d['sizes'], d['cost'], d['cols']
    nc, sig, surv = reduce_instance(sizes, cost, cols)
    starts = np.concatenate([[0], np.cumsum(sizes)])
    full = [sorted(range(starts[k], starts[k + 1]), key=lambda h: (cost[h], h)) for k in range(len(sizes))]
    obj, pick, nodes = dfs(cost, sig, full)
    # parallel estimate: every prefix over the two shortest lists searched independently with the optimum as incumbent
    K = len(full)
    order = sorted(range(K), key=lambda k: len(full[k]))
    rest = [0.0] * (K + 1)
    for i in range(K - 1, -1, -1): rest[i] = rest[i + 1] + cost[full[order[i]][0]]
    per = []
    P = min(2, K - 1)
    def count(i, mask, acc):
        n = 0
        for h in full[order[i]]:
            if acc + cost[h] + rest[i + 1] > obj + 1e-12: break
            if sig[h] & mask: continue
            n += 1
            if i + 1 < K: n += count(i + 1, mask | sig[h], acc + cost[h])
            else: break
        return n
    tot = 0
    def prefixes(i, mask, acc):
        global tot
        if i == P:
            per.append(count(i, mask, acc)); return
        for h in full[order[i]]:
            if acc + cost[h] + rest[i + 1] > obj + 1e-12: break
            if sig[h] & mask: continue
            tot += 1
            prefixes(i + 1, mask | sig[h], acc + cost[h])
    prefixes(0, 0, 0.0)
    print('%-28s K=%d lists %s: seq nodes %d | prefixes alive %d, nodes per prefix max %d mean %.1f' % (p.split('/')[-1], K, [len(f) for f in full], nodes, len(per), max(per) if per else 0, np.mean(per) if per else 0))
