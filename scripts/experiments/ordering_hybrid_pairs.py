import numpy as np, sys, glob
exec(open('blockpiv.py').read().split('if __name__')[0])
def phase1(K, n, adj, lin):
    N = K.shape[0]
    sys.setrecursionlimit(100000)
    varrows = {v: sorted([i for i in adj[v] if i >= n], key=lambda i: len(adj[i])) for v in range(n)}
    mv, mr = {}, {}
    def augv(v, seen):
        for i in varrows[v]:
            if i in seen: continue
            seen.add(i)
            if i not in mr or augv(mr[i], seen):
                mr[i] = v; mv[v] = i
                return True
        return False
    for v in sorted([v for v in range(n) if lin[v]], key=lambda v: len(varrows[v])):
        augv(v, set())
    return mr
rng = np.random.default_rng(0)
files = sorted(glob.glob(sys.argv[1] + "*.bin"))
for f in files[::int(sys.argv[2])]:
    K, n = load(f); N = K.shape[0]; adj = adjacency(K)
    hoff = np.array([sum(1 for v in adj[j] if v < n) for j in range(n)])
    lin = hoff <= 1
    mr = phase1(K, n, adj, lin)
    groups, used = [], set()
    for i, v in mr.items(): groups.append([v, i]); used |= {v, i}
    free_rows = [i for i in range(n, N) if i not in used]
    for u in range(N):
        if u not in used: groups.append([u])
    gof = {}
    for g, mem in enumerate(groups):
        for u in mem: gof[u] = g
    bef = [set() for _ in groups]
    for i in free_rows: bef[gof[i]] = {gof[v] for v in adj[i] if v < n}
    blocks = min_degree(N, adj, groups, bef)
    xt = rng.standard_normal(N); b = K @ xt
    for inside in (True, False):
        x, nnzL, gr = factor_solve(K, blocks, b, pivot_inside=inside)
        print(f.split('/')[-1], "hybrid", "blk" if inside else "scal", "pairs", len(mr), "free rows", len(free_rows), "nnzL", nnzL, "fwderr %.1e max|L| %.1e" % (np.abs(x - xt).max() / np.abs(xt).max(), gr), flush=True)
