import numpy as np, sys, glob, time
import scipy.sparse as sp

def load(path):
    with open(path, "rb") as fh:
        hdr = np.fromfile(fh, np.int64, 4); Nf, nt, n, att = [int(v) for v in hdr]
        ti = np.fromfile(fh, np.int64, nt); tj = np.fromfile(fh, np.int64, nt); tv = np.fromfile(fh, np.float64, nt)
    A = sp.coo_matrix((tv, (ti, tj)), shape=(Nf, Nf)).tocsr()
    # triplets are lower-or-upper once each; symmetrise
    D = sp.diags(A.diagonal())
    K = A + A.T - D
    return K.tocsr(), n

def adjacency(K):
    N = K.shape[0]
    Kc = K.tocsr()
    adj = [set(Kc.indices[Kc.indptr[u]:Kc.indptr[u+1]].tolist()) - {u} for u in range(N)]
    return adj

def matching2(K, n, adj, lin):
    """phase 1: maximum matching of the 'linear' variables (lin[v]) to rows, grown from the variable side; phase 2: the remaining rows
    by augmenting paths over all variables (an augmenting path never unmatches a matched vertex)"""
    N = K.shape[0]
    sys.setrecursionlimit(100000)
    varrows = {v: sorted([i for i in adj[v] if i >= n], key=lambda i: len(adj[i])) for v in range(n)}
    rowvars = {i: sorted([v for v in adj[i] if v < n], key=lambda v: (not lin[v], len(adj[v]))) for i in range(n, N)}
    mv, mr = {}, {}
    def augv(v, seen):
        for i in varrows[v]:
            if i in seen: continue
            seen.add(i)
            if i not in mr or (lin[mr[i]] and augv(mr[i], seen)):
                mr[i] = v; mv[v] = i
                return True
        return False
    for v in sorted([v for v in range(n) if lin[v]], key=lambda v: len(varrows[v])):
        augv(v, set())
    def augr(i, seen):
        for v in rowvars[i]:
            if v in seen: continue
            seen.add(v)
            if v not in mv or augr(mv[v], seen):
                mv[v] = i; mr[i] = v
                return True
        return False
    for i in sorted(rowvars, key=lambda i: len(rowvars[i])):
        if i not in mr: augr(i, set())
    return mr

def matching(K, n, adj, pref):
    """maximum matching rows (u>=n) -> variables (u<n) among structural J entries; variables tried in order of pref (lower first)"""
    N = K.shape[0]
    rowvars = {i: sorted([v for v in adj[i] if v < n], key=lambda v: pref[v]) for i in range(n, N)}
    mv = {}   # var -> row
    mr = {}   # row -> var
    def aug(i, seen):
        for v in rowvars[i]:
            if v in seen: continue
            seen.add(v)
            if v not in mv or aug(mv[v], seen):
                mv[v] = i; mr[i] = v
                return True
        return False
    sys.setrecursionlimit(100000)
    # rows with fewest candidates first
    for i in sorted(rowvars, key=lambda i: len(rowvars[i])):
        aug(i, set())
    return mr

def min_degree(N, adj, groups, before=None):
    """min degree over 'groups' (list of lists of unknowns eliminated together); before[g]: set of groups that must precede g"""
    G = len(groups)
    gof = {}
    for g, mem in enumerate(groups):
        for u in mem: gof[u] = g
    gadj = [set() for _ in range(G)]
    for g, mem in enumerate(groups):
        for u in mem:
            for v in adj[u]:
                if gof[v] != g: gadj[g].add(gof[v])
    wt = [len(m) for m in groups]
    need = [0] * G
    after = [[] for _ in range(G)]
    if before is not None:
        for g in range(G):
            need[g] = len(before[g])
            for h in before[g]: after[h].append(g)
    alive = [True] * G
    order = []
    import heapq
    def deg(g): return sum(wt[h] for h in gadj[g])
    heap = [(deg(g), g) for g in range(G) if need[g] == 0]
    heapq.heapify(heap)
    cur = {g: deg(g) for g in range(G)}
    while heap:
        d, g = heapq.heappop(heap)
        if not alive[g] or d != cur[g] or need[g] != 0: continue
        alive[g] = False
        order.append(g)
        nb = gadj[g]
        for h in nb:
            gadj[h].discard(g)
            gadj[h] |= (nb - {h})
        for h in nb:
            cur[h] = deg(h)
            if need[h] == 0: heapq.heappush(heap, (cur[h], h))
        for h in after[g]:
            need[h] -= 1
            if need[h] == 0:
                cur[h] = deg(h); heapq.heappush(heap, (cur[h], h))
    assert len(order) == G, (len(order), G)
    return [groups[g] for g in order]

def factor_solve(K, blocks, b, pivot_inside=True):
    """dense-storage block LDL^T in the order given by blocks (lists of 1..4 unknowns); returns x, nnzL, min |pivot| info"""
    N = K.shape[0]
    perm = [u for blk in blocks for u in blk]
    P = np.array(perm)
    S = K[P][:, P].toarray()
    nnzL = 0
    k = 0
    Ls = []   # (k, bw, rows, Lblock, Binv)
    growth = 0.0
    for blk in blocks:
        bw = len(blk)
        B = S[k:k+bw, k:k+bw].copy()
        C = S[k+bw:, k:k+bw]
        rows = np.nonzero(np.any(C != 0.0, axis=1))[0]
        Binv = np.linalg.inv(B) if pivot_inside else None
        if not pivot_inside:
            # scalar LDL^T inside the block, no pivoting
            Binv = scalar_inv(B)
        Cr = C[rows]
        Lr = Cr @ Binv
        growth = max(growth, np.abs(Lr).max() if Lr.size else 0.0)
        if rows.size:
            S[np.ix_(k+bw+rows, k+bw+rows)] -= Lr @ Cr.T
        nnzL += rows.size * bw + bw * (bw - 1) // 2
        Ls.append((k, bw, rows + k + bw, Lr, Binv))
        k += bw
    # solve
    y = b[P].copy()
    for (k0, bw, rows, Lr, Binv) in Ls:
        if rows.size: y[rows] -= Lr @ y[k0:k0+bw]
    for (k0, bw, rows, Lr, Binv) in Ls:
        y[k0:k0+bw] = Binv @ y[k0:k0+bw]
    for (k0, bw, rows, Lr, Binv) in reversed(Ls):
        if rows.size: y[k0:k0+bw] -= Lr.T @ y[rows]
    x = np.empty(N); x[P] = y
    return x, nnzL, growth

def scalar_inv(B):
    """inverse through unpivoted scalar LDL^T (what the present kernels do), to expose cancellation"""
    bw = B.shape[0]
    L = np.eye(bw); d = np.zeros(bw); A = B.copy()
    for k in range(bw):
        d[k] = A[k, k]
        for i in range(k+1, bw):
            L[i, k] = A[i, k] / d[k]
        for i in range(k+1, bw):
            for j in range(k+1, i+1):
                A[i, j] -= L[i, k] * d[k] * L[j, k]; A[j, i] = A[i, j]
    Li = np.linalg.inv(L)
    return Li.T @ np.diag(1.0 / d) @ Li

def run(path, rng):
    K, n = load(path)
    N = K.shape[0]
    adj = adjacency(K)
    xt = rng.standard_normal(N)
    b = K @ xt
    res = {}
    # (a) constrained scalar: rows after all their variables
    singles = [[u] for u in range(N)]
    before = [set() for _ in range(N)]
    for i in range(n, N): before[i] = {v for v in adj[i] if v < n}
    for name, blocks in (("constrained", min_degree(N, adj, singles, before)), ("unconstrained", min_degree(N, adj, singles, None))):
        x, nnzL, gr = factor_solve(K, blocks, b, pivot_inside=False)
        r = np.abs(b - K @ x).max() / max(1.0, np.abs(b).max())
        res[name] = (nnzL, r, np.abs(x - xt).max() / np.abs(xt).max(), gr)
    # (b) matched pairs
    Kc = K.tocsr()
    hoff = np.zeros(n, int)        # off-diagonal entries inside the variable block: 0 = "linear-looking" variable
    for j in range(n):
        hoff[j] = sum(1 for v in adj[j] if v < n)
    deg = np.array([len(adj[u]) for u in range(N)])
    for pname, pref in (("lin2", None),):
        lin = hoff <= 1
        mr = matching2(K, n, adj, lin)
        nl_un = sum(1 for v in range(n) if lin[v] and v not in set(mr.values()))
        groups, used = [], set()
        for i, v in mr.items(): groups.append([v, i]); used |= {v, i}
        unmatched_rows = [i for i in range(n, N) if i not in used]
        for u in range(N):
            if u not in used: groups.append([u])
        gof = {}
        for g, mem in enumerate(groups):
            for u in mem: gof[u] = g
        bef = [set() for _ in groups]
        for i in unmatched_rows: bef[gof[i]] = {gof[v] for v in adj[i] if v < n}
        blocks = min_degree(N, adj, groups, bef)
        for inside in (True, False):
            x, nnzL, gr = factor_solve(K, blocks, b, pivot_inside=inside)
            r = np.abs(b - K @ x).max() / max(1.0, np.abs(b).max())
            res[f"pairs-{pname}-{'blk' if inside else 'scal'}"] = (nnzL, r, np.abs(x - xt).max() / np.abs(xt).max(), gr, len(unmatched_rows), nl_un, int(lin.sum()))
    return res

if __name__ == "__main__":
    rng = np.random.default_rng(0)
    files = sorted(glob.glob(sys.argv[1] + "*.bin"))
    step = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    for f in files[::step]:
        t0 = time.time()
        res = run(f, rng)
        print(f.split("/")[-1], f"{time.time()-t0:.0f}s")
        for k, v in res.items(): print("   %-22s nnzL %6d  relres %.1e  fwderr %.1e  max|L| %.1e %s" % (k, v[0], v[1], v[2], v[3], v[4:] if len(v) > 4 else ""))
        sys.stdout.flush()
