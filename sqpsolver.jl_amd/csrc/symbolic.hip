// symbolic.hip -- host-only: symbolic analysis of the sparse Newton matrix of the interior-point method.
//
// Seat in the reference: the analysis phase of Ipopt's linear solver (MUMPS / MA57 ordering + symbolic
// factorisation, /root/reference/examples/acopf/opf.jl:59-64), which the reference reaches through
// JuMP.optimize! (/root/reference/src/algorithms/subproblem_JuMP.jl:178).  Done once per sparsity structure.
//
//   1. ordering: approximate minimum degree on the quotient graph (Amestoy, Davis & Duff 1996: elements,
//      element absorption, the |Le \ Lp| bound for the external degree), with one constraint of our own: a row
//      of the matrix becomes ELIGIBLE only once every variable it couples to has been eliminated.  The Newton
//      matrix is quasi-definite with a 1e-8 regularisation in the row block; a row pivoted before its variables
//      has a pivot of that size and the elimination through it loses the digits (DESIGN.md section 3 -- measured in
//      round 1 on the tile order).  Behind its variables a row has the pivot -(D + J W^-1 J') of the
//      variables-first order.
//   2. elimination tree, column structures by child merging, postorder;
//   3. supernodes (maximal chains with nested structures), relaxed amalgamation: a child is merged into its parent
//      while the merged front is tiny or the explicit zeros stay below a fraction of the supernode's entries --
//      dense fronts a wave or a workgroup can hold beat exact sparsity on this hardware;
//   4. the multifrontal plan: per supernode its columns, row structure, front offset, children and the relative
//      indices of every child's contribution block inside the parent's front; level sets of the assembly tree.
// Everything is deterministic (ties by lowest index).
#include "sparse.hpp"
#include "../../include/sqphip.h"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <numeric>

namespace sqphip {

namespace {

struct Buckets {
    std::vector<int> head, next, prev, where;
    int lo = 0;
    explicit Buckets(int n) : head(n + 1, -1), next(n, -1), prev(n, -1), where(n, -1) {}
    void insert(int i, int d)
    {
        where[i] = d; prev[i] = -1; next[i] = head[d];
        if (head[d] >= 0) prev[head[d]] = i;
        head[d] = i;
        if (d < lo) lo = d;
    }
    void remove(int i)
    {
        const int d = where[i];
        if (d < 0) return;
        if (prev[i] >= 0) next[prev[i]] = next[i]; else head[d] = next[i];
        if (next[i] >= 0) prev[next[i]] = prev[i];
        where[i] = -1;
    }
    // lowest-degree vertex; ties: the lowest index in the bucket (deterministic whatever the insertion history)
    int pop_min()
    {
        while (lo < (int)head.size() && head[lo] < 0) ++lo;
        if (lo >= (int)head.size()) return -1;
        int best = head[lo];
        for (int i = next[best]; i >= 0; i = next[i]) if (i < best) best = i;
        remove(best);
        return best;
    }
};

// constrained approximate minimum degree; returns order[k] = unknown eliminated k-th
std::vector<int> amd_order(int nu, const std::vector<std::vector<int>> &adj, const std::vector<std::vector<int>> &before)
{
    std::vector<std::vector<int>> A(adj), E(nu), M(nu), after(nu);
    std::vector<char> state(nu, 0);          // 0 variable of the quotient graph, 1 element, 2 absorbed element
    std::vector<int> deg(nu), need(nu, 0), mark(nu, -1), wst(nu, -1), w(nu, 0), order, Lp;
    order.reserve(nu);
    for (int u = 0; u < nu; ++u) deg[u] = (int)A[u].size();
    if (!before.empty())
        for (int u = 0; u < nu; ++u) {
            need[u] = (int)before[u].size();
            if (getenv("SQPHIP_SYM_SLACK") && need[u] > 0) need[u] = std::max(1, need[u] - atoi(getenv("SQPHIP_SYM_SLACK")));   // experiment
            for (int v : before[u]) after[v].push_back(u);
        }
    Buckets B(nu);
    for (int u = 0; u < nu; ++u) if (need[u] == 0) B.insert(u, deg[u]);
    int nleft = nu;
    for (int k = 0; k < nu; ++k) {
        const int p = B.pop_min();
        if (p < 0) { fprintf(stderr, "sqphip: amd_order: no eligible vertex left (cyclic precedence)\n"); abort(); }
        order.push_back(p);
        // the new element: every live neighbour of p, directly or through an adjacent element (those are absorbed)
        Lp.clear();
        mark[p] = k;
        for (int v : A[p]) if (state[v] == 0 && mark[v] != k) { mark[v] = k; Lp.push_back(v); }
        for (int e : E[p]) {
            if (state[e] != 1) continue;
            for (int v : M[e]) if (mark[v] != k) { mark[v] = k; Lp.push_back(v); }
            state[e] = 2;
            std::vector<int>().swap(M[e]);
        }
        std::sort(Lp.begin(), Lp.end());
        state[p] = 1;
        std::vector<int>().swap(A[p]);
        std::vector<int>().swap(E[p]);
        --nleft;
        const int lp = (int)Lp.size();
        // w[e] = |Le \ Lp| for every element adjacent to a member of Lp
        for (int i : Lp) {
            auto &Ei = E[i];
            size_t o = 0;
            for (int e : Ei) {
                if (state[e] != 1) continue;
                Ei[o++] = e;
                if (wst[e] != k) { wst[e] = k; w[e] = (int)M[e].size(); }
                --w[e];
            }
            Ei.resize(o);
        }
        for (int i : Lp) {
            auto &Ai = A[i];
            size_t o = 0;
            for (int v : Ai) if (state[v] == 0 && mark[v] != k) Ai[o++] = v;     // covered by the new element otherwise
            Ai.resize(o);
            long d = (long)Ai.size() + (lp - 1);
            auto &Ei = E[i];
            o = 0;
            for (int e : Ei) {
                if (state[e] != 1) continue;
                if (w[e] == 0) { state[e] = 2; std::vector<int>().swap(M[e]); continue; }    // Le inside Lp: absorbed
                d += w[e];
                Ei[o++] = e;
            }
            Ei.resize(o);
            Ei.push_back(p);
            d = std::min<long>(d, (long)deg[i] + (lp - 1));
            d = std::min<long>(d, nleft - 1);
            if (d < 0) d = 0;
            deg[i] = (int)d;
            if (B.where[i] >= 0) { B.remove(i); B.insert(i, deg[i]); }
        }
        M[p] = Lp;
        for (int u : after[p]) if (state[u] == 0 && need[u] > 0 && --need[u] == 0) B.insert(u, std::min(deg[u], nleft > 0 ? nleft - 1 : 0));
    }
    return order;
}

// column structures of L (rows below the diagonal, positions, ascending) and the elimination tree for the order
// inv[position] = unknown
void column_structures(int nu, const std::vector<std::vector<int>> &adj, const std::vector<int> &pos,
                       const std::vector<int> &inv, std::vector<std::vector<int>> &st, std::vector<int> &parent)
{
    st.assign(nu, {});
    parent.assign(nu, -1);
    std::vector<std::vector<int>> kids(nu);
    std::vector<int> mark(nu, -1);
    for (int j = 0; j < nu; ++j) {
        auto &S = st[j];
        mark[j] = j;
        for (int v : adj[inv[j]]) { const int q = pos[v]; if (q > j && mark[q] != j) { mark[q] = j; S.push_back(q); } }
        for (int c : kids[j])
            for (int q : st[c]) if (mark[q] != j) { mark[q] = j; S.push_back(q); }
        std::sort(S.begin(), S.end());
        if (!S.empty()) { parent[j] = S[0]; kids[S[0]].push_back(j); }
    }
}

}  // namespace

SparseSym sparse_symbolic(int nu, const std::vector<std::vector<int>> &adj, const std::vector<std::vector<int>> &before,
                          const SymOptions &opt)
{
    SparseSym S;
    S.nu = nu;
    std::vector<int> inv(nu), pos(nu);
    if (opt.order_method == 1) std::iota(inv.begin(), inv.end(), 0);
    else inv = amd_order(nu, adj, opt.rows_after_vars ? before : std::vector<std::vector<int>>());
    for (int k = 0; k < nu; ++k) pos[inv[k]] = k;
    std::vector<std::vector<int>> st;
    std::vector<int> parent;
    column_structures(nu, adj, pos, inv, st, parent);
    {   // postorder of the elimination tree (children ascending): subtrees become contiguous
        std::vector<std::vector<int>> kids(nu);
        std::vector<int> roots, post, stack, it(nu, 0);
        for (int j = 0; j < nu; ++j) { if (parent[j] >= 0) kids[parent[j]].push_back(j); else roots.push_back(j); }
        post.reserve(nu);
        for (int r : roots) {
            stack.assign(1, r);
            while (!stack.empty()) {
                const int j = stack.back();
                if (it[j] < (int)kids[j].size()) stack.push_back(kids[j][it[j]++]);
                else { post.push_back(j); stack.pop_back(); }
            }
        }
        std::vector<int> inv2(nu);
        for (int k = 0; k < nu; ++k) inv2[k] = inv[post[k]];
        inv.swap(inv2);
        for (int k = 0; k < nu; ++k) pos[inv[k]] = k;
        column_structures(nu, adj, pos, inv, st, parent);
    }
    for (int j = 0; j < nu; ++j) { const double c = (double)st[j].size(); S.nnzL_exact += (long)c; S.flops_exact += c * c; }
    // maximal supernodes of the postordered tree
    std::vector<int> first, nc, nr, snp;     // per supernode: first column, columns, rows, parent supernode
    std::vector<int> c2s(nu);
    for (int j = 0; j < nu; ++j) {
        const bool chain = j > 0 && parent[j - 1] == j && st[j - 1].size() == st[j].size() + 1;
        if (!chain) { first.push_back(j); nc.push_back(0); }
        nc.back()++;
        c2s[j] = (int)first.size() - 1;
    }
    int ns0 = (int)first.size();
    nr.resize(ns0); snp.resize(ns0);
    for (int s = 0; s < ns0; ++s) {
        const int last = first[s] + nc[s] - 1;
        nr[s] = (int)st[last].size();
        snp[s] = parent[last] >= 0 ? c2s[parent[last]] : -1;
    }
    // relaxed amalgamation, bottom-up: group[s] = surviving supernode that holds s
    std::vector<int> group(ns0), gnc(nc);
    std::vector<double> zeros(ns0, 0.0);
    std::iota(group.begin(), group.end(), 0);
    {
        std::vector<std::vector<int>> kids(ns0);
        for (int s = 0; s < ns0; ++s) if (snp[s] >= 0) kids[snp[s]].push_back(s);
        for (int s = 0; s < ns0; ++s) {
            // children still standing on their own (with what they absorbed), cheapest first
            auto &K = kids[s];
            std::vector<int> cand(K);
            auto added = [&](int c) { return (double)gnc[c] * (double)(gnc[s] + nr[s] - nr[c]); };
            for (;;) {
                int best = -1; double bz = 0.0;
                for (int c : cand) { const double z = added(c); if (best < 0 || z < bz || (z == bz && c < best)) { best = c; bz = z; } }
                if (best < 0) break;
                const int c = best;
                const int mnc = gnc[s] + gnc[c], fs = mnc + nr[s];
                const double z = zeros[s] + zeros[c] + bz;
                const double lnz = 0.5 * mnc * (mnc + 1.0) + (double)mnc * nr[s];
                const bool ok = fs <= opt.small_front || z <= opt.zero_frac * lnz;
                cand.erase(std::find(cand.begin(), cand.end(), c));
                if (!ok) continue;
                // merge c into s: its standing children become children of s (and candidates in their own right)
                group[c] = s; gnc[s] = mnc; zeros[s] = z;
                for (int g : kids[c]) if (group[g] == g) { K.push_back(g); cand.push_back(g); }
            }
            // children list of s for its own parent's decision: only the standing ones
            std::vector<int> keep;
            for (int c : K) if (group[c] == c) keep.push_back(c);
            K.swap(keep);
        }
    }
    // Second pass, the spine of the tree: from each root downwards the deepest child is merged into its parent while the
    // merged front stays within `chain_front` rows, whatever the zeros cost.  The fronts along the spine are eliminated
    // and solved one after the other by construction (each waits for its child): fewer, larger fronts there trade flops
    // that run in parallel for per-front latency that does not.
    const int chain_front = getenv("SQPHIP_SYM_CHAIN") ? atoi(getenv("SQPHIP_SYM_CHAIN")) : opt.chain_front;   // read per call
    if (chain_front > 0) {
        std::vector<std::vector<int>> kids(ns0);
        std::vector<int> height(ns0, 0), par(ns0, -1);
        auto find0 = [&](int s) { while (group[s] != s) s = group[s]; return s; };
        for (int s = 0; s < ns0; ++s)
            if (group[s] == s && snp[s] >= 0) { par[s] = find0(snp[s]); kids[par[s]].push_back(s); }
        for (int s = 0; s < ns0; ++s)                       // postorder: children before parents
            if (group[s] == s && par[s] >= 0) height[par[s]] = std::max(height[par[s]], height[s] + 1);
        for (int r = ns0 - 1; r >= 0; --r) {
            if (group[r] != r || par[r] >= 0) continue;      // roots only
            int s = r;
            for (;;) {
                int c = -1;
                for (int k : kids[s]) if (c < 0 || height[k] > height[c] || (height[k] == height[c] && k > c)) c = k;
                if (c < 0 || height[c] == 0) break;          // the leaves stay as they are
                if (gnc[s] + gnc[c] + nr[s] <= chain_front) {
                    group[c] = s; gnc[s] += gnc[c];
                    kids[s].erase(std::find(kids[s].begin(), kids[s].end(), c));
                    for (int g : kids[c]) { par[g] = s; kids[s].push_back(g); }
                    height[s] = 0;
                    for (int g : kids[s]) height[s] = std::max(height[s], height[g] + 1);
                } else s = c;
            }
        }
    }
    auto find = [&](int s) { while (group[s] != s) s = group[s]; return s; };
    // final order: subtrees of the standing children first, then the columns of the group in their old order
    std::vector<int> inv3;
    inv3.reserve(nu);
    std::vector<int> gfirst, gncols;
    {
        std::vector<std::vector<int>> members(ns0), gkids(ns0);
        std::vector<int> roots;
        for (int s = 0; s < ns0; ++s) members[find(s)].push_back(s);
        for (int s = 0; s < ns0; ++s) {
            if (group[s] != s) continue;
            // parent group: the group of the etree parent of the group's top column (= of s itself, the top member)
            const int pg = snp[s] >= 0 ? find(snp[s]) : -1;
            if (pg >= 0) gkids[pg].push_back(s); else roots.push_back(s);
        }
        std::vector<int> stack, it(ns0, 0);
        for (int r : roots) {
            stack.assign(1, r);
            while (!stack.empty()) {
                const int g = stack.back();
                if (it[g] < (int)gkids[g].size()) { stack.push_back(gkids[g][it[g]++]); continue; }
                gfirst.push_back((int)inv3.size());
                for (int s : members[g]) for (int j = first[s]; j < first[s] + nc[s]; ++j) inv3.push_back(inv[j]);
                gncols.push_back((int)inv3.size() - gfirst.back());
                stack.pop_back();
            }
        }
    }
    inv.swap(inv3);
    for (int k = 0; k < nu; ++k) pos[inv[k]] = k;
    column_structures(nu, adj, pos, inv, st, parent);
    S.pos = pos; S.inv = inv;
    S.ns = (int)gfirst.size();
    S.sn_first = gfirst; S.sn_nc = gncols;
    S.sn_nr.resize(S.ns); S.sn_rowptr.assign(S.ns + 1, 0); S.sn_parent.assign(S.ns, -1);
    S.col2sn.resize(nu);
    for (int s = 0; s < S.ns; ++s)
        for (int j = gfirst[s]; j < gfirst[s] + gncols[s]; ++j) S.col2sn[j] = s;
    for (int s = 0; s < S.ns; ++s) {
        const int last = gfirst[s] + gncols[s] - 1;
        const auto &R = st[last];
        S.sn_nr[s] = (int)R.size();
        S.sn_rowptr[s + 1] = S.sn_rowptr[s] + (int)R.size();
        S.sn_rows.insert(S.sn_rows.end(), R.begin(), R.end());
        if (!R.empty()) S.sn_parent[s] = S.col2sn[R[0]];
        // every column of the group must fit the front: structure inside [columns of the group | R]
        for (int j = gfirst[s]; j < last; ++j)
            for (int q : st[j])
                if (q > last && !std::binary_search(R.begin(), R.end(), q)) {
                    fprintf(stderr, "sqphip: sparse_symbolic: column %d of supernode %d leaves its front\n", j, s);
                    abort();
                }
    }
    // children, relative indices, levels, front offsets
    S.child_ptr.assign(S.ns + 1, 0);
    for (int s = 0; s < S.ns; ++s) if (S.sn_parent[s] >= 0) S.child_ptr[S.sn_parent[s] + 1]++;
    for (int s = 0; s < S.ns; ++s) S.child_ptr[s + 1] += S.child_ptr[s];
    S.child.resize(S.child_ptr[S.ns]);
    {
        std::vector<int> fill(S.child_ptr.begin(), S.child_ptr.end() - 1);
        for (int s = 0; s < S.ns; ++s) if (S.sn_parent[s] >= 0) S.child[fill[S.sn_parent[s]]++] = s;
    }
    S.rel.resize(S.sn_rows.size());
    S.sn_level.assign(S.ns, 0);
    S.front_off.resize(S.ns);
    for (int s = 0; s < S.ns; ++s) {
        const int p = S.sn_parent[s];
        if (p >= 0) {
            if (p <= s) { fprintf(stderr, "sqphip: sparse_symbolic: supernodes not in postorder\n"); abort(); }
            const int pf = S.sn_first[p], pnc = S.sn_nc[p];
            const int *PR = S.sn_rows.data() + S.sn_rowptr[p];
            const int pnr = S.sn_nr[p];
            for (int k = S.sn_rowptr[s]; k < S.sn_rowptr[s + 1]; ++k) {
                const int q = S.sn_rows[k];
                if (q < pf + pnc) S.rel[k] = q - pf;
                else {
                    const int *it = std::lower_bound(PR, PR + pnr, q);
                    if (it == PR + pnr || *it != q) { fprintf(stderr, "sqphip: sparse_symbolic: child row outside the parent front\n"); abort(); }
                    S.rel[k] = pnc + (int)(it - PR);
                }
            }
            S.sn_level[p] = std::max(S.sn_level[p], S.sn_level[s] + 1);
        }
        const long fs = S.sn_nc[s] + S.sn_nr[s];
        S.front_off[s] = S.front_total;
        S.front_total += (fs * fs + 1) / 2 * 2;           // keep every front 16-byte aligned
        S.max_front = std::max(S.max_front, (int)fs);
        S.max_nc = std::max(S.max_nc, S.sn_nc[s]);
        const double c = S.sn_nc[s], r = S.sn_nr[s];
        S.nnzL += (long)(0.5 * c * (c - 1.0) + c * r);
        for (int k = 0; k < S.sn_nc[s]; ++k) { const double t = fs - k - 1; S.flops += t * (t + 1.0) + t; }
    }
    for (int s = 0; s < S.ns; ++s) S.nlevels = std::max(S.nlevels, S.sn_level[s] + 1);
    S.level_ptr.assign(S.nlevels + 1, 0);
    for (int s = 0; s < S.ns; ++s) S.level_ptr[S.sn_level[s] + 1]++;
    for (int l = 0; l < S.nlevels; ++l) S.level_ptr[l + 1] += S.level_ptr[l];
    S.level_sn.resize(S.ns);
    {
        std::vector<int> fill(S.level_ptr.begin(), S.level_ptr.end() - 1);
        for (int s = 0; s < S.ns; ++s) S.level_sn[fill[S.sn_level[s]]++] = s;
    }
    if (getenv("SQPHIP_SYM_DUMP"))          // fronts by level: columns x rows(children)
        for (int l = 0; l < S.nlevels; ++l) {
            fprintf(stderr, "level %2d:", l);
            for (int k = S.level_ptr[l]; k < S.level_ptr[l + 1]; ++k) {
                const int s = S.level_sn[k];
                fprintf(stderr, " %dx%d(%d)", S.sn_nc[s], S.sn_nr[s], S.child_ptr[s + 1] - S.child_ptr[s]);
            }
            fprintf(stderr, "\n");
        }
    return S;
}

void kkt_graph(int n, int m, const std::vector<int> &kpos, int mk, const std::vector<int> &hcolptr,
               const std::vector<int> &hrowval, const std::vector<int> &jrowptr, const std::vector<int> &jrcol,
               std::vector<std::vector<int>> &adj, std::vector<std::vector<int>> &before)
{
    const int nu = n + mk;
    adj.assign(nu, {});
    before.assign(nu, {});
    auto edge = [&](int a, int b) { if (a != b) { adj[a].push_back(b); adj[b].push_back(a); } };
    for (int j = 0; j < n; ++j)
        for (int k = hcolptr[j]; k < hcolptr[j + 1]; ++k) if (hrowval[k] > j) edge(hrowval[k], j);
    for (int i = 0; i < m; ++i) {
        const int s = jrowptr[i], e = jrowptr[i + 1];
        if (kpos[i] >= 0) {
            for (int t = s; t < e; ++t) { edge(n + kpos[i], jrcol[t]); before[n + kpos[i]].push_back(jrcol[t]); }
        } else {
            for (int a = s; a < e; ++a) for (int b = a + 1; b < e; ++b) edge(jrcol[a], jrcol[b]);
        }
    }
    for (auto &l : adj) { std::sort(l.begin(), l.end()); l.erase(std::unique(l.begin(), l.end()), l.end()); }
    for (auto &l : before) { std::sort(l.begin(), l.end()); l.erase(std::unique(l.begin(), l.end()), l.end()); }
}

}  // namespace sqphip

// C-ABI: pure host computation (no GPU needed), 1-based COO structures as in sqphip_create.  See include/sqphip.h.
extern "C" int sqphip_kkt_symbolic(int64_t n, int64_t m, int64_t nnzJ, const int64_t *jrow, const int64_t *jcol,
                                   int64_t nnzH, const int64_t *hrow, const int64_t *hcol, const double *gL,
                                   const double *gU, int32_t condense, int32_t rows_after_vars, int32_t small_front,
                                   double zero_frac, int32_t *pos, sqphip_symbolic_stats *out)
{
    if (n <= 0 || m < 0 || (m > 0 && (!gL || !gU)) || (nnzJ > 0 && (!jrow || !jcol)) || (nnzH > 0 && (!hrow || !hcol)))
        return SQPHIP_EINVAL;
    std::vector<int> kpos(m > 0 ? m : 1, -1);
    std::vector<std::vector<int>> hc(n), jr(m);
    for (int64_t k = 0; k < nnzH; ++k) {
        const int r = (int)hrow[k] - 1, c = (int)hcol[k] - 1;
        if (r < 0 || r >= n || c < 0 || c >= n) return SQPHIP_EINVAL;
        hc[c].push_back(r); if (r != c) hc[r].push_back(c);
    }
    for (int64_t k = 0; k < nnzJ; ++k) {
        const int r = (int)jrow[k] - 1, c = (int)jcol[k] - 1;
        if (r < 0 || r >= m || c < 0 || c >= n) return SQPHIP_EINVAL;
        jr[r].push_back(c);
    }
    std::vector<int> hcolptr(n + 1, 0), hrowval, jrowptr(m + 1, 0), jrcol;
    for (int j = 0; j < n; ++j) {
        std::sort(hc[j].begin(), hc[j].end()); hc[j].erase(std::unique(hc[j].begin(), hc[j].end()), hc[j].end());
        hrowval.insert(hrowval.end(), hc[j].begin(), hc[j].end()); hcolptr[j + 1] = (int)hrowval.size();
    }
    for (int i = 0; i < m; ++i) {
        std::sort(jr[i].begin(), jr[i].end()); jr[i].erase(std::unique(jr[i].begin(), jr[i].end()), jr[i].end());
        jrcol.insert(jrcol.end(), jr[i].begin(), jr[i].end()); jrowptr[i + 1] = (int)jrcol.size();
    }
    int mk = 0;
    for (int64_t i = 0; i < m; ++i)
        if (!condense || sqphip::kkt_row_is_kept(gL[i], gU[i], jrowptr[i + 1] - jrowptr[i])) kpos[i] = mk++;
    std::vector<std::vector<int>> adj, before;
    sqphip::kkt_graph((int)n, (int)m, kpos, mk, hcolptr, hrowval, jrowptr, jrcol, adj, before);
    sqphip::SymOptions so;
    so.rows_after_vars = rows_after_vars;
    if (small_front > 0) so.small_front = small_front;
    if (zero_frac >= 0.0) so.zero_frac = zero_frac;
    sqphip::SparseSym S = sqphip::sparse_symbolic((int)n + mk, adj, before, so);
    if (pos) for (int u = 0; u < S.nu; ++u) pos[u] = S.pos[u];
    if (out) {
        out->order = S.nu; out->n_supernodes = S.ns; out->n_levels = S.nlevels; out->max_front = S.max_front;
        out->max_cols = S.max_nc; out->nnz_l = S.nnzL; out->nnz_l_exact = S.nnzL_exact; out->flops = S.flops;
        out->flops_exact = S.flops_exact; out->front_doubles = S.front_total;
        long nk = 0;
        for (auto &l : adj) nk += (long)l.size();
        out->nnz_k_lower = nk / 2 + S.nu;
    }
    return SQPHIP_OK;
}
