// order.hip -- host-only: ordering of the condensed Newton matrix that exposes its tile sparsity.
//
// The condensed matrix K_c (ipm.hip, k_kkt_assemble) of a structured NLP is very sparse: in an ACOPF instance
// everything but the bus voltages hangs off them in small pieces.  This file finds, from the PATTERN alone, a vertex
// separator R such that the rest of the unknowns S falls into connected components of at most 64 unknowns, and packs
// those components into 64-slot tiles.  Ordered [tiles of S | R], the leading Ts x Ts tile block of K_c is block
// diagonal, and stays so during the elimination (independent components create no fill between each other), so the
// factorisation can treat the Ts leading tile columns as independent (ldlt.hip, ldlt_factor) and only the last
// |R| unknowns as a dense matrix.  Any choice is numerically admissible: K_c is quasi-definite, an LDL^T without
// pivoting exists under every symmetric permutation (Vanderbei 1995) and the inertia test (n positive pivots) does
// not depend on the order -- the quality of the heuristic only decides how much work is saved.
//
// Heuristic: while some component of the graph of S (variables only: rows_last) has more than 32 vertices, move its
// vertex of highest degree (ties: lowest index) to R; then let every kept row whose variables all ended up in S
// join their cluster if the merged cluster still fits a tile.  Components sorted by size (ties: lowest vertex) are packed first-fit into tiles;
// inside a tile variables come before rows.
#include "sqphip_internal.hpp"
#include "sparse.hpp"
#include "../../include/sqphip.h"
#include <algorithm>
#include <numeric>
#include <vector>

namespace sqphip {

// unknown u: variable j (u = j < n) or kept row (u = n + kpos).  adj: symmetric adjacency lists (sorted, unique)
// long_rows: variable lists of the eliminated rows whose cliques were only chained in adj (see kkt_order)
KktOrder kkt_order_from_graph(int n, int nc, const std::vector<std::vector<int>> &adj, bool rows_last,
                              const std::vector<std::vector<int>> &long_rows)
{
    std::vector<char> inS(nc, 1);
    // largest cluster of variables in the first pass: half a tile, so that the rows joining in the second pass (which
    // may fill a cluster up to 64) find room.  Measured on the IEEE-118 shape: 64 / 40 / 32 / 24 / 16 -> dense
    // remainder 725 / 685 / 673 / 652 / 671 and 478 / 511 / 532 / 502 / 505 QP/s (SQPHIP_ORDER_VARCAP overrides).
    const int varcap = getenv("SQPHIP_ORDER_VARCAP") ? std::max(1, std::min(64, atoi(getenv("SQPHIP_ORDER_VARCAP")))) : 32;
    if (rows_last) for (int u = n; u < nc; ++u) inS[u] = 0;     // every kept row goes to the dense remainder
    std::vector<int> comp(nc), deg(nc), stack;
    std::vector<std::vector<int>> comps;
    for (;;) {
        // components of the graph induced on S
        std::fill(comp.begin(), comp.end(), -1);
        comps.clear();
        for (int s = 0; s < nc; ++s) {
            if (!inS[s] || comp[s] >= 0) continue;
            const int id = (int)comps.size();
            comps.emplace_back();
            stack.assign(1, s); comp[s] = id;
            while (!stack.empty()) {
                const int u = stack.back(); stack.pop_back();
                comps[id].push_back(u);
                for (int v : adj[u]) if (inS[v] && comp[v] < 0) { comp[v] = id; stack.push_back(v); }
            }
        }
        bool changed = false;
        for (auto &c : comps) {
            if ((int)c.size() <= varcap) continue;
            int best = -1, bestdeg = -1;
            for (int u : c) {
                int d = 0;
                for (int v : adj[u]) d += inS[v];
                if (d > bestdeg || (d == bestdeg && u < best)) { best = u; bestdeg = d; }
            }
            inS[best] = 0; changed = true;
        }
        if (!changed) break;
    }
    if (rows_last) {
        // second pass: a kept row may follow its variables into their tile when ALL of them are in S and the
        // clusters it ties together still fit one tile.  Inside a tile variables come before rows, so such a row is
        // pivoted after every variable it couples to -- the property that makes the variables-first order safe --
        // and it couples to nothing else but its own diagonal.
        std::vector<int> root(comps.size()), csize(comps.size());
        std::iota(root.begin(), root.end(), 0);
        for (size_t c = 0; c < comps.size(); ++c) csize[c] = (int)comps[c].size();
        auto find = [&](int a) { while (root[a] != a) { root[a] = root[root[a]]; a = root[a]; } return a; };
        std::vector<int> rowroot(nc, -1), seen;
        for (int u = n; u < nc; ++u) {
            bool ok = !adj[u].empty();
            seen.clear();
            int tot = 1;
            for (int v : adj[u]) {
                if (!inS[v]) { ok = false; break; }
                const int r = find(comp[v]);
                if (std::find(seen.begin(), seen.end(), r) == seen.end()) { seen.push_back(r); tot += csize[r]; }
            }
            if (!ok || tot > 64) continue;
            const int r0 = seen[0];
            for (size_t k = 1; k < seen.size(); ++k) { root[seen[k]] = r0; }
            csize[r0] = tot;
            rowroot[u] = r0; inS[u] = 1;
        }
        // rebuild the component lists from the merged clusters
        std::vector<std::vector<int>> merged(comps.size());
        for (size_t c = 0; c < comps.size(); ++c) {
            auto &dst = merged[find((int)c)];
            dst.insert(dst.end(), comps[c].begin(), comps[c].end());
        }
        for (int u = n; u < nc; ++u) if (rowroot[u] >= 0) merged[find(rowroot[u])].push_back(u);
        comps.clear();
        for (auto &c : merged) if (!c.empty()) comps.push_back(std::move(c));
    }
    for (auto &c : comps) std::sort(c.begin(), c.end());
    std::sort(comps.begin(), comps.end(), [](const std::vector<int> &a, const std::vector<int> &b) {
        return a.size() != b.size() ? a.size() > b.size() : a[0] < b[0]; });
    // first-fit packing into 64-slot tiles
    // packing: clusters in index order (neighbours in the numbering are neighbours in the network, so a row of the
    // remainder finds the clusters it touches in few tiles and the tile mask gets sparse), first-fit among the last
    // `window` tiles.  Measured on the IEEE-118 shape: window 1 / 3 / 5 / all -> 25 / 24 / 23 / 23 tiles, 192 / 186 /
    // 196 / 211 (pair, leading tile) products, 736 / 755 / 771 / 763 QP/s; largest-first first-fit (SQPHIP_ORDER_PACK=0)
    // packs 22 tiles but leaves 336 products: 731 QP/s.
    std::vector<std::vector<int>> bins;
    const int window = getenv("SQPHIP_ORDER_PACK") ? atoi(getenv("SQPHIP_ORDER_PACK")) : 5;
    if (window > 0) {
        std::sort(comps.begin(), comps.end(), [](const std::vector<int> &a, const std::vector<int> &b) { return a[0] < b[0]; });
        for (auto &c : comps) {
            size_t b = bins.size() > (size_t)window ? bins.size() - window : 0;
            while (b < bins.size() && bins[b].size() + c.size() > 64) ++b;
            if (b == bins.size()) bins.emplace_back();
            bins[b].insert(bins[b].end(), c.begin(), c.end());
        }
    } else
        for (auto &c : comps) {
            size_t b = 0;
            while (b < bins.size() && bins[b].size() + c.size() > 64) ++b;
            if (b == bins.size()) bins.emplace_back();
            bins[b].insert(bins[b].end(), c.begin(), c.end());
        }
    KktOrder o;
    o.pos.assign(nc, -1);
    o.Ts = (int)bins.size();
    for (size_t b = 0; b < bins.size(); ++b) {
        std::sort(bins[b].begin(), bins[b].end());        // variables (u < n) first, then rows
        for (size_t k = 0; k < bins[b].size(); ++k) o.pos[bins[b][k]] = (int)(64 * b + k);
    }
    // remainder: separator variables first (a row must stay behind every variable it couples to), then the rows
    // sorted by the leading tiles they touch, so that a 64-row tile of the remainder couples to few leading tiles and
    // most (remainder tile, leading tile) blocks of the panel are structurally zero.  tmask / pair lists: what the
    // panel solve and the rank-64 Ts update of ldlt_factor may skip (exact: a zero block of K_c stays zero through
    // the elimination of the independent leading tiles).
    std::vector<int> rem;
    for (int u = 0; u < nc; ++u) if (!inS[u]) rem.push_back(u);
    std::vector<std::vector<int>> touch(nc);
    for (const auto &lr : long_rows) {           // a chained clique: every pair of its variables is an entry of K_c
        std::vector<int> tiles;
        for (int v : lr) if (inS[v]) tiles.push_back(o.pos[v] / 64);
        for (int u : lr) if (!inS[u]) touch[u].insert(touch[u].end(), tiles.begin(), tiles.end());
    }
    for (int u : rem) {
        for (int v : adj[u]) if (inS[v]) touch[u].push_back(o.pos[v] / 64);
        std::sort(touch[u].begin(), touch[u].end());
        touch[u].erase(std::unique(touch[u].begin(), touch[u].end()), touch[u].end());
    }
    if (rows_last && !getenv("SQPHIP_ORDER_NO_SORT"))
        std::stable_sort(rem.begin(), rem.end(), [&](int a, int b) {
            const bool va = a < n, vb = b < n;
            if (va != vb) return va;                       // variables before rows
            if (va) return a < b;
            const bool ea = touch[a].empty(), eb = touch[b].empty();
            if (ea != eb) return eb;                       // rows touching no leading tile last
            return touch[a] < touch[b] || (touch[a] == touch[b] && a < b);
        });
    int p = 64 * o.Ts;
    for (int u : rem) o.pos[u] = p++;
    o.Nf = p;
    o.Tr = ((int)rem.size() + 63) / 64;
    o.tmask.assign((size_t)o.Tr * std::max(1, o.Ts), 0);
    for (size_t k = 0; k < rem.size(); ++k)
        for (int t : touch[rem[k]]) o.tmask[(k / 64) * o.Ts + t] = 1;
    o.pair_ptr.assign(1, 0);
    for (int ti = 0; ti < o.Tr; ++ti)
        for (int tj = 0; tj <= ti; ++tj) {
            for (int k = 0; k < o.Ts; ++k)
                if (o.tmask[(size_t)ti * o.Ts + k] && o.tmask[(size_t)tj * o.Ts + k]) o.pair_k.push_back(k);
            o.pair_ptr.push_back((int)o.pair_k.size());
        }
    return o;
}

// graph of K_c from the NLP structure: H (full symmetric CSC), J (CSR), kept rows (kpos >= 0) as vertices,
// eliminated rows as cliques among their variables (J_I' D^-1 J_I)
KktOrder kkt_order(int n, int m, const std::vector<int> &kpos, int mk, const std::vector<int> &hcolptr,
                   const std::vector<int> &hrowval, const std::vector<int> &jrowptr, const std::vector<int> &jrcol,
                   bool rows_last)
{
    const int nc = n + mk;
    std::vector<std::vector<int>> adj(nc), long_rows;
    auto edge = [&](int a, int b) { if (a != b) { adj[a].push_back(b); adj[b].push_back(a); } };
    for (int j = 0; j < n; ++j)
        for (int k = hcolptr[j]; k < hcolptr[j + 1]; ++k) if (hrowval[k] > j) edge(hrowval[k], j);
    for (int i = 0; i < m; ++i) {
        const int s = jrowptr[i], e = jrowptr[i + 1];
        if (kpos[i] >= 0) { for (int t = s; t < e; ++t) edge(n + kpos[i], jrcol[t]); }
        // an eliminated row is a clique among its variables (rows too long for that stay in the matrix:
        // kkt_row_is_kept, sparse.hpp -- a chain stood in for the clique of such rows until round 2 and split
        // coupled variables over different "independent" tiles)
        else { for (int a = s; a < e; ++a) for (int b = a + 1; b < e; ++b) edge(jrcol[a], jrcol[b]); }
    }
    for (auto &l : adj) { std::sort(l.begin(), l.end()); l.erase(std::unique(l.begin(), l.end()), l.end()); }
    KktOrder o = kkt_order_from_graph(n, nc, adj, rows_last, long_rows);
    // the factorisation treats the leading tiles as mutually independent: make sure they are.  Any structural entry
    // between two DIFFERENT leading tiles would be assembled into a block nobody zeroes or factorises;
    // the construction above cannot produce one, but a wrong answer is not an acceptable failure mode for a
    // heuristic -- fall back to the plain order (no leading tiles) if it ever does.
    const int lead_end = 64 * o.Ts;
    for (int u = 0; u < nc && o.Ts > 0; ++u) {
        if (o.pos[u] >= lead_end) continue;
        for (int v : adj[u])
            if (o.pos[v] < lead_end && o.pos[v] / 64 != o.pos[u] / 64) {
                fprintf(stderr, "sqphip: kkt_order: unknowns %d and %d are coupled across leading tiles; using the plain order\n", u, v);
                KktOrder plain;
                plain.pos.resize(nc);
                std::iota(plain.pos.begin(), plain.pos.end(), 0);
                plain.Nf = nc; plain.Tr = (nc + 63) / 64;
                plain.pair_ptr.assign(1, 0);
                return plain;
            }
    }
    return o;
}

}  // namespace sqphip

// C-ABI: pure host computation (no GPU needed), 1-based COO structures as in sqphip_create
extern "C" int sqphip_kkt_order(int64_t n, int64_t m, int64_t nnzJ, const int64_t *jrow, const int64_t *jcol,
                                int64_t nnzH, const int64_t *hrow, const int64_t *hcol, const double *gL,
                                const double *gU, int32_t rows_last, int32_t *pos, int32_t *n_lead_tiles,
                                int32_t *order_out)
{
    if (n <= 0 || m < 0 || !pos) return SQPHIP_EINVAL;
    std::vector<int> kpos(m > 0 ? m : 1, -1);
    int mk = 0;
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
    for (int64_t i = 0; i < m; ++i)
        if (sqphip::kkt_row_is_kept(gL[i], gU[i], jrowptr[i + 1] - jrowptr[i])) kpos[i] = mk++;
    sqphip::KktOrder o = sqphip::kkt_order((int)n, (int)m, kpos, mk, hcolptr, hrowval, jrowptr, jrcol, rows_last != 0);
    for (int u = 0; u < (int)n + mk; ++u) pos[u] = o.pos[u];
    if (n_lead_tiles) *n_lead_tiles = o.Ts;
    if (order_out) *order_out = o.Nf;
    return SQPHIP_OK;
}
