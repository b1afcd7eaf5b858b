// mfplan.hip -- host-only: the multifrontal plan of the sparse Newton matrix (assembly lists, front layout, launch
// schedule) and a host reference of the numeric phase that the CPU tests use to validate the plan.
//
// Seat in the reference: Ipopt's linear-solver interface (symbolic + numeric factorisation, MUMPS / MA57;
// /root/reference/examples/acopf/opf.jl:59-64), reached through /root/reference/src/algorithms/subproblem_JuMP.jl:178.
// The matrix is the (condensed) Newton matrix of ipm.hip:
//     [ W + J_I' (D_I + reg)^-1 J_I    J_K' ]     W = hsc H + hd + sigp + (delta_w + reg) I,  K = rows kept in the matrix
//     [ J_K                       -(D_K + reg) ]
#include "sparse.hpp"
#include "../../include/sqphip.h"
#include "../../include/sqphip_test_hooks.h"
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <tuple>

namespace sqphip {

MfPlan mf_build_plan(int n, int m, const std::vector<int> &kpos, int mk, const std::vector<int> &hcolptr,
                     const std::vector<int> &hrowval, const std::vector<int> &jrowptr, const std::vector<int> &jrcol,
                     const std::vector<int> &jrslot, const SymOptions &opt)
{
    MfPlan P;
    std::vector<std::vector<int>> adj, before;
    kkt_graph(n, m, kpos, mk, hcolptr, hrowval, jrowptr, jrcol, adj, before);
    P.S = sparse_symbolic(n + mk, adj, before, opt);
    const SparseSym &S = P.S;
    // front layout: (fs + 1) x fs, 16-byte aligned
    P.off.resize(S.ns);
    for (int s = 0; s < S.ns; ++s) {
        const long fs = S.sn_nc[s] + S.sn_nr[s];
        P.off[s] = P.stride;
        P.stride += ((fs + 1) * fs + 1) / 2 * 2;
    }
    // assembly lists.  Entry (unknown a, unknown b) of the lower triangle lives in the front of the supernode that
    // owns the earlier of the two positions.
    struct Rec { int sn, loc, seq; MfItem it; };
    std::vector<Rec> recs;
    auto add = [&](int ua, int ub, MfItem it) {
        int pa = S.pos[ua], pb = S.pos[ub];
        if (pa < pb) std::swap(pa, pb);              // pa = row position, pb = column position
        const int s = S.col2sn[pb], f = S.sn_first[s], nc = S.sn_nc[s], fs = nc + S.sn_nr[s];
        int li;
        if (pa < f + nc) li = pa - f;
        else {
            const int *R = S.sn_rows.data() + S.sn_rowptr[s];
            const int *q = std::lower_bound(R, R + S.sn_nr[s], pa);
            if (q == R + S.sn_nr[s] || *q != pa) { fprintf(stderr, "sqphip: mf_build_plan: entry outside its front\n"); abort(); }
            li = nc + (int)(q - R);
        }
        recs.push_back({s, (pb - f) * (fs + 1) + li, (int)recs.size(), it});
    };
    for (int j = 0; j < n; ++j) {
        add(j, j, {MF_ITEM_VDIAG, -1, j, 0});
        for (int k = hcolptr[j]; k < hcolptr[j + 1]; ++k) {
            const int i = hrowval[k];
            if (i >= j) add(i, j, {MF_ITEM_H, -1, k, 0});      // the mirrored slot (j, i) of the full pattern holds the same value
        }
    }
    for (int i = 0; i < m; ++i) {
        const int s = jrowptr[i], e = jrowptr[i + 1];
        if (kpos[i] >= 0) {
            add(n + kpos[i], n + kpos[i], {MF_ITEM_RDIAG, i, 0, 0});
            for (int t = s; t < e; ++t) add(n + kpos[i], jrcol[t], {MF_ITEM_JKEPT, i, jrslot[t], 0});
        } else {
            for (int a = s; a < e; ++a)
                for (int b = s; b <= a; ++b) add(jrcol[a], jrcol[b], {MF_ITEM_PAIR, i, jrslot[a], jrslot[b]});
        }
    }
    std::sort(recs.begin(), recs.end(), [](const Rec &x, const Rec &y) {
        return std::tie(x.sn, x.loc, x.seq) < std::tie(y.sn, y.loc, y.seq); });
    P.asm_ptr.assign(S.ns + 1, 0);
    for (size_t r = 0; r < recs.size(); ++r) {
        const bool fresh = r == 0 || recs[r].sn != recs[r - 1].sn || recs[r].loc != recs[r - 1].loc;
        if (fresh) {
            P.dest_loc.push_back(recs[r].loc);
            P.item_ptr.push_back((int)P.items.size());
            P.asm_ptr[recs[r].sn + 1]++;
        }
        P.items.push_back(recs[r].it);
    }
    P.item_ptr.push_back((int)P.items.size());
    for (int s = 0; s < S.ns; ++s) P.asm_ptr[s + 1] += P.asm_ptr[s];
    P.nnzK = (long)P.dest_loc.size();
    P.dest_rc.resize(P.dest_loc.size());
    for (int s = 0; s < S.ns; ++s) {
        const int ld = S.sn_nc[s] + S.sn_nr[s] + 1;
        for (int e = P.asm_ptr[s]; e < P.asm_ptr[s + 1]; ++e) P.dest_rc[e] = (P.dest_loc[e] % ld) | ((P.dest_loc[e] / ld) << 16);
    }
    if (P.stride >= (1L << 31)) { fprintf(stderr, "sqphip: mf_build_plan: front arena too large for 32-bit offsets\n"); abort(); }
    // The narrow top of the assembly tree (solves: k_mf_solve_top2; factorisation: k_mf_spine): from the first level on above
    // which no level holds more than two fronts (or up to four wave-sized ones) -- decided here, before the gather lists,
    // because the lists of a top front take its children in the order [children below the top, ascending | children inside
    // the top, ascending]: the last child inside the top is then the last term of every sum it contributes to, which lets
    // the spine kernel add it straight from its accumulator registers behind the terms that come from the arena, in the
    // order every other kernel (level launches, host reference) uses too.
    const bool big_solve = !(getenv("SQPHIP_MF_BIG_SOLVE") && atoi(getenv("SQPHIP_MF_BIG_SOLVE")) == 0);   // experiment switch
    P.top_level = S.nlevels;
    if (big_solve && !(getenv("SQPHIP_MF_TOP") && atoi(getenv("SQPHIP_MF_TOP")) == 0)) {
        // ... a level of up to four fronts joins the top as well when all of them are wave-sized (<= 64 rows): the four
        // waves of the instance's workgroup take one each, exactly what a level launch would do, one launch less per pass
        auto joins = [&](int lev) {
            const int cnt = S.level_ptr[lev + 1] - S.level_ptr[lev];
            if (cnt <= 2) return true;
            if (cnt > 4 || getenv("SQPHIP_MF_TOP_NARROW")) return false;
            for (int q = S.level_ptr[lev]; q < S.level_ptr[lev + 1]; ++q)
                if (S.sn_nc[S.level_sn[q]] + S.sn_nr[S.level_sn[q]] > 64) return false;
            return true;
        };
        int l = S.nlevels;
        while (l > 0 && joins(l - 1)) --l;
        if (S.nlevels - l >= 2) P.top_level = l;
    }
    P.spine_level = P.top_level; P.narrow_level = P.top_level;
    // children of s in summation order
    auto children_of = [&](int s) {
        std::vector<int> ch(S.child.begin() + S.child_ptr[s], S.child.begin() + S.child_ptr[s + 1]);
        std::stable_sort(ch.begin(), ch.end(), [&](int a, int b) { return (S.sn_level[a] >= P.spine_level) < (S.sn_level[b] >= P.spine_level); });
        return ch;
    };
    // extend-add gather lists
    P.ea_ptr.assign(S.ns + 1, 0);
    P.ea_src_ptr.push_back(0);
    {
        std::vector<std::tuple<int, int, int>> con;      // (column, row, source offset), children ascending
        for (int s = 0; s < S.ns; ++s) {
            const int fs = S.sn_nc[s] + S.sn_nr[s];
            con.clear();
            for (int c : children_of(s)) {
                const int cnc = S.sn_nc[c], cnr = S.sn_nr[c], cld = cnc + cnr + 1;
                const int *rel = S.rel.data() + S.sn_rowptr[c];
                for (int jj = 0; jj < cnr; ++jj)
                    for (int ii = jj; ii <= cnr; ++ii)
                        con.emplace_back(rel[jj], ii < cnr ? rel[ii] : fs, (int)(P.off[c] + (long)(cnc + jj) * cld + cnc + ii));
            }
            std::stable_sort(con.begin(), con.end(), [](const std::tuple<int, int, int> &a, const std::tuple<int, int, int> &b) {
                return std::tie(std::get<0>(a), std::get<1>(a)) < std::tie(std::get<0>(b), std::get<1>(b)); });
            for (size_t r = 0; r < con.size(); ++r) {
                const bool fresh = r == 0 || std::get<0>(con[r]) != std::get<0>(con[r - 1]) || std::get<1>(con[r]) != std::get<1>(con[r - 1]);
                if (fresh) {
                    if (r) P.ea_src_ptr.push_back((int)P.ea_src.size());
                    P.ea_rc.push_back(std::get<1>(con[r]) | (std::get<0>(con[r]) << 16));
                    P.ea_ptr[s + 1]++;
                }
                P.ea_src.push_back(std::get<2>(con[r]));
            }
            if (!con.empty()) P.ea_src_ptr.push_back((int)P.ea_src.size());
        }
        for (int s = 0; s < S.ns; ++s) P.ea_ptr[s + 1] += P.ea_ptr[s];
    }
    // ... and the vector version for the stand-alone forward solves: a child's update entry jj sits in the last row
    // of column cnc + jj of its front
    P.ev_ptr.assign(S.ns + 1, 0);
    P.ev_src_ptr.push_back(0);
    {
        std::vector<std::pair<int, int>> con;            // (local index, source offset)
        for (int s = 0; s < S.ns; ++s) {
            con.clear();
            for (int q = S.child_ptr[s]; q < S.child_ptr[s + 1]; ++q) {
                const int c = S.child[q], cnc = S.sn_nc[c], cnr = S.sn_nr[c], cfs = cnc + cnr, cld = cfs + 1;
                const int *rel = S.rel.data() + S.sn_rowptr[c];
                for (int jj = 0; jj < cnr; ++jj) con.emplace_back(rel[jj], (int)(P.off[c] + (long)(cnc + jj) * cld + cfs));
            }
            std::stable_sort(con.begin(), con.end(), [](const std::pair<int, int> &a, const std::pair<int, int> &b) { return a.first < b.first; });
            for (size_t r = 0; r < con.size(); ++r) {
                if (r == 0 || con[r].first != con[r - 1].first) {
                    if (r) P.ev_src_ptr.push_back((int)P.ev_src.size());
                    P.ev_idx.push_back(con[r].first);
                    P.ev_ptr[s + 1]++;
                }
                P.ev_src.push_back(con[r].second);
            }
            if (!con.empty()) P.ev_src_ptr.push_back((int)P.ev_src.size());
        }
        for (int s = 0; s < S.ns; ++s) P.ev_ptr[s + 1] += P.ev_ptr[s];
    }
    // launch schedule: level by level (leaves first); inside a level one launch per class of front height in 16-row tiles T
    // (the front with its right-hand-side row): the front kernels of mfront.hip are compiled per T (k_mf_front<T, ...>,
    // T <= 8); taller fronts share the generic kernels: class 8 = T <= 13 (k_mf_factor2<8, 12>), class 9 = larger (rank-1 kernel)
    auto tiles = [&](int s) { return (S.sn_nc[s] + S.sn_nr[s] + 1 + 15) / 16; };
    // launch classes by the kernel that runs the front: the static kernels exist for every T <= 8, but a level of a large
    // structure holds fronts of every height, and a launch per height and level made the factorisation of the 1354- and
    // 9241-bus shapes launch-bound (68 levels x up to 10 launches; measured -12 % / -17 % QP/s against round 2's six
    // classes).  Fronts of one or two tiles share the T = 2 kernel, T = 7 shares the T = 8 kernel.
    auto cls = [&](int s) { const int T = tiles(s); return T <= 2 ? 1 : (T <= 6 ? T - 1 : (T <= 8 ? 7 : (T <= 13 ? 8 : 9))); };
    for (int l = 0; l < S.nlevels; ++l) {
        // a level with a handful of fronts (the upper part of the tree) is ONE launch of the kernel of its tallest
        // front: a front of fewer tiles runs in it with empty tiles, off the critical path of the level, and every
        // launch saved is a dependent kernel boundary less in a chain of ~45 per sweep
        int cnt = 0, tmax = 0;
        for (int q = S.level_ptr[l]; q < S.level_ptr[l + 1]; ++q) { ++cnt; tmax = std::max(tmax, tiles(S.level_sn[q])); }
        // (experiment, round 4: SQPHIP_MF_MERGE_T = t merges every level whose tallest front has at most t tiles, however many
        //  fronts it holds -- the small fronts then run in the kernel of the tallest)
        //  (measured, monotone sweep, QP/s without / with t = 4: 512 resident scenarios 8 995 / 8 517, 256: 6 570 / 6 285, 128: 4 309 / 4 321,
        //  64: 2 549 / 2 611, 32: 1 373 / 1 427 -- the caller asks for it up to 64 instances, SymOptions::merge_tiles)
        const int merge_t = getenv("SQPHIP_MF_MERGE_T") ? atoi(getenv("SQPHIP_MF_MERGE_T")) : opt.merge_tiles;
        const bool merge = ((cnt <= 8 && tmax <= 8) || tmax <= merge_t) && !getenv("SQPHIP_MF_NO_LEVEL_MERGE");
        for (int c = 0; c < 10; ++c) {
            MfLaunch L{(int)P.sched.size(), 0, 0, 0, c, 0};
            L.level = l;
            for (int q = S.level_ptr[l]; q < S.level_ptr[l + 1]; ++q) {
                const int s = S.level_sn[q];
                if (merge ? c != tmax - 1 : cls(s) != c) continue;
                P.sched.push_back(s);
                L.count++;
                L.tiles = std::max(L.tiles, tiles(s));
            }
            if (!L.count) continue;
            if (!merge && c == 1) L.tiles = 2;        // the kernel of the class (a front of fewer tiles runs in it with empty tiles)
            if (!merge && c == 7) L.tiles = 8;
            P.fac.push_back(L);                       // threads and LDS are the kernel's business (mf_factor)
        }
    }
    // solves: one launch per level; a workgroup of four waves takes one front of more than 64 rows or four smaller ones
    // (the top of the tree -- P.top_level, decided above -- is one launch per pass: k_mf_solve_top2 / k_mf_solve_top)
    auto build_solve_launches = [&]() {
    P.sol_items.clear(); P.fwd.clear(); P.bwd.clear(); P.top = MfLaunch{0, 0, 256, 0, 0, 0};
    int top_maxfs = 64, top_lcap = 0;
    for (int l = 0; l < S.nlevels; ++l) {
        const bool top = l >= P.top_level;
        MfLaunch L{(int)P.sol_items.size() / 4, 0, 256, 0, 0, 0};
        int maxfs = 64, lcap = 0;                       // lcap: doubles for the LDS image of a big front's triangular corner
        std::vector<int> small;
        for (int q = S.level_ptr[l]; q < S.level_ptr[l + 1]; ++q) {
            const int s = S.level_sn[q], fs = S.sn_nc[s] + S.sn_nr[s];
            if (!big_solve) maxfs = std::max(maxfs, fs);
            if (fs > 64 && big_solve) {
                for (int t : {s, -2, -2, -2}) P.sol_items.push_back(t);
                maxfs = std::max(maxfs, fs);
                const int nc = S.sn_nc[s];
                if (nc <= 80) lcap = std::max(lcap, nc * nc);           // up to 51 KB; larger corners stay in the arena
            } else small.push_back(s);
        }
        for (size_t q = 0; q < small.size(); q += 4)
            for (size_t t = q; t < q + 4; ++t) P.sol_items.push_back(t < small.size() ? small[t] : -1);
        L.count = (int)P.sol_items.size() / 4 - L.begin;
        // four wave vectors, or one front vector + 16 block sums; tiles = doubles per wave vector
        L.tiles = big_solve ? 64 : maxfs;
        L.cls = std::max(4 * L.tiles, maxfs + 16);                     // doubles of the vector area (cls is free in solve launches)
        L.lds_bytes = 8 * (L.cls + lcap);
        {   // LDS-staged kernels: four waves x (64-entry vector + image nc x ll) or one front of up to 128 rows (128 + image)
            int wimg = 0; long bigimg = 0; bool ok2 = big_solve;
            for (int q = S.level_ptr[l]; q < S.level_ptr[l + 1]; ++q) {
                const int s = S.level_sn[q], nc = S.sn_nc[s], fs = nc + S.sn_nr[s];
                if (fs > 128 || nc > 112) { ok2 = false; break; }
                if (fs > 64) bigimg = std::max(bigimg, (long)nc * (fs | 1)); else wimg = std::max(wimg, nc * (fs | 1));
            }
            const long lds2 = 8 * std::max(4L * (64 + wimg), 128 + bigimg);
            if (ok2 && lds2 <= 150 * 1024) { L.wimg = wimg; L.lds2 = (int)lds2; L.hasbig = bigimg > 0 ? 1 : 0; }
        }
        if (!top) { P.fwd.push_back(L); continue; }
        if (l == P.top_level) P.top.begin = L.begin;
        P.top.count += L.count;
        top_maxfs = std::max(top_maxfs, maxfs); top_lcap = std::max(top_lcap, lcap);
    }
    if (P.top.count) {
        P.top.tiles = 64;
        P.top.cls = std::max(4 * 64, top_maxfs + 16);
        P.top.lds_bytes = 8 * (P.top.cls + top_lcap);
    }
    P.bwd.assign(P.fwd.rbegin(), P.fwd.rend());
    };
    build_solve_launches();
    // The top again, for the streaming kernel (k_mf_solve_top2): fronts ascending (children before parents; every ancestor
    // of a top front is a top front).  One wave walks the fronts with everything it touches in LDS -- the factor image of
    // the front (prefetched by the other waves while the previous front is solved), the updates of the children, the
    // solution of the ancestors -- so a front costs its dependent arithmetic, not a chain of memory round trips.
    if (P.top.count > 0 && !(getenv("SQPHIP_MF_TOP2") && atoi(getenv("SQPHIP_MF_TOP2")) == 0)) {
        std::vector<int> kof(S.ns, -1);
        bool ok = true;
        for (int s = 0; s < S.ns; ++s) {
            if (S.sn_level[s] < P.top_level) continue;
            const int nc = S.sn_nc[s], nr = S.sn_nr[s], fs = nc + nr;
            if (fs > 128 || nc > 84) { ok = false; break; }      // two rows per lane; 28 columns per loader wave (MF_TOP_CH)
            kof[s] = (int)P.top_fr.size();
            MfTopFront F{s, nc, nr, S.sn_first[s], (int)P.off[s], fs | 1, P.top_xtotal, P.top_utotal, 0, 0, 0, 0, 0, 0, 0, 0};
            P.top_xtotal += nc; P.top_utotal += nr;
            P.top_fr.push_back(F);
        }
        // gather lists: local row i of front s receives, children ascending, entry jj of child c where rel[jj] == i; a child
        // inside the top hands its update over in LDS (index n_ext + uoff + jj, fixed up below), one below the top has
        // left it in the arena, from where the kernel's prologue fetches all of them at once (top_ext)
        std::vector<std::pair<int, int>> con;
        for (size_t k = 0; ok && k < P.top_fr.size(); ++k) {
            MfTopFront &F = P.top_fr[k];
            const int s = F.s, fs = F.nc + F.nr;
            con.clear();
            for (int q = S.child_ptr[s]; q < S.child_ptr[s + 1]; ++q) {
                const int c = S.child[q], cnc = S.sn_nc[c], cnr = S.sn_nr[c], cfs = cnc + cnr, cld = cfs + 1;
                const int *rel = S.rel.data() + S.sn_rowptr[c];
                for (int jj = 0; jj < cnr; ++jj) {
                    int src;
                    if (kof[c] >= 0) src = -1 - (P.top_fr[kof[c]].uoff + jj);
                    else { src = (int)P.top_ext.size(); P.top_ext.push_back((int)(P.off[c] + (long)(cnc + jj) * cld + cfs)); }
                    con.emplace_back(rel[jj], src);
                }
            }
            std::stable_sort(con.begin(), con.end(), [](const std::pair<int, int> &a, const std::pair<int, int> &b) { return a.first < b.first; });
            F.gptr = (int)P.top_gptr.size(); F.gsrc0 = (int)P.top_gsrc.size(); F.nsrc = (int)con.size();
            size_t r = 0;
            for (int i = 0; i <= fs; ++i) {
                P.top_gptr.push_back((int)r);
                while (r < con.size() && con[r].first == i) { P.top_gsrc.push_back(con[r].second); ++r; }
            }
            if (r != con.size()) { fprintf(stderr, "sqphip: mf_build_plan: update entry outside the parent front\n"); abort(); }
            F.rloc = (int)P.top_rows.size();
            for (int t = 0; t < F.nr; ++t) {
                const int qpos = S.sn_rows[S.sn_rowptr[s] + t], p = S.col2sn[qpos];
                if (kof[p] < 0) { fprintf(stderr, "sqphip: mf_build_plan: ancestor of a top front below the top\n"); abort(); }
                P.top_rows.push_back(P.top_fr[kof[p]].xloc + qpos - S.sn_first[p]);
            }
            // LDS buffer: [L image nc x ll][1 / D: nc][right-hand side or D^-1 L^-1 b: nc][ints: fs + 1 pointers, sources, rows]
            F.lbuf = F.nc * F.ll + 2 * F.nc + (fs + 1 + F.nsrc + F.nr + 1) / 2;
            int &bmax = (k & 1) ? P.top_buf1 : P.top_buf0;
            bmax = std::max(bmax, F.lbuf);
        }
        const int n_ext = (int)P.top_ext.size();
        for (int &v : P.top_gsrc) if (v < 0) v = n_ext + (-1 - v);
        // [buf0][buf1][128 doubles of slack][updates: n_ext + utotal][x: xtotal][D^-1 L^-1 b: xtotal][front records: 16 ints each]
        const long lds = 8L * ((long)P.top_buf0 + P.top_buf1 + n_ext + P.top_utotal + 2L * P.top_xtotal + 128 + 8L * (long)P.top_fr.size());
        if (getenv("SQPHIP_SYM_DUMP"))
            fprintf(stderr, "top (streamed): %d fronts from level %d, %d external + %d internal update entries, %d columns, LDS %ld bytes (buffers %d + %d doubles)%s\n",
                    (int)P.top_fr.size(), P.top_level, n_ext, P.top_utotal, P.top_xtotal, lds, P.top_buf0, P.top_buf1, ok ? "" : " -- a front of more than 128 rows: not used");
        if (ok && !P.top_fr.empty() && lds <= 160 * 1024 - 2048) P.top2_lds_bytes = lds;
        else {
            P.top_fr.clear(); P.top_gptr.clear(); P.top_gsrc.clear(); P.top_rows.clear(); P.top_ext.clear();
            // no streamed top for this structure (fronts beyond 128 rows or 84 columns): then the levels of the top go back
            // to level launches, which have the LDS-staged kernels for every level they fit -- k_mf_solve_top, one
            // workgroup per instance walking the fronts with the round-2 routines, is the slower of the two by now
            // (1354-bus shape: 495 -> 524 QP/s); SQPHIP_MF_TOP=1 keeps it
            if (!(getenv("SQPHIP_MF_TOP") && atoi(getenv("SQPHIP_MF_TOP")) == 1)) { P.top_level = S.nlevels; build_solve_launches(); }
        }
    }
    // packed records for the kernels
    P.desc.resize(S.ns);
    for (int s = 0; s < S.ns; ++s)
        P.desc[s] = {S.sn_nc[s], S.sn_nr[s], S.sn_first[s], S.sn_rowptr[s], (int)P.off[s], P.asm_ptr[s], P.asm_ptr[s + 1],
                     P.ea_ptr[s], P.ea_ptr[s + 1], P.ev_ptr[s], P.ev_ptr[s + 1], 0};
    P.ea_ent.resize(P.ea_rc.size());
    for (size_t t = 0; t < P.ea_rc.size(); ++t) P.ea_ent[t] = {P.ea_rc[t], P.ea_src_ptr[t], P.ea_src_ptr[t + 1], P.ea_src[P.ea_src_ptr[t]]};
    P.ev_ent.resize(P.ev_idx.size());
    for (size_t t = 0; t < P.ev_idx.size(); ++t) P.ev_ent[t] = {P.ev_idx[t], P.ev_src_ptr[t], P.ev_src_ptr[t + 1], P.ev_src[P.ev_src_ptr[t]]};
    // ---- the spine of the factorisation (k_mf_spine): the fronts of the levels >= spine_level by one workgroup per instance
    for (const MfLaunch &L : P.fac) if (L.level < P.spine_level) P.fac_below++;
    {
        // (measured, round 4: the kernel reproduces the level launches bit for bit and keeps the blocks it hands over off the
        //  memory bus, but a front costs it what it costs a level launch -- 230 us for the 17 fronts of IEEE-118 against ~170 us
        //  of level launches, which run the two branches of the tree side by side -- and its 120 KB of LDS keeps every other
        //  kernel off its CU: 7 251 against 7 704 QP/s at 512 resident scenarios, 1 946 against 2 014 at 64.  Off unless asked for.)
        bool ok = P.spine_level < S.nlevels && getenv("SQPHIP_MF_SPINE") && atoi(getenv("SQPHIP_MF_SPINE")) == 1;
        std::vector<int> sp;                             // spine fronts, ascending
        for (int s = 0; ok && s < S.ns; ++s)
            if (S.sn_level[s] >= P.spine_level) {
                if (tiles(s) > 8) ok = false;
                sp.push_back(s);
                P.spine_T = std::max(P.spine_T, tiles(s));
            }
        for (size_t k = 0; ok && k < sp.size(); ++k) {
            const int s = sp[k], nc = S.sn_nc[s], nr = S.sn_nr[s], fs = nc + nr;
            MfSpineFront F{s, nc, nr, S.sn_first[s], (int)P.off[s], tiles(s), P.asm_ptr[s], P.asm_ptr[s + 1], 0, 0, 0, 0, 0, 0, 0, 0};
            // the child that hands its block over in registers: the front right before this one in the spine, if it is a child
            const int reg_child = (k > 0 && S.sn_parent[sp[k - 1]] == s) ? sp[k - 1] : -1;
            std::vector<std::tuple<int, int, int>> con;      // (column, row, source offset): the extend-add from the arena
            const std::vector<int> ch = children_of(s);
            if (reg_child >= 0 && ch.back() != reg_child) { fprintf(stderr, "sqphip: mf_build_plan: the spine child is not the last child\n"); abort(); }
            for (int c : ch) {
                if (c == reg_child) continue;
                const int cnc = S.sn_nc[c], cnr = S.sn_nr[c], cld = cnc + cnr + 1;
                const int *rel = S.rel.data() + S.sn_rowptr[c];
                for (int jj = 0; jj < cnr; ++jj)
                    for (int ii = jj; ii <= cnr; ++ii)
                        con.emplace_back(rel[jj], ii < cnr ? rel[ii] : fs, (int)(P.off[c] + (long)(cnc + jj) * cld + cnc + ii));
            }
            std::stable_sort(con.begin(), con.end(), [](const std::tuple<int, int, int> &a, const std::tuple<int, int, int> &b) {
                return std::tie(std::get<0>(a), std::get<1>(a)) < std::tie(std::get<0>(b), std::get<1>(b)); });
            F.ea_begin = (int)P.sp_ent.size();
            bool open = false;
            for (size_t r = 0; r < con.size(); ++r) {
                const bool fresh = r == 0 || std::get<0>(con[r]) != std::get<0>(con[r - 1]) || std::get<1>(con[r]) != std::get<1>(con[r - 1]);
                if (fresh) {
                    if (open) P.sp_ent.back().src_end = (int)P.sp_src.size();
                    P.sp_ent.push_back({std::get<1>(con[r]) | (std::get<0>(con[r]) << 16), (int)P.sp_src.size(), 0, std::get<2>(con[r])});
                    open = true;
                }
                P.sp_src.push_back(std::get<2>(con[r]));
            }
            if (open) P.sp_ent.back().src_end = (int)P.sp_src.size();
            F.ea_end = (int)P.sp_ent.size();
            // hand-off to the next front?
            if (k + 1 < sp.size() && S.sn_parent[s] == sp[k + 1]) {
                const int p = sp[k + 1], pfs = S.sn_nc[p] + S.sn_nr[p];
                F.handoff = 1; F.rel = (int)P.sp_rel.size();
                for (int t = 0; t < nr; ++t) P.sp_rel.push_back(S.rel[S.sn_rowptr[s] + t]);
                P.sp_rel.push_back(pfs);
                P.spine_stage = std::max(P.spine_stage, (nr + 1) * nr);
            }
            P.sp_fr.push_back(F);
        }
        if (ok && !P.sp_fr.empty()) {
            // LDS: the static front kernels' layout for the tallest front with eight waves (mfront.hip, mf_front_lds_doubles) + the row map
            const int R = 16 * P.spine_T;
            P.spine_stage = (P.spine_stage + 1) / 2 * 2;
            P.spine_lds_bytes = 8L * ((long)R * R + 256 + 40 + R + 64 * 8 + P.spine_stage + 136);
            if (P.spine_lds_bytes > 160 * 1024 - 1024) ok = false;
        }
        if (!ok || P.sp_fr.empty()) { P.sp_fr.clear(); P.sp_ent.clear(); P.sp_src.clear(); P.sp_rel.clear(); P.spine_lds_bytes = 0; P.spine_level = S.nlevels; P.fac_below = (int)P.fac.size(); }
        if (getenv("SQPHIP_SYM_DUMP"))
            for (const MfSpineFront &F : P.sp_fr) {
                int hist[5] = {0, 0, 0, 0, 0};
                for (int t = F.ea_begin; t < F.ea_end; ++t) hist[std::min(4, P.sp_ent[t].src_end - P.sp_ent[t].src_begin)]++;
                fprintf(stderr, "  spine front %d: %d x %d, T %d, %d structural entries, %d receiving entries from the arena (1 / 2 / 3 / 4+ sources: %d / %d / %d / %d), handoff %d\n",
                        F.s, F.nc, F.nr, F.T, F.asm_end - F.asm_begin, F.ea_end - F.ea_begin, hist[1], hist[2], hist[3], hist[4], F.handoff);
            }
        if (getenv("SQPHIP_SYM_DUMP"))
            fprintf(stderr, "spine (factorisation): %d fronts from level %d, tallest %d tiles, %zu arena gather entries, LDS %ld bytes\n",
                    (int)P.sp_fr.size(), P.spine_level, P.spine_T, P.sp_ent.size(), P.spine_lds_bytes);
    }
    return P;
}

// value of one assembly item (the device twin is mf_item_value in mfront.hip)
static inline double item_value(const MfItem &it, const MfValues &V)
{
    const double reg_p = 1e-8, reg_d = 1e-8;           // IPM_REG_P / IPM_REG_D of ipm.hip
    switch (it.type) {
    case MF_ITEM_H: return V.hsc * V.hv[it.a];
    case MF_ITEM_JKEPT: return V.rtype[it.row] != 0 ? V.jv[it.a] : 0.0;
    case MF_ITEM_PAIR: return V.rtype[it.row] != 0 ? V.jv[it.a] * V.jv[it.b] / (V.Dd[it.row] + reg_d) : 0.0;
    case MF_ITEM_VDIAG: return V.hd[it.a] + V.sigp[it.a] + V.dw + reg_p;
    default: return V.rtype[it.row] != 0 ? -(V.Dd[it.row] + reg_d) : -1.0;
    }
}

static double g_top2_err = -1.0;          // last replay of the streamed top solve on the host: relative error, -1: not applicable
static double g_spine_err = -1.0;         // last replay of the spine kernel's assembly on the host: relative error, -1: not applicable

void mf_host_factor_solve(const MfPlan &P, const MfValues &V, const double *rhs, double *sol, double *dinv)
{
    const SparseSym &S = P.S;
    std::vector<double> F(P.stride, 0.0), x(S.nu), v(S.nu);
    for (int u = 0; u < S.nu; ++u) x[S.pos[u]] = rhs[u];
    for (int s = 0; s < S.ns; ++s) {
        const int nc = S.sn_nc[s], nr = S.sn_nr[s], fs = nc + nr, ld = fs + 1, f0 = S.sn_first[s];
        double *A = F.data() + P.off[s];
        for (int e = P.asm_ptr[s]; e < P.asm_ptr[s + 1]; ++e) {
            double a = 0.0;
            for (int k = P.item_ptr[e]; k < P.item_ptr[e + 1]; ++k) a += item_value(P.items[k], V);
            A[P.dest_loc[e]] = a;
        }
        for (int j = 0; j < nc; ++j) A[j * ld + fs] = x[f0 + j];
        for (int q = S.child_ptr[s]; q < S.child_ptr[s + 1]; ++q) {      // (a sum; the kernels fix its order through the gather lists)
            const int c = S.child[q], cnc = S.sn_nc[c], cnr = S.sn_nr[c], cfs = cnc + cnr, cld = cfs + 1;
            const double *C = F.data() + P.off[c];
            const int *rel = S.rel.data() + S.sn_rowptr[c];
            for (int jj = 0; jj < cnr; ++jj)
                for (int ii = jj; ii <= cnr; ++ii)
                    A[rel[jj] * ld + (ii < cnr ? rel[ii] : fs)] += C[(cnc + jj) * cld + cnc + ii];
        }
        // the assembly of the spine kernel (k_mf_spine) replayed from ITS plan arrays -- gather entries whose sources lie in the
        // arena, the row map of the block the previous front hands over, the destination list -- against the image assembled
        // above: validates those arrays without a GPU (CPU tests read the error through sqphip_mf_host_spine_err)
        if (!P.sp_fr.empty()) {
            if (s == P.sp_fr[0].s) g_spine_err = 0.0;
            for (size_t k = 0; k < P.sp_fr.size(); ++k) {
                const MfSpineFront &R = P.sp_fr[k];
                if (R.s != s) continue;
                const int Rn = 16 * R.T;
                std::vector<double> img((size_t)Rn * Rn, 0.0);
                for (int t = R.ea_begin; t < R.ea_end; ++t) {
                    const MfGather &g = P.sp_ent[t];
                    double a = F[g.src0];
                    for (int q = g.src_begin + 1; q < g.src_end; ++q) a += F[P.sp_src[q]];
                    img[(g.where >> 16) * Rn + (g.where & 0xffff)] += a;
                }
                if (k > 0 && P.sp_fr[k - 1].handoff) {
                    const MfSpineFront &Cc = P.sp_fr[k - 1];
                    const int cld = Cc.nc + Cc.nr + 1;
                    const double *Cb = F.data() + Cc.off;
                    const int *rel = P.sp_rel.data() + Cc.rel;
                    for (int c = 0; c < Cc.nr; ++c)
                        for (int r = c; r <= Cc.nr; ++r) img[rel[c] * Rn + rel[r]] += Cb[(Cc.nc + c) * cld + Cc.nc + r];
                }
                for (int e = R.asm_begin; e < R.asm_end; ++e) {
                    double a = 0.0;
                    for (int q = P.item_ptr[e]; q < P.item_ptr[e + 1]; ++q) a += item_value(P.items[q], V);
                    double &dst = img[(P.dest_rc[e] >> 16) * Rn + (P.dest_rc[e] & 0xffff)];
                    dst = a + dst;
                }
                for (int j = 0; j < nc; ++j) img[j * Rn + fs] = x[f0 + j] + img[j * Rn + fs];
                double err = 0.0, scale = 0.0;
                for (int c = 0; c < fs; ++c)
                    for (int r = c; r <= fs; ++r) {
                        err = std::max(err, std::fabs(img[c * Rn + r] - A[c * ld + r]));
                        scale = std::max(scale, std::fabs(A[c * ld + r]));
                    }
                g_spine_err = std::max(g_spine_err, err / std::max(scale, 1e-300));
            }
        }
        for (int k = 0; k < nc; ++k) {
            const double d = A[k * ld + k], di = 1.0 / d;
            dinv[f0 + k] = di;
            for (int j = k + 1; j < fs; ++j) {
                const double lj = A[k * ld + j] * di;
                for (int i = j; i <= fs; ++i) A[j * ld + i] -= A[k * ld + i] * lj;
            }
            for (int i = k + 1; i <= fs; ++i) A[k * ld + i] *= di;      // L, and z = D^-1 L^-1 b in the last row
            v[f0 + k] = A[k * ld + fs];
        }
    }
    // the streamed top-of-tree solve (k_mf_solve_top2) replayed on the host from ITS plan arrays -- front records, gather
    // lists whose sources are indices into the vector of updates, row -> index maps -- against the plain recursion above
    // and below: validates those arrays without a GPU (CPU tests read the error through sqphip_mf_host_top2_err)
    g_top2_err = -1.0;
    std::vector<double> xtop, ytop;
    if (P.top2_lds_bytes > 0) {
        const int n_ext = (int)P.top_ext.size();
        std::vector<double> uvec(n_ext + P.top_utotal, 0.0), y;
        xtop.assign(P.top_xtotal, 0.0); ytop.assign(P.top_xtotal, 0.0);
        for (int t = 0; t < n_ext; ++t) uvec[t] = F[P.top_ext[t]];
        double err = 0.0, scale = 0.0;
        for (const MfTopFront &T : P.top_fr) {
            const int nc = T.nc, fs = T.nc + T.nr, ld = fs + 1;
            const double *A = F.data() + T.off;
            y.assign(fs, 0.0);
            for (int i = 0; i < fs; ++i) {
                if (i < nc) y[i] = x[T.first + i];
                for (int q = P.top_gptr[T.gptr + i]; q < P.top_gptr[T.gptr + i + 1]; ++q) y[i] += uvec[P.top_gsrc[T.gsrc0 + q]];
            }
            for (int k = 0; k < nc; ++k)
                for (int i = k + 1; i < fs; ++i) y[i] -= A[k * ld + i] * y[k];
            for (int i = 0; i < nc; ++i) {
                ytop[T.xloc + i] = y[i] * dinv[T.first + i];
                err = std::max(err, std::fabs(ytop[T.xloc + i] - v[T.first + i])); scale = std::max(scale, std::fabs(v[T.first + i]));
            }
            for (int i = nc; i < fs; ++i) uvec[n_ext + T.uoff + i - nc] = y[i];
        }
        g_top2_err = err / std::max(scale, 1e-300);
    }
    // backward: x_cols = L11^-T (z - L21' x_rows), roots first
    for (int s = S.ns - 1; s >= 0; --s) {
        const int nc = S.sn_nc[s], nr = S.sn_nr[s], fs = nc + nr, ld = fs + 1, f0 = S.sn_first[s];
        const double *A = F.data() + P.off[s];
        const int *R = S.sn_rows.data() + S.sn_rowptr[s];
        for (int k = nc - 1; k >= 0; --k) {
            double a = v[f0 + k];
            for (int i = k + 1; i < nc; ++i) a -= A[k * ld + i] * x[f0 + i];
            for (int r = 0; r < nr; ++r) a -= A[k * ld + nc + r] * x[R[r]];
            x[f0 + k] = a;
        }
    }
    if (P.top2_lds_bytes > 0) {
        double err = 0.0, scale = 0.0;
        std::vector<double> t;
        for (int k = (int)P.top_fr.size() - 1; k >= 0; --k) {
            const MfTopFront &T = P.top_fr[k];
            const int nc = T.nc, nr = T.nr, fs = nc + nr, ld = fs + 1;
            const double *A = F.data() + T.off;
            t.assign(nc, 0.0);
            for (int c = 0; c < nc; ++c) {
                t[c] = ytop[T.xloc + c];
                for (int r = 0; r < nr; ++r) t[c] -= A[c * ld + nc + r] * xtop[P.top_rows[T.rloc + r]];
            }
            for (int i = nc - 1; i >= 1; --i)
                for (int c = 0; c < i; ++c) t[c] -= A[c * ld + i] * t[i];
            for (int c = 0; c < nc; ++c) {
                xtop[T.xloc + c] = t[c];
                err = std::max(err, std::fabs(t[c] - x[T.first + c])); scale = std::max(scale, std::fabs(x[T.first + c]));
            }
        }
        g_top2_err = std::max(g_top2_err, err / std::max(scale, 1e-300));
    }
    for (int u = 0; u < S.nu; ++u) sol[u] = x[S.pos[u]];
}

}  // namespace sqphip

extern "C" double sqphip_mf_host_top2_err(void) { return sqphip::g_top2_err; }
extern "C" double sqphip_mf_host_spine_err(void) { return sqphip::g_spine_err; }

// C-ABI test hook (host only, no GPU): plan + host reference of the numeric phase for the NLP structure given as in
// sqphip_create; values in the library's internal layouts (see include/sqphip.h).
extern "C" int sqphip_mf_host_solve(int64_t n, int64_t m, int64_t nnzJ, const int64_t *jrow, const int64_t *jcol,
                                    int64_t nnzH, const int64_t *hrow, const int64_t *hcol, const double *gL,
                                    const double *gU, int32_t condense, const double *Jval, const double *Hval,
                                    const double *Dd, const double *sigp, const double *hd, const int32_t *rtype,
                                    double hsc, double dw, const double *rhs, double *sol, double *dinv_by_unknown,
                                    int32_t *npos)
{
    if (n <= 0 || m < 0 || !rhs || !sol) return SQPHIP_EINVAL;
    using namespace sqphip;
    // CSC of J with duplicate summation and the full symmetric CSC of H, as sqphip_create builds them
    struct Ent { int c, r; double v; };
    auto build = [](int64_t ncols, int64_t nnz, const int64_t *row, const int64_t *col, const double *val, bool sym,
                    std::vector<int> &colptr, std::vector<int> &rowval, std::vector<double> &vals) {
        std::vector<Ent> e;
        for (int64_t k = 0; k < nnz; ++k) {
            e.push_back({(int)col[k] - 1, (int)row[k] - 1, val[k]});
            if (sym && row[k] != col[k]) e.push_back({(int)row[k] - 1, (int)col[k] - 1, val[k]});
        }
        std::stable_sort(e.begin(), e.end(), [](const Ent &a, const Ent &b) { return std::tie(a.c, a.r) < std::tie(b.c, b.r); });
        colptr.assign(ncols + 1, 0);
        for (size_t i = 0; i < e.size(); ++i) {
            if (i && e[i].c == e[i - 1].c && e[i].r == e[i - 1].r) { vals.back() += e[i].v; continue; }
            rowval.push_back(e[i].r); vals.push_back(e[i].v); colptr[e[i].c + 1]++;
        }
        for (int64_t j = 0; j < ncols; ++j) colptr[j + 1] += colptr[j];
    };
    std::vector<int> jcp, jrv, hcp, hrv;
    std::vector<double> jv, hv;
    build(n, nnzJ, jrow, jcol, Jval, false, jcp, jrv, jv);
    build(n, nnzH, hrow, hcol, Hval, true, hcp, hrv, hv);
    std::vector<int> rptr(m + 1, 0), rcol(jrv.size()), rslot(jrv.size());
    for (int r : jrv) rptr[r + 1]++;
    for (int64_t i = 0; i < m; ++i) rptr[i + 1] += rptr[i];
    {
        std::vector<int> fill(rptr.begin(), rptr.end() - 1);
        for (int j = 0; j < (int)n; ++j)
            for (int s = jcp[j]; s < jcp[j + 1]; ++s) { const int i = jrv[s]; rcol[fill[i]] = j; rslot[fill[i]] = s; fill[i]++; }
    }
    std::vector<int> kpos(m > 0 ? m : 1, -1);
    int mk = 0;
    for (int64_t i = 0; i < m; ++i)
        if (!condense || kkt_row_is_kept(gL[i], gU[i], rptr[i + 1] - rptr[i])) kpos[i] = mk++;
    MfPlan P = mf_build_plan((int)n, (int)m, kpos, mk, hcp, hrv, rptr, rcol, rslot, SymOptions());
    std::vector<int> rt(m > 0 ? m : 1, 1);
    for (int64_t i = 0; i < m; ++i) rt[i] = rtype ? rtype[i] : 1;
    MfValues V{hv.data(), jv.data(), Dd, sigp, hd, rt.data(), hsc, dw};
    std::vector<double> dinv(P.S.nu);
    mf_host_factor_solve(P, V, rhs, sol, dinv.data());
    int np = 0;
    for (int u = 0; u < P.S.nu; ++u) {
        const double d = dinv[P.S.pos[u]];
        if (dinv_by_unknown) dinv_by_unknown[u] = d;
        if (d > 0.0 && std::isfinite(d)) ++np;
    }
    if (npos) *npos = np;
    return SQPHIP_OK;
}
