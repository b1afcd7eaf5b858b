// sparse.hpp -- symbolic analysis of the sparse Newton matrix (host) and the plan the multifrontal kernels run
// (mfront.hip).  Not part of the ABI.
//
// Seat in the reference: the symbolic phase of the linear solver behind Ipopt (MUMPS / MA57 analysis,
// /root/reference/examples/acopf/opf.jl:59-64), done once per sparsity structure.
#pragma once
#include <cstdint>
#include <vector>

namespace sqphip {

// Result of the symbolic analysis of one symmetric pattern.
//   unknown u: variable j (u = j < n) or n + k for the k-th row that stays in the matrix
//   position:  rank of an unknown in the elimination order
// Supernodes are numbered in postorder (every child before its parent); the columns of a supernode are
// contiguous positions [first, first + nc); its front is the dense (nc + nr) x (nc + nr) matrix over
// [its columns | rows], rows = positions > last column, ascending.
struct SparseSym {
    int nu = 0;
    std::vector<int> pos, inv;
    int ns = 0;
    std::vector<int> sn_first, sn_nc, sn_nr, sn_rowptr, sn_rows, sn_parent, sn_level;
    std::vector<int> col2sn;            // position -> supernode
    std::vector<int> rel;               // rel[sn_rowptr[c] + k]: index inside the parent's front of row k of child c
    std::vector<int> child_ptr, child;  // children of every supernode, ascending
    std::vector<long> front_off;        // offset of front s inside an instance's front arena (doubles); ld = nc + nr
    long front_total = 0;
    int nlevels = 0;
    std::vector<int> level_ptr, level_sn;   // supernodes by level (leaves first), ascending inside a level
    // statistics
    long nnzL = 0;          // entries of L the dense fronts hold (explicit zeros of amalgamation included), diagonal excluded
    long nnzL_exact = 0;    // entries of L without amalgamation
    double flops = 0;       // multiply-adds x 2 of the dense partial factorisations
    double flops_exact = 0; // sum over columns of colcount^2 (no amalgamation)
    int max_front = 0, max_nc = 0;
};

// which rows stay in the condensed Newton matrix (options.kkt_condense): the equalities, whose block -D sits at the
// regularisation, and rows too long to eliminate -- the clique J_i' D_i^-1 J_i of a row with k entries has k^2 of
// them (a dense inequality such as sum(x) <= b would fill the whole matrix).  The oracle applies the same rule
// (oracle/qp_ipm.c, ora_qp_create).
constexpr int KKT_LONG_ROW = 32;
inline bool kkt_row_is_kept(double gl, double gu, int row_nnz) { return gl == gu || row_nnz > KKT_LONG_ROW; }

struct SymOptions {
    int rows_after_vars = 1;   // a row becomes eligible for elimination only behind every variable it couples to
    int small_front = 32;      // a child is always merged into its parent while the merged front stays within this
    double zero_frac = 0.25;   // ... or while the explicit zeros stay below this fraction of the merged supernode's L
    int order_method = 0;      // 0 approximate minimum degree (constrained), 1 natural order
    int merge_tiles = 0;       // launch schedule: a level whose tallest front has at most this many 16-row tiles is ONE launch of that
                               // front's kernel however many fronts it holds (small batches: a launch less per level beats the idle waves)
    int chain_front = 0;       // > 0: along the spine of the tree the deepest child is merged into its parent while the
                               // merged front stays within this many rows (symbolic.hip, second amalgamation pass)
};

// adj: symmetric adjacency lists over nu unknowns (sorted, unique, no self loops);
// need[u] (may be empty): for rows_after_vars, the unknowns that must precede u (the variables of row u)
SparseSym sparse_symbolic(int nu, const std::vector<std::vector<int>> &adj, const std::vector<std::vector<int>> &before,
                          const SymOptions &opt);

// the graph of the Newton matrix from the NLP structure: H full symmetric CSC, J CSR; kpos[i] >= 0: row i stays in
// the matrix (unknown n + kpos[i]), kpos[i] < 0: row i is eliminated (clique among its variables)
void kkt_graph(int n, int m, const std::vector<int> &kpos, int mk, const std::vector<int> &hcolptr,
               const std::vector<int> &hrowval, const std::vector<int> &jrowptr, const std::vector<int> &jrcol,
               std::vector<std::vector<int>> &adj, std::vector<std::vector<int>> &before);

// ---------------------------------------------------------------------------------------------------------------
// The multifrontal plan: SparseSym plus the assembly lists of the Newton matrix and the launch schedule.
//
// Front s of an instance is a dense (fs + 1) x fs column-major block (fs = nc + nr, ld = fs + 1) in the instance's
// front arena: columns = [columns of the supernode | its rows]; the extra LAST ROW carries a right-hand side through
// the elimination (fused forward solve: after the partial factorisation its first nc entries are L^-1 b of these
// columns, the remaining nr are the update this front passes to its ancestors, exactly like the contribution block).
// After mf_factor: columns 0..nc-1 hold L (unit diagonal implied, rows below), dinv[position] = 1 / D.
//
// Assembly: every structural entry of the lower triangle of the Newton matrix has ONE destination (front, local
// offset) and a fixed list of items that are summed in a fixed order (no atomics, reproducible):
enum { MF_ITEM_H = 0,        // hsc * hv[a]
       MF_ITEM_JKEPT = 1,    // jv[a] unless row `row` is free
       MF_ITEM_PAIR = 2,     // jv[a] * jv[b] / (Dd[row] + reg_d) unless row `row` is free  (eliminated row)
       MF_ITEM_VDIAG = 3,    // hd[a] + sigp[a] + delta_w + reg_p
       MF_ITEM_RDIAG = 4 };  // -(Dd[row] + reg_d), or -1 for a free row
struct MfItem { int type, row, a, b; };

// everything a kernel needs to know about a front, in one 48-byte record (three 16-byte loads instead of a chain of
// dependent look-ups): columns, rows, first position, offsets into rows[] / the front arena, and its ranges in the
// destination, extend-add and vector-gather lists
struct MfFrontDesc { int nc, nr, first, rowptr, off, asm_begin, asm_end, ea_begin, ea_end, ev_begin, ev_end, pad; };
// one receiving entry of an extend-add / vector gather: where it goes (row | column << 16, or the local index), its
// sources [src_begin, src_end) in ea_src / ev_src and, inline, the first of them (most entries have exactly one)
struct MfGather { int where, src_begin, src_end, src0; };

// one kernel launch of the factorisation / of a solve sweep: fronts [begin, begin + count) of `sched`, all of one
// size class.  Factorisation: cls = kernel variant (mfront.hip, mf_factor), tiles = 16-row tiles of the largest front
// of the launch (sizes the LDS image), lds_bytes = dynamic LDS.
// Solve launches also carry what the LDS-staged kernels (k_mf_fwd2 / k_mf_bwd2) need: wimg = doubles of a wave's image
// buffer (-1: a front of the level does not fit them), lds2 = their dynamic LDS.
struct MfLaunch { int begin, count, threads, lds_bytes, cls, tiles; int wimg = -1, lds2 = 0, hasbig = 1; int level = 0; };

// One front of the narrow top of the assembly tree as the streaming solve kernel (k_mf_solve_top2, mfront.hip) sees it:
// where its factor lives in the arena, the leading dimension of its LDS image (odd: the transposed reads of the
// backward chain then hit different banks), where its columns sit in the kernel's LDS copies of x / D^-1 L^-1 b (xloc),
// where its update vector goes (uoff, into the LDS vector of updates), its gather lists (pointers relative to the
// front, sources = indices into that vector) and the LDS indices of its rows (rloc).  lbuf = doubles of its LDS buffer.
struct MfTopFront { int s, nc, nr, first, off, ll, xloc, uoff, gptr, gsrc0, nsrc, rloc, lbuf, pad0, pad1, pad2; };

// One front of the spine of the factorisation (k_mf_spine, mfront.hip): the fronts of the levels >= spine_level, ascending
// (children first), eliminated one after the other by ONE workgroup per instance with the front image in LDS.  What
// the kernel needs per front in one 64-byte record: geometry, its tiles T (16-row tiles of the front with its
// right-hand-side row), its ranges in the destination list (dest_rc / vals: unchanged) and in the spine's own gather list
// sp_ent / sp_src -- the extend-add WITHOUT the contributions of the child that hands its block over in registers --,
// handoff = 1: the NEXT front of the spine is this front's parent and receives the contribution block straight from
// the accumulator registers (no trip through the arena), through the row map sp_rel[rel .. rel + nr] (local row of the
// block -> local index in the parent; entry nr = the parent's right-hand-side row).
struct MfSpineFront { int s, nc, nr, first, off, T, asm_begin, asm_end, ea_begin, ea_end, handoff, rel, pad0, pad1, pad2, pad3; };

struct MfPlan {
    SparseSym S;
    std::vector<long> off;                       // front offsets (doubles) with the (fs + 1) x fs layout
    long stride = 0;                             // doubles per instance
    std::vector<int> asm_ptr, dest_loc, item_ptr; // per front: destinations [asm_ptr[s], asm_ptr[s+1]); per destination: items
    std::vector<MfItem> items;
    // per front: (row | column << 16) of each destination, for kernels that keep the front in another leading dimension
    std::vector<int> dest_rc;
    // extend-add as a gather: per front the entries that receive contributions [ea_ptr[s], ea_ptr[s+1]), each with
    // its position (row | column << 16) and its sources [ea_src_ptr[t], ea_src_ptr[t+1]) = offsets into the
    // instance's front arena, children in ascending order (fixed summation order, no barrier between children)
    std::vector<int> ea_ptr, ea_rc, ea_src_ptr, ea_src;
    // the same for vectors (forward solves): per front the local indices that receive a child's update
    // [ev_ptr[s], ev_ptr[s+1]), each with its sources (offsets of the children's update entries in the arena)
    std::vector<int> ev_ptr, ev_idx, ev_src_ptr, ev_src;
    // solve launches: work items of one 256-thread workgroup each, four ints per item = one front of more than 64
    // rows, shared by the four waves: {front, -2, -2, -2}, or up to four smaller fronts, one wave each (-1: none)
    std::vector<int> sol_items;
    std::vector<MfFrontDesc> desc;               // packed per-front records (device copies of the arrays above)
    std::vector<MfGather> ea_ent, ev_ent;
    std::vector<int> sched;                      // fronts in launch order
    std::vector<MfLaunch> fac, fwd, bwd;
    MfLaunch top{0, 0, 256, 0, 0, 0};           // solves: the narrow top of the assembly tree (levels >= top_level) in one launch
    int top_level = 0;                            // = S.nlevels when there is no such launch
    // the same fronts for k_mf_solve_top2 (fronts ascending = children first); top2_lds_bytes == 0: not applicable
    // (a front of more than 128 rows, or the images do not fit the LDS) and k_mf_solve_top runs instead
    std::vector<MfTopFront> top_fr;
    std::vector<int> top_gptr, top_gsrc, top_rows, top_ext;     // gather pointers / sources, row indices, arena offsets of the updates that come from below the top
    int top_xtotal = 0, top_utotal = 0, top_buf0 = 0, top_buf1 = 0;
    long top2_lds_bytes = 0;
    long nnzK = 0;                               // structural entries of the lower triangle (destinations)
    // the spine of the factorisation (k_mf_spine): spine_level == S.nlevels: not applicable (a front beyond 8 tiles) or off
    int spine_level = 0, spine_T = 0;             // first level of the spine; tiles of its tallest front (sizes the LDS image)
    int narrow_level = 0;                         // first level of the narrow top of the tree, whether or not the top / spine kernels apply
    std::vector<MfSpineFront> sp_fr;
    std::vector<MfGather> sp_ent;                 // receiving entries of the spine fronts' extend-add from the ARENA
    std::vector<int> sp_src, sp_rel;
    long spine_lds_bytes = 0;
    int spine_stage = 0;                          // doubles of the LDS staging area: the largest (nr + 1) x nr among the blocks handed over
    int fac_below = 0;                            // launches of `fac` that lie below the spine (the rest are the level launches of the spine's fronts)
};

// kpos / krow as in DV; jcolptr.. = CSC of J, jrowptr/jrcol/jrslot its CSR view, hcolptr/hrowval full symmetric CSC
MfPlan mf_build_plan(int n, int m, const std::vector<int> &kpos, int mk, const std::vector<int> &hcolptr,
                     const std::vector<int> &hrowval, const std::vector<int> &jrowptr, const std::vector<int> &jrcol,
                     const std::vector<int> &jrslot, const SymOptions &opt);

// Host reference of the numeric phase (tests of the plan; never on the product path): assembles from the item lists,
// factorises front by front, solves K x = rhs.  Vectors in unknown order (variables, then kept rows); dinv by position.
struct MfValues { const double *hv, *jv, *Dd, *sigp, *hd; const int *rtype; double hsc, dw; };
void mf_host_factor_solve(const MfPlan &P, const MfValues &V, const double *rhs, double *sol, double *dinv);

}  // namespace sqphip
