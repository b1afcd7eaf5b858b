// sqphip_internal.hpp -- shared declarations of libsqphip's translation units (not part of the ABI)
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

namespace sqphip {

// kernel classes of a sweep for the per-kernel records of the bench line (Timers::detail; sqphip_get_kernel_times)
enum { KC_VALUES = 0, KC_FRONTS_LOW = 1, KC_FRONTS_TOP = 2, KC_SOLVE_TOP = 3, KC_SOLVE_LEVELS = 4, KC_POST = 5, KC_TRANS = 6, KC_COUNT = 7 };

struct Timers {
    bool enabled = false;
    // detail (sqphip_set_timing(ctx, 2)): an event pair around every launch group of a class -- off in the timed region of
    // the bench (twenty more event records per sweep), on in a short leg behind it
    bool detail = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending_class[KC_COUNT];
    double class_seconds[KC_COUNT] = {0, 0, 0, 0, 0, 0, 0};
    long class_groups[KC_COUNT] = {0, 0, 0, 0, 0, 0, 0};
    std::pair<hipEvent_t, hipEvent_t> open_ev;
    void open(hipStream_t s) { if (detail) { open_ev = get(); hipEventRecord(open_ev.first, s); } }
    void close(int cls, hipStream_t s) { if (detail) { hipEventRecord(open_ev.second, s); pending_class[cls].push_back(open_ev); class_groups[cls]++; } }
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pool;   // recycled event pairs
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending_trailing, pending_factor, pending_solve;
    double trailing_seconds = 0, factor_seconds = 0, solve_seconds = 0;
    long trailing_launches = 0, n_factor = 0;
    std::pair<hipEvent_t, hipEvent_t> get()
    {
        if (!pool.empty()) { auto p = pool.back(); pool.pop_back(); return p; }
        hipEvent_t a, b;
        hipEventCreate(&a); hipEventCreate(&b);
        return {a, b};
    }
    static double drain(std::vector<std::pair<hipEvent_t, hipEvent_t>> &v,
                        std::vector<std::pair<hipEvent_t, hipEvent_t>> &pool)
    {
        double s = 0;
        for (auto &p : v) {
            hipEventSynchronize(p.second);
            float ms = 0.f;
            hipEventElapsedTime(&ms, p.first, p.second);
            s += 1e-3 * (double)ms;
            pool.push_back(p);
        }
        v.clear();
        return s;
    }
    void flush()
    {
        trailing_seconds += drain(pending_trailing, pool);
        factor_seconds += drain(pending_factor, pool);
        solve_seconds += drain(pending_solve, pool);
        for (int c = 0; c < KC_COUNT; ++c) class_seconds[c] += drain(pending_class[c], pool);
    }
    ~Timers()
    {
        flush();
        for (auto &p : pool) { hipEventDestroy(p.first); hipEventDestroy(p.second); }
    }
};

struct LdltPlan {
    int N = 0, Npad = 0, T = 0, ld = 0, B = 0;
    int R = 4;                  // 64-wide sub-panels per outer panel: the trailing update has rank 64 R
    static constexpr int MAX_R = 4;
    double *Wbuf = nullptr;     // 2 MAX_R slots of [B][64][Npad] : W = L D of the sub-panels of two outer panels
    hipStream_t stream = nullptr;   // main stream (everything but the panel look-ahead)
    hipStream_t aux = nullptr;      // panel factorisation of the next outer panel
    hipEvent_t ev[6] = {};          // start, panel[2], head[2], join
    int trsm_mfma = 1;          // panel solve on the MFMA pipe, one wave per tile (SQPHIP_TRSM_MFMA=0: LDS substitution)
    int supertile = 8;          // tile columns per super-tile of the Schur-update schedule (SQPHIP_SUPERTILE; 1 = column-major)
    int Ts = 0;                 // leading tile columns that are mutually independent (order.hip); 0 = plain dense
    const unsigned char *tmask = nullptr;       // device copies of KktOrder::tmask / pair lists (null: no skipping)
    const int *pair_ptr = nullptr, *pair_k = nullptr;
    double lead_update_flops = 0.0;             // algorithmic flops of the list-restricted update behind the leading tiles
    int lookahead_min = 24;     // the look-ahead stream is used when the dense chain has at least this many tile
                                // columns (SQPHIP_LOOKAHEAD_MIN): +6 % QP/s at 44 columns, nothing at 33, -2.8 % at the
                                // 11 of the tile-ordered IEEE-118 matrices (events and a second queue for nothing)
    int trail_pad = 0;          // extra dynamic LDS bytes per k_trailing workgroup (SQPHIP_TRAIL_PAD): caps its residency
    int kc = 16;                // k-columns per LDS stage of the Schur-update kernel (SQPHIP_KC = 16 | 32)
    int tpb_max = 1;            // longest run of tiles one Schur-update workgroup takes (SQPHIP_TPB): runs of 8
                                // speed the bulk kernel up by 10 % but starve the look-ahead panel chain of CU
                                // slots; 1 gives the shortest whole factorisation (measured, N = 2813, B = 64)
    void init_lookahead()
    {
        // the panel chain is latency-critical and shares the chip with the bulk update: give its stream the
        // highest queue priority so its workgroups take the first CU slots that free up
        if (!getenv("SQPHIP_NO_LOOKAHEAD")) {
            int least = 0, greatest = 0;
            hipDeviceGetStreamPriorityRange(&least, &greatest);
            if (getenv("SQPHIP_NO_PRIORITY")) greatest = least;
            hipStreamCreateWithPriority(&aux, hipStreamNonBlocking, greatest);
        }
        if (const char *e = getenv("SQPHIP_LOOKAHEAD_MIN")) lookahead_min = atoi(e);
        if (const char *e = getenv("SQPHIP_OUTER")) { R = atoi(e); if (R < 1) R = 1; if (R > MAX_R) R = MAX_R; }
        if (const char *e = getenv("SQPHIP_TRSM_MFMA")) trsm_mfma = atoi(e);
        if (const char *e = getenv("SQPHIP_SUPERTILE")) { supertile = atoi(e); if (supertile < 1) supertile = 1; }
        if (const char *e = getenv("SQPHIP_KC")) kc = atoi(e) == 32 ? 32 : 16;
        // k_trailing holds 32 KB of static LDS; the pad may take it to at most 128 KB of the CU's 160 KB
        if (const char *e = getenv("SQPHIP_TRAIL_PAD")) { trail_pad = atoi(e); if (trail_pad < 0) trail_pad = 0; if (trail_pad > 96 * 1024) trail_pad = 96 * 1024; }
        if (const char *e = getenv("SQPHIP_TPB")) { tpb_max = atoi(e); if (tpb_max < 1) tpb_max = 1; }
        for (auto &e : ev) hipEventCreateWithFlags(&e, hipEventDisableTiming);
    }
    void destroy_lookahead()
    {
        if (aux) hipStreamDestroy(aux);
        for (auto &e : ev) if (e) hipEventDestroy(e);
        aux = nullptr;
    }
};

// phase-filtered launches: kernels skip instances whose phase[inst] != want (phase may be null)
void ldlt_factor(const LdltPlan &P, double *K, double *dinv, const int *phase, int want, Timers *tm,
                 double *b = nullptr, double *v = nullptr);
double ldlt_trailing_flops(const LdltPlan &P);   // algorithmic flops of the k_trailing launches of one factorisation
void ldlt_solve(const LdltPlan &P, const double *K, const double *dinv, double *x, double *v,
                const int *phase, int want, bool skip_fwd = false);

#define SQPHIP_HIP_OK(expr)                                                                    \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) {                                                                \
            char buf_[512];                                                                    \
            snprintf(buf_, sizeof buf_, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                     __FILE__, __LINE__);                                                      \
            throw std::string(buf_);                                                           \
        }                                                                                      \
    } while (0)

// order.hip: ordering of the condensed Newton matrix that exposes its tile sparsity
struct KktOrder {
    std::vector<int> pos;   // unknown (variable j, or n + kept-row position) -> position in the factorised matrix
    int Ts = 0;             // leading tile columns, mutually independent (block-diagonal leading Ts x Ts tile block)
    int Nf = 0;             // positions used: 64 * Ts for the tiles (identity padding inside) + the dense remainder
    int Tr = 0;             // tiles of the remainder
    std::vector<unsigned char> tmask;   // [Tr][Ts]: remainder tile r couples to leading tile k
    std::vector<int> pair_ptr, pair_k;  // per remainder tile pair (ti >= tj, index ti (ti + 1) / 2 + tj): the leading
                                        // tiles both couple to = the sub-panels of the rank-64 Ts update that matter
};
KktOrder kkt_order(int n, int m, const std::vector<int> &kpos, int mk, const std::vector<int> &hcolptr,
                   const std::vector<int> &hrowval, const std::vector<int> &jrowptr, const std::vector<int> &jrcol,
                   bool rows_last);

}  // namespace sqphip
