// Cycle-stamp trace of the Schur-update workgroups (one "rest" launch of a 64-instance batch).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -DSQPHIP_TRACE_TRAILING -I sqpsolver.jl_amd/csrc
#include "../../sqpsolver.jl_amd/csrc/ldlt.hip"
#include <algorithm>
#include <map>
#include <vector>
using namespace sqphip;

int main(int argc, char **argv)
{
    const int N = 2813, B = 64, R = argc > 1 ? atoi(argv[1]) : 4;
    LdltPlan P;
    P.N = N; P.Npad = (N + 63) / 64 * 64; P.T = P.Npad / 64; P.ld = P.Npad; P.B = B;
    hipStreamCreate(&P.stream);
    double *K;
    const size_t nk = (size_t)B * P.Npad * P.Npad, nw = (size_t)2 * LdltPlan::MAX_R * B * P.Npad * 64;
    hipMalloc(&K, nk * 8); hipMalloc(&P.Wbuf, nw * 8);
    hipMemset(K, 0, nk * 8); hipMemset(P.Wbuf, 0, nw * 8);
    P.init_lookahead();
    P.R = R;
    if (argc > 2) P.tpb_max = atoi(argv[2]);
    Timers tm; tm.enabled = true;
    for (int rep = 0; rep < 3; ++rep) launch_update(P, P.stream, K, 0, R, 0, 2 * R, P.T, nullptr, 0, &tm, true);
    hipStreamSynchronize(P.stream);
    tm.flush();
    static long long h[2048][16];
    hipMemcpyFromSymbol(h, HIP_SYMBOL(g_trace), sizeof h);
    const int ntl = tiles_in_cols(P.T, 2 * R, P.T);
    int tpb = std::max(1, std::min(P.tpb_max, ntl * B / 2048));
    const int nsamp = std::min(2048, (ntl + tpb - 1) / tpb * B / 61);
    double seg[16] = {0};
    long long tmin = h[0][0], tmax = 0;
    for (int i = 0; i < nsamp; ++i) {
        tmin = std::min(tmin, h[i][0]); tmax = std::max(tmax, h[i][15]);
        seg[0] += h[i][1] - h[i][0];
        for (int s = 0; s < R; ++s) {
            seg[1 + 2 * s] += h[i][2 + 2 * s] - (s ? h[i][1 + 2 * s] : h[i][1]);   // wait + LDS store + barriers
            seg[2 + 2 * s] += h[i][3 + 2 * s] - h[i][2 + 2 * s];                    // fetch issue + MFMA loop
        }
        seg[15] += h[i][15] - h[i][1 + 2 * R];
    }
    printf("(stamps of the LAST tile of each sampled run) R=%d tiles/instance %d, launch %.3f ms (avg of 3), span of sampled stamps %lld ticks\n", R, ntl,
           tm.trailing_seconds * 1e3 / 3, tmax - tmin);
    printf("prologue (index, C loads, fetch issue): %.0f\n", seg[0] / nsamp);
    for (int s = 0; s < R; ++s)
        printf("sub %d: stage %.0f   multiply %.0f   (ideal multiply 64 MFMA x 64 = 4096)\n", s, seg[1 + 2 * s] / nsamp,
               seg[2 + 2 * s] / nsamp);
    printf("epilogue (C add + stores issued): %.0f\n", seg[15] / nsamp);
    // occupancy per CU: sum of workgroup lifetimes / (slots x busy span of that CU)
    {
        const int nblk = std::min(1 << 17, (ntl + tpb - 1) / tpb * B);
        static long long sp[1 << 17][3];
        hipMemcpyFromSymbol(sp, HIP_SYMBOL(g_span), sizeof sp);
        std::map<long long, std::vector<std::pair<long long, long long>>> cu;
        for (int i = 0; i < nblk; ++i) {
            const long long xcc = (sp[i][2] >> 32) & 0xf, hw = sp[i][2] & 0xffffffff;
            const long long key = xcc << 16 | ((hw >> 13) & 7) << 8 | ((hw >> 8) & 15);   // XCC, SE_ID, CU_ID
            cu[key].push_back({sp[i][0], sp[i][1]});
        }
        double occ = 0, life = 0, gapmax = 0; long long spanmax = 0, nb_min = 1 << 30, nb_max = 0;
        for (auto &kv : cu) {
            long long lo = kv.second[0].first, hi = 0, busy = 0;
            for (auto &p : kv.second) { lo = std::min(lo, p.first); hi = std::max(hi, p.second); busy += p.second - p.first; }
            occ += (double)busy / (double)(hi - lo);
            life += (double)busy / kv.second.size();
            spanmax = std::max(spanmax, hi - lo);
            nb_min = std::min<long long>(nb_min, kv.second.size()); nb_max = std::max<long long>(nb_max, kv.second.size());
        }
        printf("CUs seen %zu; workgroups per CU min %lld max %lld; mean resident workgroups per CU %.2f; mean workgroup "
               "lifetime %.0f ticks; longest CU span %lld ticks\n", cu.size(), nb_min, nb_max, occ / cu.size(),
               life / cu.size(), spanmax);
    }
    return 0;
}
