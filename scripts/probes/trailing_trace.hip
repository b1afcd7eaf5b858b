// Cycle-stamp trace of the Schur-update workgroups (one "rest" launch of a 64-instance batch).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -DSQPHIP_TRACE_TRAILING -I sqpsolver.jl_amd/csrc
#include "../../sqpsolver.jl_amd/csrc/ldlt.hip"
#include <algorithm>
#include <map>
#include <vector>
using namespace sqphip;

// pseudo-random doubles in (-1, 1): real mantissa activity (an all-zero matrix keeps the MFMA datapath quiet,
// draws 900 W and holds 2.4 GHz; random data is what the product multiplies)
__global__ void k_fill_random(double *p, size_t n, unsigned seed)
{
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        unsigned long long z = (i + 1) * 0x9E3779B97F4A7C15ull + seed;
        z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 27; z *= 0x94D049BB133111EBull; z ^= z >> 31;
        p[i] = ((double)(z >> 11) * (1.0 / 9007199254740992.0) * 2.0 - 1.0) * 1e-3;
    }
}

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
    if (argc > 4 && atoi(argv[4])) {
        k_fill_random<<<4096, 256>>>(K, nk, 1u);
        k_fill_random<<<4096, 256>>>(P.Wbuf, nw, 2u);
        hipDeviceSynchronize();
    }
    P.init_lookahead();
    P.R = R;
    if (argc > 2) P.tpb_max = atoi(argv[2]);
    Timers tm; tm.enabled = true;
    const int reps = argc > 3 ? atoi(argv[3]) : 3;   // many reps: hold the kernel on the chip while rocm-smi samples clocks
    for (int rep = 0; rep < reps; ++rep) {
        launch_update(P, P.stream, K, 0, R, 0, 2 * R, P.T, nullptr, 0, &tm, true);
        if (tm.pending_trailing.size() > 256) tm.flush();
    }
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
    printf("(stamps of the LAST tile of each sampled run) R=%d tiles/instance %d, launch %.3f ms (avg), span of sampled stamps %lld ticks\n", R, ntl,
           tm.trailing_seconds * 1e3 / reps, tmax - tmin);
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
