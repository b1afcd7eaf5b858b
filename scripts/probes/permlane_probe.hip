#include <hip/hip_runtime.h>
__global__ void k(double *o, const double *i)
{
    const int lane = threadIdx.x;
    double x = i[lane];
    unsigned lo = __double2loint(x), hi = __double2hiint(x);
    // a' = (r0,r0,r2,r2), b' = (r1,r1,r3,r3)
    auto p16l = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    auto p16h = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    unsigned al = p16l[0], bl = p16l[1], ah = p16h[0], bh = p16h[1];
    auto q0l = __builtin_amdgcn_permlane32_swap(al, al, false, false);
    auto q0h = __builtin_amdgcn_permlane32_swap(ah, ah, false, false);
    auto q1l = __builtin_amdgcn_permlane32_swap(bl, bl, false, false);
    auto q1h = __builtin_amdgcn_permlane32_swap(bh, bh, false, false);
    o[lane] = __hiloint2double(q0h[0], q0l[0]);          // expect r0 everywhere
    o[64 + lane] = __hiloint2double(q1h[0], q1l[0]);     // expect r1
    o[128 + lane] = __hiloint2double(q0h[1], q0l[1]);    // expect r2
    o[192 + lane] = __hiloint2double(q1h[1], q1l[1]);    // r3
    // readlane with constant lane
    o[256 + lane] = __hiloint2double(__builtin_amdgcn_readlane(hi, 17), __builtin_amdgcn_readlane(lo, 17));
}
int main()
{
    double h[64], r[320], *di, *dout;
    for (int i = 0; i < 64; ++i) h[i] = i;
    hipMalloc(&di, sizeof h); hipMalloc(&dout, sizeof r);
    hipMemcpy(di, h, sizeof h, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dout, di);
    hipMemcpy(r, dout, sizeof r, hipMemcpyDeviceToHost);
    for (int v = 0; v < 5; ++v) { printf("v%d:", v); for (int i = 0; i < 64; i += 5) printf(" %g", r[64 * v + i]); printf("\n"); }
    return 0;
}
