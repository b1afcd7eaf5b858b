// acopf.hip -- K1: the five NLP callbacks of an ACOPF instance evaluated on the device.
//
// Seat in the reference: the closures eval_f / eval_grad_f / eval_g / eval_jac_g / eval_h built in
// /root/reference/src/MOI_wrapper.jl:1115-1146 and called at src/algorithms/sqp.jl:86-117,
// :130-138, :170-183, for a PowerModels ACPPowerModel + build_opf model
// (/root/reference/test/opf.jl:5-9; equations SURVEY.md Appendix B, generalised to tap ratios and phase shifts).
// Variable / row / COO-entry layout is the one documented in sqpsolver.jl_amd/acopf_synth.py.
// One workgroup per instance; threads stride over branches, buses and generators; every output
// entry is written by exactly one thread (no atomics).
#include "acopf_dev.hpp"

namespace sqphip {

__global__ __launch_bounds__(TPB) void k_acopf_eval_point(DV d, int inst, const double *x, double sigma,
                                                         const double *lam, double *f, double *grad,
                                                         double *g, double *jv, double *hv)
{
    acopf_eval(d, inst, x, sigma, lam, f, grad, g, jv, hv);
}

void launch_acopf_eval_point(Ctx &C, int inst, const double *x_dev, double sigma, const double *lam_dev,
                             double *f_dev, double *grad_dev, double *g_dev, double *jcoo_dev,
                             double *hcoo_dev)
{
    hipLaunchKernelGGL(k_acopf_eval_point, dim3(1), dim3(TPB), 0, C.stream, C.d, inst, x_dev, sigma, lam_dev,
                       f_dev, grad_dev, g_dev, jcoo_dev, hcoo_dev);
}

}  // namespace sqphip
