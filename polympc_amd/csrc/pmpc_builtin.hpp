// polympc_amd — per-model host entry points of the built-in OCPs. Declared for pmpc_api.hip (the C ABI dispatches on the model id);
// DEFINED (PMPC_BUILTIN_DEFINITIONS) and explicitly instantiated in one translation unit per model, pmpc_model_*.hip, so that the
// ~45 specialisations of the fused SQP kernel compile in parallel instead of in one three-minute translation unit.
#pragma once
#include "pmpc_context.hpp"
#include "pmpc_models.hpp"

template <class Model> pmpc_status sqp_builtin_dev(pmpc_context* ctx, int P, int S, double t0, double tf, const double* mp, int nmp, int B,
                                                   const double* x_guess, const double* lam_guess, const double* d, const double* lbx,
                                                   const double* ubx, const double* lbg, const double* ubg, const pmpc_sqp_settings* ss,
                                                   const pmpc_qp_settings* qs, double* x, double* lam, pmpc_sqp_info* info);
template <class Model> pmpc_status linearise_impl(pmpc_context* ctx, int P, int S, double t0, double tf, const double* mp, int nmp, int B,
                                                  const double* var, const double* d, const double* lam, double* cost, double* constr, double* jac,
                                                  double* cost_grad, double* lag_grad, double* lag_hess);

#ifdef PMPC_BUILTIN_DEFINITIONS
#include <vector>
#include "pmpc_launch.hpp"
using namespace pmpc;

template <class Model> static inline Model make_model(const double* mp, int nmp) { Model mdl; mdl.set_params(mp, nmp); return mdl; }

template <class Model>
pmpc_status sqp_builtin_dev(pmpc_context* ctx, int P, int S, double t0, double tf, const double* mp, int nmp, int B,
                                   const double* x_guess, const double* lam_guess, const double* d, const double* lbx,
                                   const double* ubx, const double* lbg, const double* ubg, const pmpc_sqp_settings* ss,
                                   const pmpc_qp_settings* qs, double* x, double* lam, pmpc_sqp_info* info) {
    const Model mdl = make_model<Model>(mp, nmp);
    return pmpc::sqp_launch_dev<Model>(ctx, mdl, P, S, t0, tf, B, x_guess, lam_guess, d, lbx, ubx, lbg, ubg, ss, qs, x, lam, info);
}

template <class Model>
pmpc_status linearise_impl(pmpc_context* ctx, int P, int S, double t0, double tf, const double* mp, int nmp, int B,
                                  const double* var, const double* d, const double* lam, double* cost, double* constr, double* jac,
                                  double* cost_grad, double* lag_grad, double* lag_hess) {
    const ChebData* cd = nullptr;
    pmpc_status st = get_cheb(ctx, P, S, t0, tf, &cd);
    if (st != PMPC_OK) return st;
    OcpDims<Model> dm(P, S);
    const int n = dm.n, m = dm.m;
    const size_t lds = linearise_kernel_lds_bytes<Model>(P, S);
    if (lds > ctx->lds_limit) return PMPC_ERR_UNSUPPORTED_SIZE;
    double *dvar, *dd, *dlam, *dcost, *dc, *dj, *dcg, *dlg, *dlh;
    H2D(0, var, (size_t)B * n, dvar); H2D(1, (Model::ND ? d : nullptr), (size_t)B * Model::ND, dd); H2D(2, lam, (size_t)B * (m + n), dlam);
    if (!Model::ND) DEVOUT(1, 8, dd);
    DEVOUT(3, (size_t)B * 2 * sizeof(double), dcost); DEVOUT(4, (size_t)B * m * sizeof(double), dc);
    DEVOUT(5, (size_t)B * m * n * sizeof(double), dj); DEVOUT(6, (size_t)B * n * sizeof(double), dcg);
    DEVOUT(7, (size_t)B * n * sizeof(double), dlg); DEVOUT(8, (size_t)B * n * n * sizeof(double), dlh);
    HIPCHK(hipFuncSetAttribute((const void*)linearise_kernel<Model>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    Model mdl = make_model<Model>(mp, nmp);
    hipLaunchKernelGGL(linearise_kernel<Model>, dim3(B), dim3(WAVE), lds, ctx->stream, mdl, cd, B, dvar, dd, dlam, dcost, dc, dj, dcg, dlg, dlh);
    HIPCHK(hipGetLastError());
    std::vector<double> c2((size_t)B * 2);
    HIPCHK(hipMemcpyAsync(c2.data(), dcost, (size_t)B * 2 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    if (constr) HIPCHK(hipMemcpyAsync(constr, dc, (size_t)B * m * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    if (jac) HIPCHK(hipMemcpyAsync(jac, dj, (size_t)B * m * n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    if (cost_grad) HIPCHK(hipMemcpyAsync(cost_grad, dcg, (size_t)B * n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    if (lag_grad) HIPCHK(hipMemcpyAsync(lag_grad, dlg, (size_t)B * n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    if (lag_hess) HIPCHK(hipMemcpyAsync(lag_hess, dlh, (size_t)B * n * n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (cost) for (int b = 0; b < 2 * B; ++b) cost[b] = c2[b];
    return PMPC_OK;
}


#define PMPC_INSTANTIATE_BUILTIN(Model)                                                                                                  \
    template pmpc_status sqp_builtin_dev<Model>(pmpc_context*, int, int, double, double, const double*, int, int, const double*,        \
                                                const double*, const double*, const double*, const double*, const double*, const double*, \
                                                const pmpc_sqp_settings*, const pmpc_qp_settings*, double*, double*, pmpc_sqp_info*);    \
    template pmpc_status linearise_impl<Model>(pmpc_context*, int, int, double, double, const double*, int, int, const double*,         \
                                               const double*, const double*, double*, double*, double*, double*, double*, double*);
#endif
