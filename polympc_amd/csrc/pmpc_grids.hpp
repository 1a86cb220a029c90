// polympc_amd — register-resident specialisations of the fused SQP kernel for further node counts of the built-in models: 3, 4, 6, 8, 9, 10, 12, 13 and 14 nodes — one KKT row
// per lane (pmpc_qp_reg.hpp) where n + m <= 64, two rows per lane (pmpc_qp_reg2.hpp) where 64 < n + m <= 112, nothing where the system is larger. The 5-, 7- and
// 11-node grids are part of every model's own translation unit (pmpc_launch.hpp); these compile in parallel to them, one translation unit per model
// (pmpc_grids_*.hip), without the phase-timer and block-BFGS variants (such requests take the LDS-resident kernel).
#pragma once
#include "pmpc_context.hpp"
#include "pmpc_models.hpp"
#include "pmpc_launch.hpp"

namespace pmpc {

template <class Model>
bool try_launch_extra_grids(pmpc_context* ctx, const Model& mdl, const ChebData* cd, int P, int S, int B, const double* x_guess,
                            const double* lam_guess, const double* d, const double* lbx, const double* ubx, const double* lbg,
                            const double* ubg, const pmpc_sqp_settings* ss, const pmpc_qp_settings* qs, double* Hws, double* Aws, double* x,
                            double* lam, pmpc_sqp_info* info, hipStream_t stream, size_t lds_limit, unsigned long long* phase, pmpc_status* st,
                            double* slice_state, int slice_iters) {
#define PMPC_TRY_GRID(NNODES_)                                                                                                                       \
    if (try_launch_reg<Model, NNODES_, true>(ctx, mdl, cd, P, S, B, x_guess, lam_guess, d, lbx, ubx, lbg, ubg, ss, qs, Hws, Aws, x, lam, info, stream, \
                                             lds_limit, phase, st, slice_state, slice_iters))                                                         \
        return true;
    PMPC_TRY_GRID(3)
    PMPC_TRY_GRID(4)
    PMPC_TRY_GRID(6)
    PMPC_TRY_GRID(8)
    PMPC_TRY_GRID(9)
    PMPC_TRY_GRID(10)
    PMPC_TRY_GRID(12)
    PMPC_TRY_GRID(13)
    PMPC_TRY_GRID(14)
    PMPC_TRY_GRID(15)   // 113..128 rows where the model fits them (robot: 15 and 16 nodes — the reference's mpc_wrapper_test grid; CSTR: 12 above)
    PMPC_TRY_GRID(16)
#undef PMPC_TRY_GRID
    return false;
}

}  // namespace pmpc

#define PMPC_INSTANTIATE_GRIDS(MODEL)                                                                                                                 \
    template bool pmpc::try_launch_extra_grids<MODEL>(pmpc_context*, const MODEL&, const pmpc::ChebData*, int, int, int, const double*, const double*,  \
                                                      const double*, const double*, const double*, const double*, const double*,                       \
                                                      const pmpc_sqp_settings*, const pmpc_qp_settings*, double*, double*, double*, double*,            \
                                                      pmpc_sqp_info*, hipStream_t, size_t, unsigned long long*, pmpc_status*, double*, int);
