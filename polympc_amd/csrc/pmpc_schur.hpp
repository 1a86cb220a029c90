// polympc_amd — block-structured specialisations of the fused SQP kernel (pmpc_qp_schur.hpp) per built-in model: one translation unit per model
// (pmpc_schur_*.hip), compiled in parallel to the dense kernels. Grids: the BASELINE configurations and the reference's own test grids.
#pragma once
#include "pmpc_context.hpp"
#include "pmpc_models.hpp"
#include "pmpc_launch.hpp"

#define PMPC_SCHUR_ARGS pmpc_context* ctx, const MODEL& mdl, const pmpc::ChebData* cd, int P, int S, int B, const double* x_guess, const double* lam_guess, \
                        const double* d, const double* lbx, const double* ubx, const pmpc_sqp_settings* ss, const pmpc_qp_settings* qs, double* x,         \
                        double* lam, pmpc_sqp_info* info, hipStream_t stream, size_t lds_limit, unsigned long long* phase, pmpc_status* st
#define PMPC_SCHUR_TRY(PP, SS)                                                                                                                              \
    if (pmpc::try_launch_schur<MODEL, PP, SS>(ctx, mdl, cd, P, S, B, x_guess, lam_guess, d, lbx, ubx, ss, qs, x, lam, info, stream, lds_limit, phase, st)) \
        return true;
