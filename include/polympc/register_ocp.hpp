// polympc_amd — register a user-defined OCP for the GPU.
//
// The reference lets a user define an OCP as a C++ class with templated dynamics_impl<T> / lagrange_term_impl<T> /
// mayer_term_impl<T> / inequality_constraints_impl<T> (continuous_ocp.hpp:191-288; example
// tests/control/mpc_wrapper_test.cpp:37-80) and differentiates it with nested AutoDiffScalar. Here the same class body
// is compiled by hipcc: the templates are instantiated with pmpc::Dual (device forward-mode AD) inside the fused SQP
// kernel. Write the class against pmpc::cref<T> / pmpc::vref<T> views (element access x(i), as with Eigen::Ref), mark
// the four functions __device__, give it `enum { NX, NU, NP, ND, NG }`, and in ONE .hip translation unit:
//
//     #include <polympc/register_ocp.hpp>
//     struct MyOCP { enum { NX = 2, NU = 1, NP = 0, ND = 0, NG = 0 }; ... };
//     PMPC_REGISTER_OCP(MyOCP)
//
// compiled with:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared my_ocp.hip
//                       -I<repo>/include -L<repo>/polympc_amd -lpolympc_amd -o libmy_ocp.so
// This emits the C symbols pmpc_user_sqp_dev_MyOCP (device buffers, type pmpc_sqp_dev_fn) and pmpc_user_dims_MyOCP.
// Host code (any C++ compiler) then uses polympc::Solver<...> from <polympc/polympc.hpp> with POLYMPC_USE_REGISTERED_OCP.
#pragma once
#include "../polympc_amd.h"
#include "../../polympc_amd/csrc/pmpc_launch.hpp"

#define PMPC_REGISTER_OCP(Name)                                                                                                  \
    extern "C" pmpc_status pmpc_user_sqp_dev_##Name(pmpc_context* ctx, const void* model, int P, int S, double t0, double tf,   \
                                                    int B, const double* x_guess, const double* lam_guess, const double* d,     \
                                                    const double* lbx, const double* ubx, const double* lbg, const double* ubg, \
                                                    const pmpc_sqp_settings* ss, const pmpc_qp_settings* qs, double* x,         \
                                                    double* lam, pmpc_sqp_info* info) {                                         \
        return pmpc::sqp_launch_dev<Name>(ctx, *static_cast<const Name*>(model), P, S, t0, tf, B, x_guess, lam_guess, d, lbx,   \
                                          ubx, lbg, ubg, ss, qs, x, lam, info);                                                 \
    }                                                                                                                            \
    extern "C" void pmpc_user_dims_##Name(int* nx, int* nu, int* np, int* nd, int* ng) {                                        \
        *nx = Name::NX; *nu = Name::NU; *np = Name::NP; *nd = Name::ND; *ng = Name::NG;                                          \
    }
