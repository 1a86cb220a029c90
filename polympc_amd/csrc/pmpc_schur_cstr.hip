// polympc_amd — block-structured SQP kernels of CstrOCP: config B = cstr_control_test.cpp's grid (P = 5, S = 2) and the one-segment grid P = 6
#include "pmpc_schur.hpp"
#define MODEL pmpc::CstrOCP
namespace pmpc {
template <> bool try_launch_schur_grids<MODEL>(PMPC_SCHUR_ARGS) {
    PMPC_SCHUR_TRY(5, 2)
    PMPC_SCHUR_TRY(6, 1)
    return false;
}
}  // namespace pmpc
