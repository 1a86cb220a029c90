// polympc_amd — block-structured SQP kernels of RobotOCP: config A / D (P = 6, S = 1), the reference's own grids (P = 5, S = 2: codegen robot;
// P = 5, S = 3: mpc_wrapper_test.cpp). (Config A's 56-row system is routed to the dense one-row-per-lane kernel unless PMPC_SCHUR_SMALL=1.)
#include "pmpc_schur.hpp"
#define MODEL pmpc::RobotOCP
namespace pmpc {
template <> bool try_launch_schur_grids<MODEL>(PMPC_SCHUR_ARGS) {
    PMPC_SCHUR_TRY(6, 1)
    PMPC_SCHUR_TRY(5, 2)
    PMPC_SCHUR_TRY(5, 3)
    return false;
}
}  // namespace pmpc
