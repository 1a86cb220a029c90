// polympc_amd — block-structured kernel for ParkingOCP (NP = 1: the bordered form of pmpc_qp_schur.hpp) on the grid of the reference's minimal_time_test.cpp
// (P = 5, S = 2: 56 variables, 33 equality rows). ON REQUEST (pmpc_sqp_settings::kkt_form = 2 — an API field since round 6, no environment switch), built,
// checked bit for bit against its restatement, and not made the default: on the reference's minimal-time problem EVERY instance meets the conditioning gate of
// the range-space solve (PMPC_SCHUR_COND_GATE) as soon as the ADMM penalty adapts beyond ~25 — the states hardly enter the dynamics, so the state block of the
// collocation Jacobian is singular (sigma_min 1e-16 at the reference's guess AND at its solution: what keeps the problem well posed are the BOUNDS that pin the
// initial state, which no elimination of the equality rows sees — the null-space / reduced-Hessian form fails for the same reason, EXPERIMENTS.md round 6) —
// and is solved again by the redo launch; the two-rows-per-lane dense kernel serves that problem directly (EXPERIMENTS.md round 5).
#include "pmpc_schur.hpp"
#define MODEL pmpc::ParkingOCP
namespace pmpc {
template <> bool try_launch_schur_grids<MODEL>(PMPC_SCHUR_ARGS) {
    if (ss->kkt_form != 2) return false;
    PMPC_SCHUR_TRY(5, 2)
    return false;
}
}  // namespace pmpc
