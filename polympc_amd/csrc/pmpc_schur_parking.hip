// polympc_amd — block-structured kernel for ParkingOCP (NP = 1: the bordered form of pmpc_qp_schur.hpp) on the grid of the reference's minimal_time_test.cpp
// (P = 5, S = 2: 56 variables, 33 equality rows). A DEVELOPER SWITCH (PMPC_SCHUR_NP=1), built, checked bit for bit against its restatement, and not made the
// default: on the reference's minimal-time problem EVERY instance meets the conditioning gate of the range-space solve (PMPC_SCHUR_COND_GATE) as soon as the ADMM
// penalty adapts beyond ~25 — the states hardly enter the dynamics, so the state columns of the collocation Jacobian are nearly singular — and is solved again by
// the redo launch; the two-rows-per-lane dense kernel serves that problem directly (EXPERIMENTS.md round 5).
#include <cstdlib>
#include "pmpc_schur.hpp"
#define MODEL pmpc::ParkingOCP
namespace pmpc {
template <> bool try_launch_schur_grids<MODEL>(PMPC_SCHUR_ARGS) {
    if (!getenv("PMPC_SCHUR_NP")) return false;
    PMPC_SCHUR_TRY(5, 2)
    return false;
}
}  // namespace pmpc
