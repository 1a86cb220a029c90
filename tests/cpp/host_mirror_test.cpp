// Host-side tests written like the reference's own gtest cases, against include/polympc/polympc.hpp.
//   box_admmSimpleQP / SimpleLP / NonConvex    tests/solvers/qp/box_admm_test.cpp:15-45, :266-297, :299-334
//   box_admmSinglePrecisionFloat                tests/solvers/qp/box_admm_test.cpp:85-115 (boxADMM<2, 1, float>)
//   admmSinglePrecisionFloat                    tests/solvers/qp/admm_solver_test.cpp:84-113 (ADMM<2, 1, float>)
//   MPCWrapperTest                              tests/control/mpc_wrapper_test.cpp:120-199 (dense-BFGS variant)
//   ValetParkingTest                            tests/control/valet_parking_mpc_test.cpp:183-240 (Ruiz + filter line search + block BFGS)
//   ShardedBatchSolver...                       SURVEY 8e: BatchSolver::set_devices / pmpc_sqp_solve_batch_multi, shards == single context
//   user-registered OCP                         docs/source/ocp.rst:229-481 workflow, compiled by hipcc (user_ocp.hip)
#include <cstdio>
#include <cstring>
#include <polympc/polympc.hpp>

static int failures = 0;
#define EXPECT_TRUE(c) do { if (!(c)) { std::printf("  EXPECT_TRUE failed: %s (%s:%d)\n", #c, __FILE__, __LINE__); ++failures; } } while (0)
#define EXPECT_LT(a, b) EXPECT_TRUE((a) < (b))
#define EXPECT_EQ(a, b) EXPECT_TRUE((a) == (b))

using namespace polympc;

static void box_admmSimpleQP() {
    std::printf("box_admmSimpleQP\n");
    boxADMM<2, 1>::qp_hessian_t H; boxADMM<2, 1>::qp_var_t h, xl, xu, solution; boxADMM<2, 1>::qp_constraint_t A; boxADMM<2, 1>::qp_dual_a_t Al, Au;
    H(0, 0) = 4; H(0, 1) = 1; H(1, 0) = 1; H(1, 1) = 2;
    h(0) = 1; h(1) = 1; A(0, 0) = 1; A(0, 1) = 1; Al(0) = 1; Au(0) = 1; xl(0) = 0; xl(1) = 0; xu(0) = 0.7; xu(1) = 0.7;
    solution(0) = 0.3; solution(1) = 0.7;
    boxADMM<2, 1> prob;
    prob.settings().max_iter = 150;
    prob.solve(H, h, A, Al, Au, xl, xu);
    EXPECT_TRUE(prob.primal_solution().isApprox(solution, 1e-2));
    EXPECT_LT(prob.iter, prob.settings().max_iter);
    EXPECT_EQ(prob.info().status, SOLVED);
}

static void box_admmSinglePrecisionFloat() {   // box_admm_test.cpp:85-115
    std::printf("box_admmSinglePrecisionFloat\n");
    using Scalar = float;
    using QP = boxADMM<2, 1, Scalar>;
    QP::qp_hessian_t H; QP::qp_var_t h, xl, xu, solution; QP::qp_constraint_t A; QP::qp_dual_a_t Al, Au;
    H(0, 0) = 4; H(0, 1) = 1; H(1, 0) = 1; H(1, 1) = 2;
    h(0) = 1; h(1) = 1; A(0, 0) = 1; A(0, 1) = 1; Al(0) = 1; Au(0) = 1; xl(0) = 0; xl(1) = 0; xu(0) = Scalar(0.7); xu(1) = Scalar(0.7);
    solution(0) = Scalar(0.3); solution(1) = Scalar(0.7);
    QP prob;
    prob.settings().max_iter = 150;
    prob.solve(H, h, A, Al, Au, xl, xu);
    const QP::qp_var_t sol = prob.primal_solution();
    EXPECT_TRUE(sol.isApprox(solution, Scalar(1e-2)));
    EXPECT_LT(prob.iter, prob.settings().max_iter);
    EXPECT_EQ(prob.info().status, SOLVED);
}

static void admmSinglePrecisionFloat() {   // admm_solver_test.cpp:84-113
    std::printf("admmSinglePrecisionFloat\n");
    using Scalar = float;
    using QP = ADMM<2, 1, Scalar>;
    QP::qp_hessian_t H; QP::qp_var_t h, xl, xu, solution; QP::qp_constraint_t A; QP::qp_dual_a_t al, au;
    H(0, 0) = 4; H(0, 1) = 1; H(1, 0) = 1; H(1, 1) = 2;
    h(0) = 1; h(1) = 1; A(0, 0) = 1; A(0, 1) = 1; al(0) = 1; xl(0) = 0; xl(1) = 0; au(0) = 1; xu(0) = Scalar(0.7); xu(1) = Scalar(0.7);
    solution(0) = Scalar(0.3); solution(1) = Scalar(0.7);
    QP prob;
    prob.solve(H, h, A, al, au, xl, xu);
    const QP::qp_var_t sol = prob.primal_solution();
    EXPECT_TRUE(sol.isApprox(solution, Scalar(1e-2)));
    EXPECT_LT(prob.iter, prob.settings().max_iter);
    EXPECT_EQ(prob.info().status, SOLVED);
}

static void box_admmRuizEquilibration() {   // box_admm_test.cpp:47-83
    std::printf("box_admmRuizEquilibration\n");
    boxADMM<2, 1>::qp_hessian_t H; boxADMM<2, 1>::qp_var_t h, xl, xu, solution; boxADMM<2, 1>::qp_constraint_t A; boxADMM<2, 1>::qp_dual_a_t Al, Au;
    H(0, 0) = 4; H(0, 1) = 1; H(1, 0) = 1; H(1, 1) = 2;
    h(0) = 1; h(1) = 1; A(0, 0) = 1; A(0, 1) = 1; Al(0) = 1; Au(0) = 1; xl(0) = 0; xl(1) = 0; xu(0) = 0.7; xu(1) = 0.7;
    solution(0) = 0.3; solution(1) = 0.7;
    boxADMM<2, 1> prob;
    prob.settings().max_iter = 150;
    polympc::RuizEquilibration<2, 1> preconditioner;
    preconditioner.compute(H, h, A, Al, Au, xl, xu);
    prob.solve(H, h, A, Al, Au, xl, xu);
    boxADMM<2, 1>::qp_var_t sol = prob.primal_solution();
    boxADMM<2, 1>::qp_dual_t sol_dual = prob.dual_solution();
    preconditioner.unscale(sol, sol_dual);
    EXPECT_TRUE(sol.isApprox(solution, 1e-2));
    EXPECT_LT(prob.iter, prob.settings().max_iter);
    EXPECT_EQ(prob.info().status, SOLVED);
}

static void admmSimpleQP() {   // admm_solver_test.cpp:16-45
    std::printf("admmSimpleQP\n");
    ADMM<2, 1>::qp_hessian_t H; ADMM<2, 1>::qp_var_t h, xl, xu, solution; ADMM<2, 1>::qp_constraint_t A; ADMM<2, 1>::qp_dual_a_t Al, Au;
    H(0, 0) = 4; H(0, 1) = 1; H(1, 0) = 1; H(1, 1) = 2;
    h(0) = 1; h(1) = 1; A(0, 0) = 1; A(0, 1) = 1; Al(0) = 1; Au(0) = 1; xl(0) = 0; xl(1) = 0; xu(0) = 0.7; xu(1) = 0.7;
    solution(0) = 0.3; solution(1) = 0.7;
    ADMM<2, 1> prob;
    prob.settings().max_iter = 1000;
    prob.solve(H, h, A, Al, Au, xl, xu);
    EXPECT_TRUE(prob.primal_solution().isApprox(solution, 1e-2));
    EXPECT_LT(prob.iter, prob.settings().max_iter);
    EXPECT_EQ(prob.info().status, SOLVED);
}

static void box_admmNonConvex() {
    std::printf("box_admmNonConvex\n");
    boxADMM<1, 0>::qp_hessian_t H; boxADMM<1, 0>::qp_var_t h, xl, xu, guess; boxADMM<1, 0>::qp_constraint_t A; boxADMM<1, 0>::qp_dual_a_t al, au;
    boxADMM<1, 0>::qp_dual_t dual_guess;
    H(0, 0) = -1; h(0) = 0; xl(0) = -1; xu(0) = 2; guess(0) = 0.1; dual_guess(0) = 0.1;
    boxADMM<1, 0> prob;
    prob.settings().max_iter = 200; prob.settings().alpha = 1.0; prob.settings().adaptive_rho = 1; prob.settings().rho = 2; prob.settings().check_termination = 10;
    prob.solve(H, h, A, al, au, xl, xu, guess, dual_guess);
    EXPECT_TRUE(std::fabs(prob.primal_solution()(0) - 2.0) <= 2e-2);
    EXPECT_LT(prob.iter, prob.settings().max_iter);
    EXPECT_EQ(prob.info().status, SOLVED);
}

using Polynomial = polympc::Chebyshev<5, polympc::GAUSS_LOBATTO, double>;
using Approximation = polympc::Spline<Polynomial, 3>;
using RobotOCP = polympc::models::MobileRobot<Approximation>;

static void BatchMPCTest() {   // B controllers whose state stays on the device between steps (mpc_wrapper_test.cpp:120-166 per controller)
    std::printf("BatchMPCTest\n");
    const int B = 6;
    BatchMPC<RobotOCP> mpc(B);
    mpc.ocp().set_Q_coeff(2.0);
    mpc.settings().max_iter = 10; mpc.settings().line_search_max_iter = 10;
    mpc.set_time_limits(0, 2);
    Vector<2> lbu, ubu; lbu(0) = -1.5; lbu(1) = -0.75; ubu(0) = 1.5; ubu(1) = 0.75;
    Vector<1> p; p(0) = 2.0;
    for (int b = 0; b < B; ++b) { mpc.control_bounds(b, lbu, ubu); mpc.set_static_parameters(b, p); }
    std::vector<double> x0(B * 3), u0(B * 2), u1(B * 2);
    for (int b = 0; b < B; ++b) { x0[3 * b] = 0.5 + 0.01 * b; x0[3 * b + 1] = 0.5; x0[3 * b + 2] = 0.5; }
    EXPECT_EQ(mpc.step(x0.data(), u0.data()), PMPC_OK);
    int first = 0, second = 0;
    for (int b = 0; b < B; ++b) { EXPECT_EQ(mpc.info(b).status, (int)PMPC_SQP_SOLVED); first += mpc.info(b).iter; }
    for (int b = 0; b < B; ++b) { EXPECT_TRUE(u0[2 * b] >= -1.5 - 1e-3 && u0[2 * b] <= 1.5 + 1e-3 && std::fabs(u0[2 * b + 1]) <= 0.75 + 1e-3); }
    // the single-controller facade solves controller 0's problem to the same control
    {
        MPC<RobotOCP> one; one.ocp().set_Q_coeff(2.0); one.settings().max_iter = 10; one.settings().line_search_max_iter = 10; one.set_time_limits(0, 2);
        MPC<RobotOCP>::static_param pp; pp(0) = 2.0; MPC<RobotOCP>::state_t s0; s0(0) = 0.5; s0(1) = 0.5; s0(2) = 0.5;
        MPC<RobotOCP>::control_t l2, u2; l2(0) = -1.5; l2(1) = -0.75; u2(0) = 1.5; u2(1) = 0.75;
        one.set_static_parameters(pp); one.control_bounds(l2, u2); one.initial_conditions(s0); one.solve();
        const auto uu = one.solution_u_at(0);   // index 0 = t_start: the accessor counts nodes from the end of the block (mpc_wrapper.hpp:241-244)
        EXPECT_TRUE(std::fabs(uu(0) - u0[0]) <= 1e-9 && std::fabs(uu(1) - u0[1]) <= 1e-9);
    }
    for (int b = 0; b < B; ++b) { x0[3 * b] = 0.3; x0[3 * b + 1] = 0.4; x0[3 * b + 2] = 0.5; }   // moved state, warm start from the resident solution
    EXPECT_EQ(mpc.step(x0.data(), u1.data()), PMPC_OK);
    for (int b = 0; b < B; ++b) { EXPECT_EQ(mpc.info(b).status, (int)PMPC_SQP_SOLVED); second += mpc.info(b).iter; }
    EXPECT_LT(second, first);
    std::vector<double> xs;
    EXPECT_EQ(mpc.solution(xs), PMPC_OK);
    for (int b = 0; b < B; ++b) EXPECT_TRUE(std::fabs(xs[(size_t)b * 80 + 45] - 0.3) <= 1e-3);   // pinned initial state honoured (last nx entries of the x block)
}

static void MPCWrapperTest() {
    std::printf("MPCWrapperTest\n");
    using mpc_t = MPC<RobotOCP>;
    mpc_t mpc;
    mpc.ocp().set_Q_coeff(2.0);
    mpc.settings().max_iter = 10;
    mpc.settings().line_search_max_iter = 10;
    mpc.set_time_limits(0, 2);
    mpc_t::static_param p; p(0) = 2.0;
    mpc_t::state_t x0; x0(0) = 0.5; x0(1) = 0.5; x0(2) = 0.5;
    mpc_t::control_t lbu, ubu; lbu(0) = -1.5; lbu(1) = -0.75; ubu(0) = 1.5; ubu(1) = 0.75;
    mpc.set_static_parameters(p);
    mpc.control_bounds(lbu, ubu);
    mpc.initial_conditions(x0);
    mpc.solve();
    const int first_solve_iter = mpc.info().iter;
    EXPECT_TRUE(mpc.info().status.value == sqp_status_t::SOLVED);
    x0(0) = 0.3; x0(1) = 0.4; x0(2) = 0.5;
    mpc.initial_conditions(x0, x0);
    mpc.solve();
    const int second_solve_iter = mpc.info().iter;
    EXPECT_LT(second_solve_iter, first_solve_iter);
    EXPECT_TRUE(mpc.info().status.value == sqp_status_t::SOLVED);
    std::printf("  iterations: cold %d, warm %d; cost %.6f violation %.2e\n", first_solve_iter, second_solve_iter, mpc.cost(), mpc.constr_violation());
    EXPECT_TRUE(mpc.solution_x_at(0).isApprox(mpc.solution_x_at(0.0), 1e-3));
    EXPECT_TRUE(mpc.solution_x_at(5).isApprox(mpc.solution_x_at(0.666), 1e-3));
    EXPECT_TRUE(mpc.solution_x_at(10).isApprox(mpc.solution_x_at(1.333), 1e-3));
    EXPECT_TRUE(mpc.solution_u_at(0).isApprox(mpc.solution_u_at(0.0), 1e-3));
    EXPECT_TRUE(mpc.solution_u_at(1).isApprox(mpc.solution_u_at(0.063), 1e-3));
    EXPECT_TRUE(std::fabs(mpc.solution_x_at(0)(0) - 0.3) < 1e-3);   // the pinned initial state is the FIRST point in time
}

// sqp_settings_t::iteration_callback (sqp_base.hpp:33, :685-686): called once per SQP iteration from the second one on with the solver, whose getters
// then describe THAT iteration. The fused kernel records those values; Solver<OCP>::solve() replays them.
static std::vector<int> cb_iters;
static std::vector<double> cb_primal;
static void record_iteration(void* solver) {
    auto* s = static_cast<Solver<RobotOCP>*>(solver);
    cb_iters.push_back(s->info().iter);
    cb_primal.push_back(s->primal_norm());
}
static void IterationCallbackTest() {
    std::printf("IterationCallbackTest\n");
    using mpc_t = MPC<RobotOCP>;
    mpc_t mpc;
    mpc.ocp().set_Q_coeff(2.0);
    mpc.settings().max_iter = 10;
    mpc.settings().line_search_max_iter = 10;
    mpc.settings().iteration_callback = &record_iteration;
    mpc.set_time_limits(0, 2);
    mpc_t::static_param p; p(0) = 2.0;
    mpc_t::state_t x0; x0(0) = 0.5; x0(1) = 0.5; x0(2) = 0.5;
    mpc_t::control_t lbu, ubu; lbu(0) = -1.5; lbu(1) = -0.75; ubu(0) = 1.5; ubu(1) = 0.75;
    mpc.set_static_parameters(p);
    mpc.control_bounds(lbu, ubu);
    mpc.initial_conditions(x0);
    cb_iters.clear(); cb_primal.clear();
    mpc.solve();
    const int iters = mpc.info().iter;
    EXPECT_TRUE(mpc.info().status.value == sqp_status_t::SOLVED);
    EXPECT_TRUE(iters >= 2 && (int)cb_iters.size() == iters - 1);
    for (size_t k = 0; k < cb_iters.size(); ++k) EXPECT_TRUE(cb_iters[k] == (int)k + 2);
    EXPECT_TRUE(!cb_primal.empty() && cb_primal.back() == mpc.primal_norm());
    EXPECT_TRUE(cb_primal.size() < 2 || cb_primal.front() > cb_primal.back());   // the steps shrink towards the solution
    std::printf("  %d iterations, %zu callbacks, |step| %.3e -> %.3e\n", iters, cb_iters.size(), cb_primal.empty() ? 0.0 : cb_primal.front(), cb_primal.empty() ? 0.0 : cb_primal.back());
    mpc.settings().iteration_callback = nullptr;
}

// valet_parking_mpc_test.cpp:183-240 — the reference plugs three hooks into SQPBase there (Ruiz preconditioner as template argument,
// filter line search :116-158, block BFGS :160-165); here they are the three settings flags, `solver.filter.beta` keeps its name.
static void ValetParkingTest() {
    std::printf("ValetParkingTest\n");
    using OCP = polympc::models::MobileRobot<polympc::Spline<polympc::Chebyshev<5>, 2>>;
    Solver<OCP> solver;
    solver.get_problem().set_Q_coeff(1.0);
    solver.get_problem().set_time_limits(0, 2);
    solver.settings().max_iter = 10;
    solver.settings().line_search_max_iter = 10;
    solver.settings().preconditioner = 1; solver.settings().hessian_update = 1; solver.settings().line_search = 1;
    solver.qp_settings().max_iter = 1000;
    solver.parameters()(0) = 2.0;
    solver.filter.beta = 0.1;
    double init_cond[3] = {0.5, 0.5, 0.5};
    for (int k = 0; k < 11; ++k) {
        solver.upper_bound_x()(33 + 2 * k) = 1.5;  solver.upper_bound_x()(34 + 2 * k) = 0.75;
        solver.lower_bound_x()(33 + 2 * k) = -1.5; solver.lower_bound_x()(34 + 2 * k) = -0.75;
    }
    for (int i = 0; i < 3; ++i) { solver.upper_bound_x()(30 + i) = init_cond[i]; solver.lower_bound_x()(30 + i) = init_cond[i]; }
    solver.solve();
    const int cold = solver.info().iter;
    EXPECT_TRUE(solver.info().status.value == sqp_status_t::SOLVED);
    EXPECT_TRUE(solver.info_flags() == 0);   // (the device's information word: nothing non-finite, no conditioning gate met)
    EXPECT_LT(solver.info().iter, solver.settings().max_iter);
    const size_t kept = solver.filter.entries(0).size();
    EXPECT_TRUE(kept >= 1 && kept <= 10);
    // warm started iteration
    init_cond[0] = 0.3; init_cond[1] = 0.4; init_cond[2] = 0.45;
    for (int i = 0; i < 3; ++i) { solver.upper_bound_x()(30 + i) = init_cond[i]; solver.lower_bound_x()(30 + i) = init_cond[i]; }
    solver.solve();
    EXPECT_TRUE(solver.info().status.value == sqp_status_t::SOLVED);
    EXPECT_LT(solver.info().iter, solver.settings().max_iter);
    std::printf("  iterations: cold %d, warm %d; filter entries after cold solve %zu, after warm solve %zu\n", cold, solver.info().iter, kept,
                solver.filter.entries(0).size());
    solver.filter.clear();
    EXPECT_TRUE(solver.filter.entries(0).empty());
}

// ---- user-registered OCPs (device code in libuser_ocp.so, built from tests/cpp/user_ocp.hip) ------------------------------
struct UserRobot { double q = 1.0; };          // must match the layout of the struct registered in user_ocp.hip
struct Pendulum {};
using ApproxA = polympc::Spline<polympc::Chebyshev<6>, 1>;
POLYMPC_FORWARD_DECLARATION(UserRobotOCP, 3, 2, 0, 1, 0, double)
class UserRobotOCP : public polympc::ContinuousOCP<UserRobotOCP, ApproxA, polympc::DENSE> {
public:
    UserRobot device_model() const { return UserRobot(); }
};
POLYMPC_FORWARD_DECLARATION(PendulumOCP, 2, 1, 0, 1, 1, double)
class PendulumOCP : public polympc::ContinuousOCP<PendulumOCP, polympc::Spline<polympc::Chebyshev<5>, 2>, polympc::DENSE> {
public:
    Pendulum device_model() const { return Pendulum(); }
};
POLYMPC_USE_REGISTERED_OCP(UserRobotOCP, UserRobot)
POLYMPC_USE_REGISTERED_OCP(PendulumOCP, Pendulum)

static void UserRegisteredRobotMatchesBuiltin() {
    std::printf("UserRegisteredRobotMatchesBuiltin\n");
    using Builtin = polympc::models::MobileRobot<ApproxA>;
    const int B = 32;
    BatchSolver<UserRobotOCP> us(B); BatchSolver<Builtin> bs(B);
    us.get_problem().set_time_limits(0, 2); bs.get_problem().set_time_limits(0, 2);
    us.settings().max_iter = 10; us.settings().line_search_max_iter = 10; bs.settings() = us.settings();
    for (int b = 0; b < B; ++b) {
        for (int k = 0; k < 7; ++k) {
            us.lower_bound_x(b)[21 + 2 * k] = -1.5; us.upper_bound_x(b)[21 + 2 * k] = 1.5;
            us.lower_bound_x(b)[22 + 2 * k] = -0.75; us.upper_bound_x(b)[22 + 2 * k] = 0.75;
        }
        for (int i = 0; i < 3; ++i) { const double x0 = 0.5 + 0.01 * (b % 7) * (i + 1) - 0.02 * i; us.lower_bound_x(b)[18 + i] = x0; us.upper_bound_x(b)[18 + i] = x0; }
        us.parameters(b)[0] = 2.0 + 0.01 * b;
        std::memcpy(bs.lower_bound_x(b), us.lower_bound_x(b), sizeof(double) * 35);
        std::memcpy(bs.upper_bound_x(b), us.upper_bound_x(b), sizeof(double) * 35);
        bs.parameters(b)[0] = us.parameters(b)[0];
    }
    EXPECT_EQ(us.solve(), PMPC_OK);
    EXPECT_EQ(bs.solve(), PMPC_OK);
    int solved = 0; bool identical = true;
    for (int b = 0; b < B; ++b) {
        solved += us.info(b).status == PMPC_SQP_SOLVED;
        identical = identical && us.info(b).iter == bs.info(b).iter && std::memcmp(us.primal_solution(b), bs.primal_solution(b), sizeof(double) * 35) == 0;
    }
    std::printf("  %d/%d solved, user == builtin bitwise: %s\n", solved, B, identical ? "yes" : "NO");
    EXPECT_TRUE(identical);
    EXPECT_TRUE(solved > B / 2);
}

// SURVEY 8e through the C++ front-end: the batch in three contiguous shards over three contexts driven by three host threads (all on device 0 here;
// one per GPU on a node) against the single-context batch — built-in and user-registered OCP, bit for bit — and the plain-C entry point
static void ShardedBatchSolverMatchesSingleContext() {
    std::printf("ShardedBatchSolverMatchesSingleContext\n");
    using Builtin = polympc::models::MobileRobot<ApproxA>;
    const int B = 37;   // not a multiple of the shard count
    auto fill = [&](auto& s) {
        s.get_problem().set_time_limits(0, 2);
        s.settings().max_iter = 10; s.settings().line_search_max_iter = 10;
        for (int b = 0; b < B; ++b) {
            for (int k = 0; k < 7; ++k) {
                s.lower_bound_x(b)[21 + 2 * k] = -1.5; s.upper_bound_x(b)[21 + 2 * k] = 1.5;
                s.lower_bound_x(b)[22 + 2 * k] = -0.75; s.upper_bound_x(b)[22 + 2 * k] = 0.75;
            }
            for (int i = 0; i < 3; ++i) { const double x0 = 0.5 + 0.013 * (b % 11) * (i + 1) - 0.02 * i; s.lower_bound_x(b)[18 + i] = x0; s.upper_bound_x(b)[18 + i] = x0; }
            s.parameters(b)[0] = 2.0 + 0.01 * b;
        }
    };
    BatchSolver<Builtin> one(B), many(B); BatchSolver<UserRobotOCP> umany(B);
    fill(one); fill(many); fill(umany);
    EXPECT_EQ(many.set_devices({0, 0, 0}), PMPC_OK);
    EXPECT_EQ(umany.set_devices({0, 0}), PMPC_OK);
    EXPECT_EQ(many.num_shards(), 3);
    EXPECT_EQ(one.solve(), PMPC_OK);
    EXPECT_EQ(many.solve(), PMPC_OK);
    EXPECT_EQ(umany.solve(), PMPC_OK);
    bool same = true, usame = true;
    for (int b = 0; b < B; ++b) {
        same = same && one.info(b).iter == many.info(b).iter && one.info(b).status == many.info(b).status &&
               std::memcmp(one.primal_solution(b), many.primal_solution(b), sizeof(double) * 35) == 0 &&
               std::memcmp(one.dual_solution(b), many.dual_solution(b), sizeof(double) * 56) == 0;
        usame = usame && one.info(b).iter == umany.info(b).iter && std::memcmp(one.primal_solution(b), umany.primal_solution(b), sizeof(double) * 35) == 0;
    }
    std::printf("  3 shards == 1 context bitwise: %s; user OCP in 2 shards: %s\n", same ? "yes" : "NO", usame ? "yes" : "NO");
    EXPECT_TRUE(same); EXPECT_TRUE(usame);
    // the C entry point: two contexts, two host threads inside the library
    pmpc_context* c2[2] = {nullptr, nullptr};
    EXPECT_EQ(pmpc_create(0, nullptr, &c2[0]), PMPC_OK); EXPECT_EQ(pmpc_create(0, nullptr, &c2[1]), PMPC_OK);
    pmpc_sqp_settings ss; pmpc_sqp_settings_default(&ss); ss.max_iter = 10; ss.line_search_max_iter = 10;
    pmpc_qp_settings qs; pmpc_qp_settings_sqp_default(&qs);
    const double mp[3] = {1.0, 1.0, 1.0};
    std::vector<double> x((size_t)B * 35), lam((size_t)B * 56); std::vector<pmpc_sqp_info> inf(B);
    EXPECT_EQ(pmpc_sqp_solve_batch_multi(c2, 2, PMPC_MODEL_ROBOT, 6, 1, 0.0, 2.0, mp, 3, B, nullptr, nullptr, many.parameters(0), many.lower_bound_x(0),
                                         many.upper_bound_x(0), nullptr, nullptr, &ss, &qs, x.data(), lam.data(), inf.data()), PMPC_OK);
    bool csame = true;
    for (int b = 0; b < B; ++b) csame = csame && inf[b].iter == one.info(b).iter && std::memcmp(&x[(size_t)b * 35], one.primal_solution(b), sizeof(double) * 35) == 0;
    std::printf("  pmpc_sqp_solve_batch_multi over 2 contexts == 1 context bitwise: %s; route of the shard launches: %d\n", csame ? "yes" : "NO", pmpc_sqp_last_route(c2[0]));
    EXPECT_TRUE(csame);
    EXPECT_EQ(pmpc_sqp_last_route(c2[0]), (int)PMPC_ROUTE_REG1);
    // a copied solver is a solver (the reference's SQPBase is copyable): same data, no shared device contexts, same results
    BatchSolver<Builtin> copy(many);
    EXPECT_EQ(copy.num_shards(), 1);
    for (int b = 0; b < B; ++b) for (int i = 0; i < 35; ++i) copy.primal_solution(b)[i] = 0.0;
    for (int b = 0; b < B; ++b) for (int i = 0; i < 56; ++i) copy.dual_solution(b)[i] = 0.0;
    EXPECT_EQ(copy.solve(), PMPC_OK);
    Solver<Builtin> single; single.settings().max_iter = 7; single.settings().kkt_form = 1;
    Solver<Builtin> single_copy(single); Solver<Builtin> assigned; assigned = single;
    EXPECT_EQ(single_copy.settings().max_iter, 7); EXPECT_EQ(assigned.settings().kkt_form, 1);
    bool copysame = true;
    for (int b = 0; b < B; ++b) copysame = copysame && copy.info(b).iter == one.info(b).iter && std::memcmp(copy.primal_solution(b), one.primal_solution(b), sizeof(double) * 35) == 0;
    std::printf("  copy of a sharded solver, re-solved on the thread's context == original bitwise: %s\n", copysame ? "yes" : "NO");
    EXPECT_TRUE(copysame);
    // the filter line search without a carried filter is available in shards (the C entry's rule)
    many.settings().line_search = 1; one.settings().line_search = 1;
    for (int b = 0; b < B; ++b) for (int i = 0; i < 35; ++i) { many.primal_solution(b)[i] = 0.0; one.primal_solution(b)[i] = 0.0; }
    for (int b = 0; b < B; ++b) for (int i = 0; i < 56; ++i) { many.dual_solution(b)[i] = 0.0; one.dual_solution(b)[i] = 0.0; }
    EXPECT_EQ(many.solve(), PMPC_OK); EXPECT_EQ(one.solve(), PMPC_OK);
    bool fsame = true;
    for (int b = 0; b < B; ++b) fsame = fsame && many.info(b).iter == one.info(b).iter && std::memcmp(many.primal_solution(b), one.primal_solution(b), sizeof(double) * 35) == 0;
    std::printf("  filter line search in 3 shards == 1 context (fresh filters) bitwise: %s\n", fsame ? "yes" : "NO");
    EXPECT_TRUE(fsame);
    // the same context twice would put two host threads on one stream and workspace: rejected
    pmpc_context* dup[2] = {c2[0], c2[0]};
    EXPECT_EQ(pmpc_sqp_solve_batch_multi(dup, 2, PMPC_MODEL_ROBOT, 6, 1, 0.0, 2.0, mp, 3, B, nullptr, nullptr, many.parameters(0), many.lower_bound_x(0),
                                         many.upper_bound_x(0), nullptr, nullptr, &ss, &qs, x.data(), lam.data(), inf.data()), PMPC_ERR_INVALID_ARGUMENT);
    ss.iteration_trace = x.data();   // device state of one context: rejected
    EXPECT_EQ(pmpc_sqp_solve_batch_multi(c2, 2, PMPC_MODEL_ROBOT, 6, 1, 0.0, 2.0, mp, 3, B, nullptr, nullptr, many.parameters(0), many.lower_bound_x(0),
                                         many.upper_bound_x(0), nullptr, nullptr, &ss, &qs, x.data(), lam.data(), inf.data()), PMPC_ERR_INVALID_ARGUMENT);
    pmpc_destroy(c2[0]); pmpc_destroy(c2[1]);
}

static void UserPendulumWithPathConstraint() {
    std::printf("UserPendulumWithPathConstraint\n");
    Solver<PendulumOCP> solver;
    solver.get_problem().set_time_limits(0, 2);
    solver.settings().max_iter = 30; solver.settings().line_search_max_iter = 10;
    const int nn = PendulumOCP::NUM_NODES;
    solver.parameters()(0) = 1.0;
    solver.lower_bound_x()(2 * nn - 2) = 0.5; solver.upper_bound_x()(2 * nn - 2) = 0.5;   // theta(0) = 0.5
    solver.lower_bound_x()(2 * nn - 1) = 0.0; solver.upper_bound_x()(2 * nn - 1) = 0.0;   // omega(0) = 0
    for (int k = 0; k < nn; ++k) { solver.lower_bound_x()(2 * nn + k) = -3.0; solver.upper_bound_x()(2 * nn + k) = 3.0; }
    for (int k = 0; k < nn; ++k) { solver.lower_bound_g()(k) = -1.0; solver.upper_bound_g()(k) = 1.0; }
    solver.solve();
    std::printf("  status %d iter %d violation %.2e cost %.5f\n", (int)solver.info().status.value, solver.info().iter, solver.constr_violation(), solver.cost());
    EXPECT_TRUE(solver.info().status.value == sqp_status_t::SOLVED);
    EXPECT_TRUE(solver.constr_violation() <= 1e-3);
    for (int k = 0; k < nn; ++k) {   // the path constraint holds at every node
        const double g = solver.primal_solution()(2 * k + 1) + 0.5 * solver.primal_solution()(2 * nn + k);
        EXPECT_TRUE(g <= 1.0 + 2e-3 && g >= -1.0 - 2e-3);
    }
}

int main() {
    if (!polympc::context()) { std::printf("no GPU: %s\n", pmpc_status_string(polympc::last_error())); return 77; }
    box_admmSimpleQP();
    box_admmSinglePrecisionFloat();
    admmSinglePrecisionFloat();
    box_admmRuizEquilibration();
    ValetParkingTest();
    admmSimpleQP();
    box_admmNonConvex();
    MPCWrapperTest();
    IterationCallbackTest();
    BatchMPCTest();
    UserRegisteredRobotMatchesBuiltin();
    ShardedBatchSolverMatchesSingleContext();
    UserPendulumWithPathConstraint();
    std::printf(failures ? "FAILED (%d)\n" : "ALL PASSED\n", failures);
    return failures ? 1 : 0;
}
