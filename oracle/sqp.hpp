// ORACLE — TEST INFRASTRUCTURE ONLY (CPU restatement of the reference algorithm).
//
// Line-search SQP driver, damped BFGS and the two shipped Hessian-regularisation policies.
// Follows /root/reference/src/solvers/sqp_base.hpp (settings :24-47, info :57-61, ctor QP overrides :83-90,
// step_size_selection_impl :380-419, constraints_violation_impl :423-444, max_constraints_violation_impl
// :448-474, update_linearisation_dense_impl :490-504, termination_criteria_impl :524-529, solve_qp :533-565,
// solve :569-696), /root/reference/src/solvers/bfgs.hpp:23-52, and the regularisers at
// tests/solvers/sqp/sqp_test_autodiff.cpp:29-45 (eigenvalue mirroring) and
// tests/control/dense_sparse_compare.cpp:109-122 (Gershgorin shift).
//
// Problem concept (ContinuousOCP in ocp.hpp, GenericNLP in nlp.hpp):
//   int VAR_SIZE, NUM_EQ, NUM_INEQ;
//   cost(x,d,c); equalities(x,d,c); inequalities(x,d,g);
//   lagrangian_gradient(x,d,lam,lag,lag_grad,cost_grad,g,jac);
//   lagrangian_gradient_hessian(x,d,lam,lag,lag_grad,H,cost_grad,g,jac);
#pragma once
#include <cmath>
#include <limits>
#include <utility>
#include <vector>
#include "qp.hpp"
#include "ruiz.hpp"
#include "admm.hpp"

namespace oracle {

// bfgs.hpp:23-52 ; B is n x n column-major
inline void BFGS_update(double* B, const double* s, const double* y, int n) {
    std::vector<double> Bs(n), r(n);
    for (int i = 0; i < n; ++i) { double a = 0; for (int j = 0; j < n; ++j) a += B[i + j * n] * s[j]; Bs[i] = a; }
    double sBs = 0, sy = 0;
    for (int i = 0; i < n; ++i) sBs += s[i] * Bs[i];
    for (int i = 0; i < n; ++i) sy += s[i] * y[i];
    double sr;
    if (sy < 0.2 * sBs) {
        const double theta = 0.8 * sBs / (sBs - sy);
        for (int i = 0; i < n; ++i) r[i] = theta * y[i] + (1 - theta) * Bs[i];
        sr = theta * sy + (1 - theta) * sBs;
    } else {
        for (int i = 0; i < n; ++i) r[i] = y[i];
        sr = sy;
    }
    if (sr < std::numeric_limits<double>::epsilon()) return;
    for (int j = 0; j < n; ++j) for (int i = 0; i < n; ++i) B[i + j * n] += (-Bs[i] * Bs[j]) / sBs;
    for (int j = 0; j < n; ++j) for (int i = 0; i < n; ++i) B[i + j * n] += (r[i] * r[j]) / sr;
}

// Jacobi eigen-decomposition of a symmetric matrix (stand-in for Eigen::EigenSolver on a symmetric H) in the ROUND-ROBIN order of the kernel (pmpc_sqp.hpp regularise_eig_mirror,
// late round 6): a tournament over np = n (+ 1 bye when n is odd) players; in round r pair 0 is (np - 1, r), pair i >= 1 is ((r + i) mod (np - 1), (r - i) mod (np - 1)); the pairs
// of a round are disjoint and rotate together — angles from the matrix at the start of the round, then every pair mixes its two COLUMNS of A and V over all rows, then every pair
// its two ROWS of A over all columns, then the pair's own off-diagonal entry is set to the exact zero. Stops when max |a_ij|^2 < 1e-300 (i != j): ~10 sweeps.
// n <= JACOBI_CYCLIC_MAX (the sizes of the reference's own uses: the 2- and 4-variable NLPs of sqp_test_autodiff.cpp) keeps the pair-by-pair cyclic iteration of rounds 1 .. 6, bit for bit:
// HS071's Hessian has an eigenvalue that is zero in exact arithmetic, the rule `w <= 0 -> -w + 0.1` is discontinuous there, and the known-answer tests of that problem were pinned
// with the rounding of that iteration (test_sqp_hs071_iteration_bound_is_a_last_bit_property says what that means). The kernel runs the round-robin order at every size; no OCP grid has 8 variables or fewer
// (3 nodes of a one-state, one-input model have 6: such a model would need this switch in the kernel too).
constexpr int JACOBI_CYCLIC_MAX = 8;
inline void jacobi_eig_cyclic(std::vector<double> A, int n, std::vector<double>& w, std::vector<double>& V) {
    V.assign(n * n, 0.0); for (int i = 0; i < n; ++i) V[i + i * n] = 1.0;
    for (int sweep = 0; sweep < 100; ++sweep) {
        double off = 0; for (int i = 0; i < n; ++i) for (int j = 0; j < i; ++j) off += A[i + j * n] * A[i + j * n];
        if (off < 1e-300) break;
        for (int p = 0; p < n; ++p)
            for (int q = p + 1; q < n; ++q) {
                double apq = A[p + q * n]; if (std::fabs(apq) < 1e-300) continue;
                double theta = (A[q + q * n] - A[p + p * n]) / (2 * apq);
                double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1));
                double c = 1 / std::sqrt(t * t + 1), s = t * c;
                for (int k = 0; k < n; ++k) { double akp = A[k + p * n], akq = A[k + q * n]; A[k + p * n] = c * akp - s * akq; A[k + q * n] = s * akp + c * akq; }
                for (int k = 0; k < n; ++k) { double apk = A[p + k * n], aqk = A[q + k * n]; A[p + k * n] = c * apk - s * aqk; A[q + k * n] = s * apk + c * aqk; }
                for (int k = 0; k < n; ++k) { double vkp = V[k + p * n], vkq = V[k + q * n]; V[k + p * n] = c * vkp - s * vkq; V[k + q * n] = s * vkp + c * vkq; }
            }
    }
    w.resize(n); for (int i = 0; i < n; ++i) w[i] = A[i + i * n];
}
inline void jacobi_eig(std::vector<double> A, int n, std::vector<double>& w, std::vector<double>& V) {
    if (n <= JACOBI_CYCLIC_MAX) { jacobi_eig_cyclic(A, n, w, V); return; }
    V.assign(n * n, 0.0); for (int i = 0; i < n; ++i) V[i + i * n] = 1.0;
    const int np = n + (n & 1), m2 = np / 2, nr = np - 1;
    auto pair_of = [&](int r, int i, int& p, int& q) {
        const int a = (i == 0) ? np - 1 : (r + i) % nr, b = (i == 0) ? r : (r - i + nr) % nr;
        p = a < b ? a : b; q = a < b ? b : a;
    };
    std::vector<double> cs(2 * m2);
    for (int sweep = 0; sweep < 100; ++sweep) {
        double amax = 0; for (int j = 0; j < n; ++j) for (int i = j + 1; i < n; ++i) amax = std::fmax(amax, std::fabs(A[i + j * n]));
        if (amax * amax < 1e-300) break;
        for (int r = 0; r < nr; ++r) {
            for (int i = 0; i < m2; ++i) {
                int p, q; pair_of(r, i, p, q);
                double c = 1.0, s = 0.0;
                if (q < n) {
                    const double apq = A[p + q * n];
                    if (!(std::fabs(apq) < 1e-300)) {
                        const double theta = (A[q + q * n] - A[p + p * n]) / (2 * apq);
                        const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1));
                        c = 1 / std::sqrt(t * t + 1); s = t * c;
                    }
                }
                cs[i] = c; cs[m2 + i] = s;
            }
            for (int i = 0; i < m2; ++i) {
                int p, q; pair_of(r, i, p, q);
                const double c = cs[i], s = cs[m2 + i];
                if (!(q < n && s != 0.0)) continue;
                for (int k = 0; k < n; ++k) { double akp = A[k + p * n], akq = A[k + q * n]; A[k + p * n] = c * akp - s * akq; A[k + q * n] = s * akp + c * akq; }
                for (int k = 0; k < n; ++k) { double vkp = V[k + p * n], vkq = V[k + q * n]; V[k + p * n] = c * vkp - s * vkq; V[k + q * n] = s * vkp + c * vkq; }
            }
            for (int i = 0; i < m2; ++i) {
                int p, q; pair_of(r, i, p, q);
                const double c = cs[i], s = cs[m2 + i];
                if (!(q < n && s != 0.0)) continue;
                for (int k = 0; k < n; ++k) { double apk = A[p + k * n], aqk = A[q + k * n]; A[p + k * n] = c * apk - s * aqk; A[q + k * n] = s * apk + c * aqk; }
            }
            for (int i = 0; i < m2; ++i) {   // the annihilated entries as exact zeros (what the mixes leave there is the rounding of the diagonal entries, which never decays)
                int p, q; pair_of(r, i, p, q);
                if (q < n && cs[m2 + i] != 0.0) { A[p + q * n] = 0.0; A[q + p * n] = 0.0; }
            }
        }
    }
    w.resize(n); for (int i = 0; i < n; ++i) w[i] = A[i + i * n];
}

enum regularisation { REG_NONE = 0, REG_EIG_MIRROR = 1, REG_GERSHGORIN = 2 };

// sqp_test_autodiff.cpp:29-45
inline void regularise_eig_mirror(double* H, int n) {
    std::vector<double> w, V; jacobi_eig(std::vector<double>(H, H + n * n), n, w, V);
    double mn = w[0]; for (int i = 1; i < n; ++i) mn = std::fmin(mn, w[i]);
    if (mn <= 0) {
        for (int i = 0; i < n; ++i) if (w[i] <= 0) w[i] = -1 * w[i] + 0.1;
        for (int j = 0; j < n; ++j) for (int i = 0; i < n; ++i) {
            double a = 0; for (int k = 0; k < n; ++k) a += (V[i + k * n] * w[k]) * V[j + k * n];
            H[i + j * n] = a;
        }
    }
}
// dense_sparse_compare.cpp:109-122
inline void regularise_gershgorin(double* H, int n) {
    for (int i = 0; i < n; ++i) {
        double aii = H[i + i * n];
        double ri = 0; for (int k = 0; k < n; ++k) ri += std::fabs(H[k + i * n]);
        ri -= std::fabs(aii);
        if (aii - ri <= 0) H[i + i * n] += (ri - aii) + 0.01;
    }
}

struct sqp_settings {  // sqp_base.hpp:24-47
    double tau = 0.5, eta = 0.25, rho = 0.5, eps_prim = 1e-3, eps_dual = 1e-3;
    int max_iter = 100, line_search_max_iter = 100;
    int regularisation = REG_NONE;          // hook of :277-306 (default no-op)
    bool exact_hessian_every_iter = false;  // override used by codegen_test.cpp:381-398 / minimal_time_test.cpp:126-133
    int preconditioner = 0;                 // SQPBase's Preconditioner template argument: 0 IdentityPreconditioner (default), 1 RuizEquilibration
    int qp_solver = 0;                      // SQPBase's QPSolver template argument: 0 boxADMM (box_admm.hpp, default), 1 ADMM (admm.hpp, OSQP form)
    int hessian_update = 0;                 // hessian_update_impl: 0 damped BFGS on the whole matrix (bfgs.hpp, DENSE default), 1 the block BFGS of
                                            // ContinuousOCP (continuous_ocp.hpp:2304-2431), which keeps the Hessian block-diagonal per node
    int line_search = 0;                    // step_size_selection_impl: 0 l1-merit backtracking (:380-419, default), 1 the filter line search the
                                            // reference's valet-parking test plugs in (valet_parking_mpc_test.cpp:116-158) on LSFilter
    int filter_max_depth = 10;              // LSFilter::max_depth (line_search.hpp:38)
    double filter_beta = 1e-5;              // LSFilter::beta (line_search.hpp:39)
};

// LSFilter, /root/reference/src/solvers/line_search.hpp:31-98: a list of (cost, constraint violation) pairs, newest first.
// State layout shared with the GPU path: st[0] = number of pairs, then the pairs front to back (FILTER_CAPACITY at most).
struct LSFilter {
    static constexpr int CAPACITY = 10, STATE_DOUBLES = 1 + 2 * CAPACITY;
    std::vector<std::pair<double, double>> f;   // front = index 0
    int max_depth = 10;
    double beta = 1e-5;
    void clear() { f.clear(); }
    bool is_acceptable(double cost, double constraint) const {   // :65-74
        for (const auto& e : f)
            if (((e.first - beta * e.second) <= cost) && ((e.second - beta * e.second) <= constraint)) return false;
        return true;
    }
    void add(double cost, double constraint) {                   // :76-92
        if ((int)f.size() < max_depth) {
            std::vector<std::pair<double, double>> keep;             // remove_if(dominated_by(cost, constraint)) :14-29, order preserved
            for (const auto& e : f) if (!((e.first >= cost) && (e.second >= constraint))) keep.push_back(e);
            f.swap(keep);
            f.insert(f.begin(), std::make_pair(cost, constraint));
        } else {
            f.pop_back();
            f.insert(f.begin(), std::make_pair(cost, constraint));
        }
    }
    void load(const double* st) { f.clear(); const int k = (int)st[0]; for (int i = 0; i < k && i < CAPACITY; ++i) f.emplace_back(st[1 + 2 * i], st[2 + 2 * i]); }
    void store(double* st) const {
        for (int i = 0; i < STATE_DOUBLES; ++i) st[i] = 0.0;
        st[0] = (double)f.size();
        for (size_t i = 0; i < f.size(); ++i) { st[1 + 2 * i] = f[i].first; st[2 + 2 * i] = f[i].second; }
    }
};
enum sqp_status { SQP_SOLVED = 0, SQP_MAX_ITER_EXCEEDED = 1, SQP_INVALID_SETTINGS = 2, SQP_REDO = 4 /* internal: a QP of the condensed orders gave up at its conditioning gate — the driver re-solves the instance in the full KKT form */ };
struct sqp_info { int iter = 0, qp_solver_iter = 0, status = SQP_MAX_ITER_EXCEEDED, flags = 0; };

template <class Problem>
struct SQP {
    static constexpr double EPSILON = std::numeric_limits<double>::epsilon();
    static constexpr double INF = std::numeric_limits<double>::infinity();

    Problem& problem;
    int n, me, mi, m;
    std::vector<double> H, h, x, lam, lam_k, A, al, au, p_static, lbx, ubx, lx, ux, lbg, ubg, lag_gradient, step_prev;
    double cost_ = 0, primal_norm = 0, dual_norm = 0, max_violation = 0;
    // iteration records: what sqp_settings_t::iteration_callback (sqp_base.hpp:33, called at :685-686 from the second iteration on) could read,
    // kept per iteration — [iter, alpha, primal_norm, dual_norm, cost, qp iterations, qp status, max violation] after each termination test
    double alpha_last = 0; int qp_iter_last = 0, qp_status_last = 0;
    double* trace = nullptr; int trace_capacity = 0;
    void record() {
        if (!trace || info.iter > trace_capacity) return;
        double* r = trace + (size_t)(info.iter - 1) * 8;
        r[0] = info.iter; r[1] = alpha_last; r[2] = primal_norm; r[3] = dual_norm; r[4] = cost_; r[5] = qp_iter_last; r[6] = qp_status_last; r[7] = max_violation;
    }
    sqp_settings settings;
    sqp_info info;
    BoxADMM qp;
    ADMM qp_admm;   // used when settings.qp_solver == 1 (same settings and pivot policy as qp)
    // optional trace of every QP handed to the QP solver (used to build QP replay batches)
    bool record_qps = false;
    struct qp_record { std::vector<double> H, h, A, al, au, lx, ux; };
    std::vector<qp_record> qp_trace;

    SQP(Problem& prob, int nd)
        : problem(prob), n(prob.VAR_SIZE), me(prob.NUM_EQ), mi(prob.NUM_INEQ), m(me + mi), qp(n, m), qp_admm(n, m) {
        H.assign(n * n, 0); h.assign(n, 0); x.assign(n, 0); lam.assign(m + n, 0); lam_k.assign(m + n, 0);
        A.assign(m * n, 0); al.assign(m, 0); au.assign(m, 0); p_static.assign(nd > 0 ? nd : 1, 0);
        lbx.assign(n, -INF); ubx.assign(n, INF); lx.assign(n, 0); ux.assign(n, 0);
        lbg.assign(mi, -INF); ubg.assign(mi, INF); lag_gradient.assign(n, 0); step_prev.assign(n, 0);
        // sqp_base.hpp:83-90
        qp.settings.warm_start = false; qp.settings.check_termination = 10;
        qp.settings.eps_abs = 1e-4; qp.settings.eps_rel = 1e-4; qp.settings.max_iter = 100;
        qp.settings.adaptive_rho = true; qp.settings.adaptive_rho_interval = 50; qp.settings.alpha = 1.0;
    }

    double constraints_violation(const double* xx) const {  // :423-444
        double cl1 = EPSILON;
        std::vector<double> c(me), g(mi);
        problem.equalities(xx, p_static.data(), c.data());
        double s = 0; for (int i = 0; i < me; ++i) s += std::fabs(c[i]); cl1 += s;
        problem.inequalities(xx, p_static.data(), g.data());
        s = 0; for (int i = 0; i < mi; ++i) s += std::fmax(lbg[i] - g[i], 0.0); cl1 += s;
        s = 0; for (int i = 0; i < mi; ++i) s += std::fmax(g[i] - ubg[i], 0.0); cl1 += s;
        s = 0; for (int i = 0; i < n; ++i) s += std::fmax(lbx[i] - xx[i], 0.0); cl1 += s;
        s = 0; for (int i = 0; i < n; ++i) s += std::fmax(xx[i] - ubx[i], 0.0); cl1 += s;
        return cl1;
    }
    double max_constraints_violation(const double* xx) const {  // :448-474
        double c = 0;
        if (me > 0) { std::vector<double> ce(me); problem.equalities(xx, p_static.data(), ce.data()); c = BoxADMM::inf_norm(ce.data(), me); }
        if (mi > 0) {
            std::vector<double> g(mi); problem.inequalities(xx, p_static.data(), g.data());
            double a = -INF, b = -INF;
            for (int i = 0; i < mi; ++i) { a = std::fmax(a, lbg[i] - g[i]); b = std::fmax(b, g[i] - ubg[i]); }
            c = std::fmax(c, a); c = std::fmax(c, b);
        }
        double a = -INF, b = -INF;
        for (int i = 0; i < n; ++i) { a = std::fmax(a, lbx[i] - xx[i]); b = std::fmax(b, xx[i] - ubx[i]); }
        c = std::fmax(c, a); c = std::fmax(c, b);
        return c;
    }

    LSFilter filter;   // member of the reference's test solver: it outlives solve() (valet_parking_mpc_test.cpp:114)

    // step_size_selection_impl of valet_parking_mpc_test.cpp:116-158
    double step_size_selection_filter(const double* p) {
        const double tau = settings.tau;
        filter.max_depth = settings.filter_max_depth; filter.beta = settings.filter_beta;
        double constr_l1 = constraints_violation(x.data());
        double cost_1; problem.cost(x.data(), p_static.data(), cost_1);
        if (filter.is_acceptable(cost_1, constr_l1)) filter.add(cost_1, constr_l1);
        double alpha = 1.0, cost_step;
        std::vector<double> x_step(n);
        for (int i = 1; i < settings.line_search_max_iter; i++) {
            for (int j = 0; j < n; ++j) { x_step[j] = alpha * p[j]; x_step[j] += x[j]; }
            problem.cost(x_step.data(), p_static.data(), cost_step);
            const double constr_step = constraints_violation(x_step.data());
            cost_ = cost_step;
            if (filter.is_acceptable(cost_step, constr_step)) { filter.add(cost_step, constr_step); return alpha; }
            else alpha *= tau;
        }
        return alpha;
    }

    double step_size_selection(const double* p) {  // :380-419
        if (settings.line_search == 1) return step_size_selection_filter(p);
        const double tau = settings.tau;
        double constr_l1 = constraints_violation(x.data());
        double mu = BoxADMM::inf_norm(lam_k.data(), m + n);
        double cost_1; problem.cost(x.data(), p_static.data(), cost_1);
        double phi_l1 = cost_1 + mu * constr_l1;
        double gp = 0; for (int i = 0; i < n; ++i) gp += h[i] * p[i];
        double Dp_phi_l1 = gp - mu * constr_l1;
        double alpha = 1.0, cost_step;
        std::vector<double> x_step(n);
        for (int i = 1; i < settings.line_search_max_iter; i++) {
            for (int j = 0; j < n; ++j) { x_step[j] = alpha * p[j]; x_step[j] += x[j]; }
            problem.cost(x_step.data(), p_static.data(), cost_step);
            cost_ = cost_step;
            double phi_l1_step = cost_step + mu * constraints_violation(x_step.data());
            if (phi_l1_step <= (phi_l1 + alpha * settings.eta * Dp_phi_l1)) return alpha;
            else alpha = tau * alpha;
        }
        return alpha;
    }

    void hessian_regularisation() {
        if (settings.regularisation == REG_EIG_MIRROR) regularise_eig_mirror(H.data(), n);
        else if (settings.regularisation == REG_GERSHGORIN) regularise_gershgorin(H.data(), n);
    }
    void linearisation() {  // linearisation_dense_impl :310-318
        double lag = 0;
        problem.lagrangian_gradient_hessian(x.data(), p_static.data(), lam.data(), lag, lag_gradient.data(), H.data(),
                                            h.data(), al.data(), A.data());
        hessian_regularisation();
    }
    void update_linearisation() {  // :490-504
        if (settings.exact_hessian_every_iter) { linearisation(); return; }
        double lag; std::vector<double> lag_grad(n), yk(n);
        problem.lagrangian_gradient(x.data(), p_static.data(), lam.data(), lag, lag_grad.data(), h.data(), al.data(), A.data());
        for (int i = 0; i < n; ++i) yk[i] = lag_grad[i] - lag_gradient[i];
        if constexpr (Problem::HAS_BLOCK_BFGS) {
            if (settings.hessian_update == 1) problem.hessian_update_block(H.data(), step_prev.data(), yk.data());
            else BFGS_update(H.data(), step_prev.data(), yk.data(), n);
        } else BFGS_update(H.data(), step_prev.data(), yk.data(), n);
        lag_gradient = lag_grad;
    }
    void form_qp_bounds() {  // :588-593
        for (int i = 0; i < m; ++i) { al[i] = -al[i]; au[i] = al[i]; }
        for (int i = 0; i < mi; ++i) { al[me + i] += lbg[i]; au[me + i] += ubg[i]; }
        for (int i = 0; i < n; ++i) { lx[i] = lbx[i] - x[i]; ux[i] = ubx[i] - x[i]; }
    }
    void solve_qp(std::vector<double>& p, std::vector<double>& p_lambda) {  // :533-565
        // m_preconditioner.compute / solve_qp / unscale(p, p_lambda) / unscale(H, h, A, ...): sqp_base.hpp:605-611, :661-665.
        // The matrices are scaled and unscaled IN PLACE every iteration, as in the reference (H carries the rounding).
        Ruiz ruiz(n, m);
        if (settings.preconditioner == 1) ruiz.compute(H.data(), h.data(), A.data(), al.data(), au.data(), lx.data(), ux.data());
        if (record_qps) qp_trace.push_back({H, h, A, al, au, lx, ux});
        if (settings.qp_solver == 1) {   // Solver<Problem, ADMM<...>>: zero guesses as in the 7-argument form (admm.hpp:104-109)
            qp_admm.settings = qp.settings; qp_admm.pivot = qp.pivot;
            qp_admm.solve(H.data(), h.data(), A.data(), al.data(), au.data(), lx.data(), ux.data(), nullptr, nullptr);
            info.qp_solver_iter += qp_admm.info.iter; qp_iter_last = qp_admm.info.iter; qp_status_last = qp_admm.info.status;
            p = qp_admm.x; p_lambda = qp_admm.y;
        } else {
            qp.solve(H.data(), h.data(), A.data(), al.data(), au.data(), lx.data(), ux.data());
            info.qp_solver_iter += qp.info.iter; qp_iter_last = qp.info.iter; qp_status_last = qp.info.status;
            info.flags |= qp.info.flags;
            qp_gave_up = qp.gives_up();
            p = qp.x; p_lambda = qp.y;
        }
        if (settings.preconditioner == 1) {
            ruiz.unscale(p.data(), p_lambda.data());
            ruiz.unscale(H.data(), h.data(), A.data(), al.data(), au.data(), lx.data(), ux.data());
        }
    }
    bool termination_criteria() {  // :524-529
        max_violation = max_constraints_violation(x.data());
        return (primal_norm <= settings.eps_prim) && (dual_norm <= settings.eps_dual) && (max_violation <= settings.eps_prim);
    }

    void iterate_tail(std::vector<double>& p, std::vector<double>& p_lambda) {
        lam_k = p_lambda;
        for (int i = 0; i < m + n; ++i) p_lambda[i] -= lam[i];
        const double alpha = step_size_selection(p.data());
        alpha_last = alpha;
        for (int i = 0; i < n; ++i) x[i] += alpha * p[i];
        for (int i = 0; i < m + n; ++i) lam[i] += alpha * p_lambda[i];
        for (int i = 0; i < n; ++i) step_prev[i] = alpha * p[i];
        primal_norm = alpha * BoxADMM::inf_norm(p.data(), n);
        dual_norm = alpha * BoxADMM::inf_norm(p_lambda.data(), m + n);
    }

    bool qp_gave_up = false;
    void solve() {  // :569-696
        info.status = SQP_MAX_ITER_EXCEEDED;
        std::vector<double> p(n), p_lambda(m + n, 0.0);
        info.qp_solver_iter = 0;
        info.iter = 1;
        info.flags = 0; qp_gave_up = false;
        linearisation();
        form_qp_bounds();
        solve_qp(p, p_lambda);
        if (qp_gave_up) { info.status = SQP_REDO; return; }
        iterate_tail(p, p_lambda);
        { const bool done = termination_criteria(); record(); if (done) { info.status = SQP_SOLVED; return; } }
        while (info.iter < settings.max_iter) {
            info.iter++;
            update_linearisation();
            form_qp_bounds();
            solve_qp(p, p_lambda);
            if (qp_gave_up) { info.status = SQP_REDO; return; }
            iterate_tail(p, p_lambda);
            const bool done = termination_criteria(); record();
            if (done) { info.status = SQP_SOLVED; break; }
        }
    }
    void solve(const double* x_guess, const double* lam_guess) {
        for (int i = 0; i < n; ++i) x[i] = x_guess[i];
        for (int i = 0; i < m + n; ++i) lam[i] = lam_guess[i];
        solve();
    }
};

}  // namespace oracle
