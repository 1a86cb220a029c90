// ORACLE — TEST INFRASTRUCTURE ONLY (CPU restatement of the reference algorithm).
// C entry points over the header-only restatement; see oracle_capi.h.
#include "qp_f32.hpp"
#include "oracle_capi.h"
#include <cstring>
#include <vector>
#include "cheb.hpp"
#include "models.hpp"
#include "nlp.hpp"
#include "ocp.hpp"
#include "qp.hpp"
#include "admm.hpp"
#include "sqp.hpp"

using namespace oracle;

static qp_settings to_qp(const orc_qp_settings* s) {
    qp_settings q;
    q.eps_rel = s->eps_rel; q.eps_abs = s->eps_abs; q.max_iter = s->max_iter;
    q.rho = s->rho; q.sigma = s->sigma; q.alpha = s->alpha; q.check_termination = s->check_termination;
    q.adaptive_rho = s->adaptive_rho != 0; q.adaptive_rho_tolerance = s->adaptive_rho_tolerance;
    q.adaptive_rho_interval = s->adaptive_rho_interval;
    return q;
}
static void from_qp(const qp_settings& q, orc_qp_settings* s) {
    s->eps_rel = q.eps_rel; s->eps_abs = q.eps_abs; s->max_iter = q.max_iter; s->rho = q.rho; s->sigma = q.sigma;
    s->alpha = q.alpha; s->check_termination = q.check_termination; s->adaptive_rho = q.adaptive_rho ? 1 : 0;
    s->adaptive_rho_tolerance = q.adaptive_rho_tolerance; s->adaptive_rho_interval = q.adaptive_rho_interval;
}
static sqp_settings to_sqp(const orc_sqp_settings* s) {
    sqp_settings q;
    q.tau = s->tau; q.eta = s->eta; q.rho = s->rho; q.eps_prim = s->eps_prim; q.eps_dual = s->eps_dual;
    q.max_iter = s->max_iter; q.line_search_max_iter = s->line_search_max_iter;
    q.regularisation = s->regularisation; q.exact_hessian_every_iter = s->exact_hessian_every_iter != 0;
    q.preconditioner = s->preconditioner; q.hessian_update = s->hessian_update; q.qp_solver = s->qp_solver;
    q.line_search = s->line_search; q.filter_max_depth = s->filter_max_depth; q.filter_beta = s->filter_beta;
    return q;
}

template <class Model> static Model make_model(const double*, int) { return Model(); }
template <> RobotOCP make_model<RobotOCP>(const double* mp, int nmp) {
    RobotOCP r;
    if (nmp >= 1) for (int i = 0; i < 3; ++i) r.Q[i] = mp[0];
    if (nmp >= 2) for (int i = 0; i < 2; ++i) r.R[i] = mp[1];
    if (nmp >= 3) for (int i = 0; i < 3; ++i) r.QN[i] = mp[2];
    return r;
}
template <> RobotNGOCP make_model<RobotNGOCP>(const double* mp, int nmp) {
    RobotNGOCP r;
    if (nmp >= 1) for (int i = 0; i < 3; ++i) r.Q[i] = mp[0];
    if (nmp >= 2) for (int i = 0; i < 2; ++i) r.R[i] = mp[1];
    if (nmp >= 3) for (int i = 0; i < 3; ++i) r.QN[i] = mp[2];
    return r;
}

#define DISPATCH_MODEL(model, F, ...)                                   \
    switch (model) {                                                    \
        case ORC_MODEL_ROBOT: F<RobotOCP>(__VA_ARGS__); break;          \
        case ORC_MODEL_CSTR: F<CstrOCP>(__VA_ARGS__); break;            \
        case ORC_MODEL_PARKING: F<ParkingOCP>(__VA_ARGS__); break;      \
        case ORC_MODEL_ROBOT_NG: F<RobotNGOCP>(__VA_ARGS__); break;     \
        case ORC_MODEL_KITE_STANDIN: F<KiteStandInOCP>(__VA_ARGS__); break; \
        case ORC_MODEL_PARKING_NG: F<ParkingNGOCP>(__VA_ARGS__); break;    \
        default: break;                                                 \
    }

template <class Solver>
static void qp_solve_batch_f32_impl(int B, int n, int m, const float* H, const float* h, const float* A, const float* Alb, const float* Aub,
                                    const float* xlb, const float* xub, const float* x0, const float* y0, const orc_qp_settings* s, int pivot,
                                    float* x, float* y, orc_qp_info* info) {
    for (int b = 0; b < B; ++b) {
        Solver q(n, m);
        q.settings.eps_rel = (float)s->eps_rel; q.settings.eps_abs = (float)s->eps_abs; q.settings.max_iter = s->max_iter;
        q.settings.rho = (float)s->rho; q.settings.sigma = (float)s->sigma; q.settings.alpha = (float)s->alpha;
        q.settings.check_termination = s->check_termination; q.settings.adaptive_rho = s->adaptive_rho != 0;
        q.settings.adaptive_rho_tolerance = (float)s->adaptive_rho_tolerance; q.settings.adaptive_rho_interval = s->adaptive_rho_interval;
        q.pivot = (pivot_policy)pivot;
        q.solve(H + (size_t)b * n * n, h + (size_t)b * n, A + (size_t)b * m * n, Alb + (size_t)b * m, Aub + (size_t)b * m, xlb + (size_t)b * n,
                xub + (size_t)b * n, x0 ? x0 + (size_t)b * n : nullptr, y0 ? y0 + (size_t)b * (n + m) : nullptr);
        for (int i = 0; i < n; ++i) x[(size_t)b * n + i] = q.x[i];
        for (int i = 0; i < n + m; ++i) y[(size_t)b * (n + m) + i] = q.y[i];
        info[b].status = q.info.status; info[b].iter = q.info.iter; info[b].rho_updates = q.info.rho_updates; info[b].flags = 0;
        info[b].rho_estimate = q.info.rho_estimate; info[b].res_prim = q.info.res_prim; info[b].res_dual = q.info.res_dual;
    }
}

extern "C" {

double orc_set_schur_refine_gate(double g) { const double old = oracle::BoxADMM::schur_refine_gate(); oracle::BoxADMM::schur_refine_gate() = g; return old; }
void orc_schur_refine_counts(long long* out, int reset) { auto* c = oracle::BoxADMM::schur_refine_counts(); out[0] = c[0]; out[1] = c[1]; if (reset) { c[0] = 0; c[1] = 0; } }
int orc_set_hx_identity(int on) { const int old = oracle::BoxADMM::hx_identity(); oracle::BoxADMM::hx_identity() = on; return old; }
int orc_set_libm(int use_libm) { const int old = oracle::use_libm() ? 1 : 0; oracle::use_libm() = use_libm != 0; return old; }
void orc_math_eval(int kind, int impl, int count, const double* x, double* y) {
    for (int i = 0; i < count; ++i) {
        const double v = x[i];
        if (impl == 0) y[i] = kind == 0 ? pmpc::detmath::sin(v) : (kind == 1 ? pmpc::detmath::cos(v) : pmpc::detmath::exp(v));
        else y[i] = kind == 0 ? std::sin(v) : (kind == 1 ? std::cos(v) : std::exp(v));
    }
}

void orc_qp_default_settings(orc_qp_settings* s) { from_qp(qp_settings(), s); }
void orc_sqp_qp_default_settings(orc_qp_settings* s) {
    qp_settings q;  // sqp_base.hpp:83-90
    q.warm_start = false; q.check_termination = 10; q.eps_abs = 1e-4; q.eps_rel = 1e-4; q.max_iter = 100;
    q.adaptive_rho = true; q.adaptive_rho_interval = 50; q.alpha = 1.0;
    from_qp(q, s);
}
void orc_sqp_default_settings(orc_sqp_settings* s) {
    sqp_settings q;
    s->tau = q.tau; s->eta = q.eta; s->rho = q.rho; s->eps_prim = q.eps_prim; s->eps_dual = q.eps_dual;
    s->max_iter = q.max_iter; s->line_search_max_iter = q.line_search_max_iter;
    s->regularisation = 0; s->exact_hessian_every_iter = 0; s->preconditioner = 0; s->hessian_update = 0; s->qp_solver = 0;
    s->line_search = q.line_search; s->filter_max_depth = q.filter_max_depth; s->filter_beta = q.filter_beta; s->filter_state = nullptr;
    s->iteration_trace = nullptr; s->iteration_trace_capacity = 0;
}

void orc_cheb(int P, double* nodes, double* weights, double* D) {
    Chebyshev c(P);
    std::memcpy(nodes, c.nodes.data(), sizeof(double) * (P + 1));
    std::memcpy(weights, c.weights.data(), sizeof(double) * (P + 1));
    std::memcpy(D, c.D.data(), sizeof(double) * (P + 1) * (P + 1));
}
int orc_classify(double lb, double ub) { return BoxADMM::classify(lb, ub); }
void orc_bfgs(int n, double* B, const double* s, const double* y) { BFGS_update(B, s, y, n); }
void orc_regularise(int kind, int n, double* H) {
    if (kind == REG_EIG_MIRROR) regularise_eig_mirror(H, n);
    else if (kind == REG_GERSHGORIN) regularise_gershgorin(H, n);
}
void orc_ldlt_solve(int n, const double* K, const double* b, int pivot, double* x) {
    LDLT f; f.compute(std::vector<double>(K, K + n * n), n, (pivot_policy)pivot); f.solve(b, x);
}

void orc_qp_admm_solve_batch(int B, int n, int m, const double* H, const double* h, const double* A, const double* Alb,
                             const double* Aub, const double* xlb, const double* xub, const double* x0, const double* y0,
                             const orc_qp_settings* s, int pivot, int threads, double* x, double* y, orc_qp_info* info) {
#pragma omp parallel for schedule(dynamic, 4) num_threads(threads > 1 ? threads : 1)
    for (int b = 0; b < B; ++b) {
        ADMM qp(n, m);
        qp.settings = to_qp(s);
        qp.pivot = (pivot_policy)pivot;
        qp.solve(H + (size_t)b * n * n, h + (size_t)b * n, A + (size_t)b * m * n, Alb + (size_t)b * m, Aub + (size_t)b * m, xlb + (size_t)b * n,
                 xub + (size_t)b * n, x0 ? x0 + (size_t)b * n : nullptr, y0 ? y0 + (size_t)b * (n + m) : nullptr);
        for (int i = 0; i < n; ++i) x[(size_t)b * n + i] = qp.x[i];
        for (int i = 0; i < n + m; ++i) y[(size_t)b * (n + m) + i] = qp.y[i];
        info[b].status = qp.info.status; info[b].iter = qp.info.iter; info[b].rho_updates = qp.info.rho_updates; info[b].flags = 0;
        info[b].rho_estimate = qp.info.rho_estimate; info[b].res_prim = qp.info.res_prim; info[b].res_dual = qp.info.res_dual;
    }
}

void orc_ruiz_compute_batch(int B, int n, int m, double* H, double* h, double* A, double* Al, double* Au, double* l, double* u,
                            double* D, double* E, double* c) {
    for (int b = 0; b < B; ++b) {
        Ruiz r(n, m);
        r.compute(H + (size_t)b * n * n, h + (size_t)b * n, A + (size_t)b * m * n, Al + (size_t)b * m, Au + (size_t)b * m,
                  l + (size_t)b * n, u + (size_t)b * n);
        for (int k = 0; k < n; ++k) D[(size_t)b * n + k] = r.D[k];
        for (int k = 0; k < m; ++k) E[(size_t)b * m + k] = r.E[k];
        c[b] = r.c;
    }
}
void orc_ruiz_unscale_solution_batch(int B, int n, int m, const double* D, const double* E, const double* c, double* x, double* y) {
    for (int b = 0; b < B; ++b) {
        Ruiz r(n, m);
        for (int k = 0; k < n; ++k) r.D[k] = D[(size_t)b * n + k];
        for (int k = 0; k < m; ++k) r.E[k] = E[(size_t)b * m + k];
        r.c = c[b];
        r.unscale(x + (size_t)b * n, y + (size_t)b * (n + m));
    }
}

static BoxADMM::SchurStruct g_schur;   // collocation structure of the QPs handed to orc_qp_solve_batch with PIVOT_SCHUR (set before the call)
void orc_set_schur_structure(int nx, int nu, int nn, int P) { g_schur.nx = nx; g_schur.nu = nu; g_schur.nn = nn; g_schur.P = P; g_schur.np = 0; }
void orc_set_schur_structure_np(int nx, int nu, int nn, int P, int np) { orc_set_schur_structure(nx, nu, nn, P); g_schur.np = np; }

/* one KKT solve in the order of `pivot` (any policy, PIVOT_CONDENSED / PIVOT_SCHUR included): K = [P A'; A -diag(rho_inv)] given by its lower
   triangle, (n+m)^2 column-major; accuracy probes of the restated orders */
void orc_kkt_solve(int n, int m, const double* K, const double* rho_vec, const double* rhs, int pivot, double* sol) {
    BoxADMM qp(n, m);
    qp.pivot = (pivot_policy)pivot; qp.schur = g_schur;
    qp.K.assign(K, K + (size_t)(n + m) * (n + m));
    for (int i = 0; i < m; ++i) { qp.rho_vec[i] = rho_vec[i]; qp.rho_inv_vec[i] = 1.0 / rho_vec[i]; }
    qp.factorise();
    qp.kkt_solve(rhs, sol);
}

void orc_qp_solve_batch(int B, int n, int m, const double* H, const double* h, const double* A, const double* Alb,
                        const double* Aub, const double* xlb, const double* xub, const double* x0, const double* y0,
                        const orc_qp_settings* s, int pivot, int threads, double* x, double* y, orc_qp_info* info) {
#pragma omp parallel for schedule(dynamic, 4) num_threads(threads > 1 ? threads : 1)
    for (int b = 0; b < B; ++b) {
        BoxADMM qp(n, m);
        qp.settings = to_qp(s);
        qp.pivot = (pivot_policy)pivot;
        qp.schur = g_schur;
        const double* Hb = H + (size_t)b * n * n; const double* hb = h + (size_t)b * n;
        const double* Ab = A + (size_t)b * m * n;
        const double* alb = Alb + (size_t)b * m; const double* aub = Aub + (size_t)b * m;
        const double* xl = xlb + (size_t)b * n; const double* xu = xub + (size_t)b * n;
        if (x0 && y0) qp.solve(Hb, hb, Ab, alb, aub, xl, xu, x0 + (size_t)b * n, y0 + (size_t)b * (n + m));
        else qp.solve(Hb, hb, Ab, alb, aub, xl, xu);
        if (qp.gives_up() && pivot == PIVOT_SWEEP) {   // the product's QP entry point: redo launch on the LDS-resident kernel (static LDL^T), flag kept
            BoxADMM qp2(n, m);
            qp2.settings = to_qp(s); qp2.pivot = PIVOT_STATIC;
            if (x0 && y0) qp2.solve(Hb, hb, Ab, alb, aub, xl, xu, x0 + (size_t)b * n, y0 + (size_t)b * (n + m));
            else qp2.solve(Hb, hb, Ab, alb, aub, xl, xu);
            qp2.info.flags |= QP_FLAG_ILLCOND;
            qp.x = qp2.x; qp.y = qp2.y; qp.info = qp2.info;
        }
        std::memcpy(x + (size_t)b * n, qp.x.data(), sizeof(double) * n);
        std::memcpy(y + (size_t)b * (n + m), qp.y.data(), sizeof(double) * (n + m));
        info[b].status = qp.info.status; info[b].iter = qp.info.iter; info[b].rho_updates = qp.info.rho_updates; info[b].flags = qp.info.flags;
        info[b].rho_estimate = qp.info.rho_estimate; info[b].res_prim = qp.info.res_prim; info[b].res_dual = qp.info.res_dual;
    }
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------------
template <class Model>
static void dims_impl(int P, int S, int* nx, int* nu, int* np, int* nd, int* ng, int* n, int* me, int* mi) {
    ContinuousOCP<Model> ocp(P, S);
    *nx = Model::NX; *nu = Model::NU; *np = Model::NP; *nd = Model::ND; *ng = Model::NG;
    *n = ocp.VAR_SIZE; *me = ocp.NUM_EQ; *mi = ocp.NUM_INEQ;
}
template <class Model> static void tn_impl(int P, int S, double t0, double tf, double* tn) {
    ContinuousOCP<Model> ocp(P, S); ocp.set_time_limits(t0, tf);
    std::memcpy(tn, ocp.time_nodes.data(), sizeof(double) * ocp.NN);
}

template <class Model>
static void eval_impl(int P, int S, double t0, double tf, const double* mp, int nmp, const double* var, const double* d,
                      const double* lam, double* cost, double* c_eq, double* g_ineq, double* jac, double* cost_grad,
                      double* cost_hess, double* lag_grad, double* lag_hess) {
    ContinuousOCP<Model> ocp(P, S, make_model<Model>(mp, nmp));
    ocp.set_time_limits(t0, tf);
    const int n = ocp.VAR_SIZE, m = ocp.NUM_EQ + ocp.NUM_INEQ;
    std::vector<double> dd(Model::ND > 0 ? Model::ND : 1, 0.0);
    for (int i = 0; i < Model::ND; ++i) dd[i] = d[i];
    if (cost) ocp.cost(var, dd.data(), *cost);
    if (c_eq) ocp.equalities(var, dd.data(), c_eq);
    if (g_ineq) ocp.inequalities(var, dd.data(), g_ineq);
    if (cost_hess || cost_grad) {
        std::vector<double> g(n), Hh(n * n); double c;
        ocp.cost_gradient_hessian(var, dd.data(), c, g.data(), Hh.data());
        if (cost_hess) std::memcpy(cost_hess, Hh.data(), sizeof(double) * n * n);
        if (cost_grad) { ocp.cost_gradient(var, dd.data(), c, g.data()); std::memcpy(cost_grad, g.data(), sizeof(double) * n); }
    }
    if (jac || lag_grad || lag_hess) {
        std::vector<double> lg(n), cg(n), gg(m), J(m * n), Hh(n * n), lz(m + n, 0.0); double l;
        const double* L = lam ? lam : lz.data();
        ocp.lagrangian_gradient_hessian(var, dd.data(), L, l, lg.data(), Hh.data(), cg.data(), gg.data(), J.data());
        if (lag_hess) std::memcpy(lag_hess, Hh.data(), sizeof(double) * n * n);
        if (jac) std::memcpy(jac, J.data(), sizeof(double) * m * n);
        if (lag_grad) {
            ocp.lagrangian_gradient(var, dd.data(), L, l, lg.data(), cg.data(), gg.data(), J.data());
            std::memcpy(lag_grad, lg.data(), sizeof(double) * n);
        }
    }
}

template <class Model>
static void setup_solver(SQP<ContinuousOCP<Model>>& sqp, int b, const double* x_guess, const double* lam_guess,
                         const double* d, const double* lbx, const double* ubx, const double* lbg, const double* ubg,
                         const orc_sqp_settings* ss, const orc_qp_settings* qs, int pivot) {
    const int n = sqp.n, m = sqp.m, mi = sqp.mi;
    sqp.settings = to_sqp(ss);
    sqp.qp.settings = to_qp(qs);
    sqp.qp.pivot = (pivot_policy)pivot;
    if (pivot == PIVOT_SCHUR || pivot == PIVOT_CONDSWEEP) { sqp.qp.schur.nx = Model::NX; sqp.qp.schur.nu = Model::NU; sqp.qp.schur.nn = sqp.problem.NN; sqp.qp.schur.P = sqp.problem.P; sqp.qp.schur.np = Model::NP; }
    for (int i = 0; i < Model::ND; ++i) sqp.p_static[i] = d[(size_t)b * Model::ND + i];
    if (lbx) for (int i = 0; i < n; ++i) sqp.lbx[i] = lbx[(size_t)b * n + i];
    if (ubx) for (int i = 0; i < n; ++i) sqp.ubx[i] = ubx[(size_t)b * n + i];
    if (lbg) for (int i = 0; i < mi; ++i) sqp.lbg[i] = lbg[(size_t)b * mi + i];
    if (ubg) for (int i = 0; i < mi; ++i) sqp.ubg[i] = ubg[(size_t)b * mi + i];
    if (x_guess) for (int i = 0; i < n; ++i) sqp.x[i] = x_guess[(size_t)b * n + i];
    if (lam_guess) for (int i = 0; i < m + n; ++i) sqp.lam[i] = lam_guess[(size_t)b * (m + n) + i];
}

template <class Model>
static void sqp_batch_impl(int P, int S, double t0, double tf, const double* mp, int nmp, int B, const double* x_guess,
                           const double* lam_guess, const double* d, const double* lbx, const double* ubx,
                           const double* lbg, const double* ubg, const orc_sqp_settings* ss, const orc_qp_settings* qs,
                           int pivot, int threads, double* x, double* lam, orc_sqp_info* info) {
#pragma omp parallel for schedule(dynamic, 4) num_threads(threads > 1 ? threads : 1)
    for (int b = 0; b < B; ++b) {
        ContinuousOCP<Model> ocp(P, S, make_model<Model>(mp, nmp));
        ocp.set_time_limits(t0, tf);
        SQP<ContinuousOCP<Model>> sqp(ocp, Model::ND);
        setup_solver<Model>(sqp, b, x_guess, lam_guess, d, lbx, ubx, lbg, ubg, ss, qs, pivot);
        if (ss->filter_state) sqp.filter.load(ss->filter_state + (size_t)b * ORC_FILTER_STATE_DOUBLES);
        if (ss->iteration_trace) { sqp.trace = ss->iteration_trace + (size_t)b * ss->iteration_trace_capacity * ORC_TRACE_DOUBLES; sqp.trace_capacity = ss->iteration_trace_capacity; }
        // the conditioning rule of the register-resident kernels (sqp_kernel, pmpc_launch.hpp): an instance with an unbounded control or parameter
        // (LOOSE_BOUNDS: rho_box = RHO_MIN in a direction the collocation Jacobian leaves free) goes to the redo launch before any work
        bool structural_redo = false;
        if (pivot == PIVOT_SWEEP || pivot == PIVOT_CONDSWEEP) {
            sqp.qp.numeric_gate = false;
            const int varx = Model::NX * (P * S + 1);
            for (int i = varx; i < sqp.n; ++i) structural_redo |= BoxADMM::classify(sqp.lbx[i], sqp.ubx[i]) == BoxADMM::LOOSE_BOUNDS;
        }
        if (structural_redo) sqp.info.status = SQP_REDO; else sqp.solve();
        auto store = [&](SQP<ContinuousOCP<Model>>& s, int extra_flags) {
            if (ss->filter_state) s.filter.store(ss->filter_state + (size_t)b * ORC_FILTER_STATE_DOUBLES);
            const int n = s.n, m = s.m;
            std::memcpy(x + (size_t)b * n, s.x.data(), sizeof(double) * n);
            std::memcpy(lam + (size_t)b * (m + n), s.lam.data(), sizeof(double) * (m + n));
            info[b].iter = s.info.iter; info[b].qp_solver_iter = s.info.qp_solver_iter; info[b].status = s.info.status; info[b].flags = s.info.flags | extra_flags;
            info[b].primal_norm = s.primal_norm; info[b].dual_norm = s.dual_norm;
            info[b].max_violation = s.max_violation; info[b].cost = s.cost_;
        };
        if (sqp.info.status == SQP_REDO) {
            // a QP of the condensed orders gave up at its conditioning gate (BoxADMM::COND_GATE): the instance is solved again from its guesses in the full
            // KKT form — what the product's redo launch does (pmpc_launch.hpp): the LDS-resident static LDL^T behind the one-row-per-lane kernel, the
            // two-rows-per-lane full inverse behind the condensed register kernel, the (n + m)-row blocked LDL^T behind the large-instance kernel
            const int redo = (int)BoxADMM::redo_policy((pivot_policy)pivot);
            ContinuousOCP<Model> ocp2(P, S, make_model<Model>(mp, nmp));
            ocp2.set_time_limits(t0, tf);
            SQP<ContinuousOCP<Model>> sqp2(ocp2, Model::ND);
            setup_solver<Model>(sqp2, b, x_guess, lam_guess, d, lbx, ubx, lbg, ubg, ss, qs, redo);
            if (ss->filter_state) sqp2.filter.load(ss->filter_state + (size_t)b * ORC_FILTER_STATE_DOUBLES);
            if (ss->iteration_trace) { sqp2.trace = ss->iteration_trace + (size_t)b * ss->iteration_trace_capacity * ORC_TRACE_DOUBLES; sqp2.trace_capacity = ss->iteration_trace_capacity; }
            sqp2.solve();
            store(sqp2, QP_FLAG_ILLCOND);
        } else store(sqp, 0);
    }
}

template <class Model>
static void sqp_trace_impl(int* nrec, int P, int S, double t0, double tf, const double* mp, int nmp, const double* x_guess,
                           const double* lam_guess, const double* d, const double* lbx, const double* ubx,
                           const double* lbg, const double* ubg, const orc_sqp_settings* ss, const orc_qp_settings* qs,
                           int pivot, int max_qps, double* H, double* h, double* A, double* al, double* au, double* lx,
                           double* ux) {
    ContinuousOCP<Model> ocp(P, S, make_model<Model>(mp, nmp));
    ocp.set_time_limits(t0, tf);
    SQP<ContinuousOCP<Model>> sqp(ocp, Model::ND);
    setup_solver<Model>(sqp, 0, x_guess, lam_guess, d, lbx, ubx, lbg, ubg, ss, qs, pivot);
    sqp.record_qps = true;
    sqp.solve();
    const int n = sqp.n, m = sqp.m;
    int k = 0;
    for (auto& r : sqp.qp_trace) {
        if (k >= max_qps) break;
        std::memcpy(H + (size_t)k * n * n, r.H.data(), sizeof(double) * n * n);
        std::memcpy(h + (size_t)k * n, r.h.data(), sizeof(double) * n);
        std::memcpy(A + (size_t)k * m * n, r.A.data(), sizeof(double) * m * n);
        std::memcpy(al + (size_t)k * m, r.al.data(), sizeof(double) * m);
        std::memcpy(au + (size_t)k * m, r.au.data(), sizeof(double) * m);
        std::memcpy(lx + (size_t)k * n, r.lx.data(), sizeof(double) * n);
        std::memcpy(ux + (size_t)k * n, r.ux.data(), sizeof(double) * n);
        ++k;
    }
    *nrec = k;
}

template <class Def>
static void nlp_impl(const double* x0, const double* lam0, const double* lbx, const double* ubx, const double* lbg,
                     const double* ubg, const orc_sqp_settings* ss, const orc_qp_settings* qs, int pivot, double* x,
                     double* lam, orc_sqp_info* info) {
    GenericNLP<Def> prob;
    SQP<GenericNLP<Def>> sqp(prob, 0);
    sqp.settings = to_sqp(ss); sqp.qp.settings = to_qp(qs); sqp.qp.pivot = (pivot_policy)pivot;
    const int n = sqp.n, m = sqp.m, mi = sqp.mi;
    if (lbx) for (int i = 0; i < n; ++i) sqp.lbx[i] = lbx[i];
    if (ubx) for (int i = 0; i < n; ++i) sqp.ubx[i] = ubx[i];
    if (lbg) for (int i = 0; i < mi; ++i) sqp.lbg[i] = lbg[i];
    if (ubg) for (int i = 0; i < mi; ++i) sqp.ubg[i] = ubg[i];
    std::vector<double> lz(m + n, 0.0);
    sqp.solve(x0, lam0 ? lam0 : lz.data());
    std::memcpy(x, sqp.x.data(), sizeof(double) * n);
    std::memcpy(lam, sqp.lam.data(), sizeof(double) * (m + n));
    info->iter = sqp.info.iter; info->qp_solver_iter = sqp.info.qp_solver_iter; info->status = sqp.info.status; info->flags = sqp.info.flags;
    info->primal_norm = sqp.primal_norm; info->dual_norm = sqp.dual_norm; info->max_violation = sqp.max_violation;
    info->cost = sqp.cost_;
}

extern "C" {

void orc_ocp_dims(int model, int P, int S, int* nx, int* nu, int* np, int* nd, int* ng, int* n, int* m_eq, int* m_ineq) {
    DISPATCH_MODEL(model, dims_impl, P, S, nx, nu, np, nd, ng, n, m_eq, m_ineq);
}
void orc_ocp_time_nodes(int model, int P, int S, double t0, double tf, double* tn) {
    DISPATCH_MODEL(model, tn_impl, P, S, t0, tf, tn);
}
void orc_ocp_eval(int model, int P, int S, double t0, double tf, const double* mparams, int n_mparams, const double* var,
                  const double* d, const double* lam, double* cost, double* c_eq, double* g_ineq, double* jac,
                  double* cost_grad, double* cost_hess, double* lag_grad, double* lag_hess) {
    DISPATCH_MODEL(model, eval_impl, P, S, t0, tf, mparams, n_mparams, var, d, lam, cost, c_eq, g_ineq, jac, cost_grad,
                   cost_hess, lag_grad, lag_hess);
}
// boxADMM<N, M, float> (box_admm_test.cpp:85-115) / ADMM<N, M, float> (admm_solver_test.cpp:84-113): float arrays, the settings narrowed to float as
// qp_solver_settings_t<float> holds them
void orc_qp_solve_batch_f32(int B, int n, int m, const float* H, const float* h, const float* A, const float* Alb, const float* Aub,
                            const float* xlb, const float* xub, const float* x0, const float* y0, const orc_qp_settings* s, int pivot,
                            float* x, float* y, orc_qp_info* info) {
    qp_solve_batch_f32_impl<BoxADMMf>(B, n, m, H, h, A, Alb, Aub, xlb, xub, x0, y0, s, pivot, x, y, info);
}
void orc_qp_admm_solve_batch_f32(int B, int n, int m, const float* H, const float* h, const float* A, const float* Alb, const float* Aub,
                                 const float* xlb, const float* xub, const float* x0, const float* y0, const orc_qp_settings* s, int pivot,
                                 float* x, float* y, orc_qp_info* info) {
    qp_solve_batch_f32_impl<ADMMf>(B, n, m, H, h, A, Alb, Aub, xlb, xub, x0, y0, s, pivot, x, y, info);
}

void orc_sqp_solve_batch(int model, int P, int S, double t0, double tf, const double* mparams, int n_mparams, int B,
                         const double* x_guess, const double* lam_guess, const double* d, const double* lbx,
                         const double* ubx, const double* lbg, const double* ubg, const orc_sqp_settings* ss,
                         const orc_qp_settings* qs, int pivot, int threads, double* x, double* lam, orc_sqp_info* info) {
    DISPATCH_MODEL(model, sqp_batch_impl, P, S, t0, tf, mparams, n_mparams, B, x_guess, lam_guess, d, lbx, ubx, lbg, ubg,
                   ss, qs, pivot, threads, x, lam, info);
}
int orc_sqp_trace_qps(int model, int P, int S, double t0, double tf, const double* mparams, int n_mparams,
                      const double* x_guess, const double* lam_guess, const double* d, const double* lbx,
                      const double* ubx, const double* lbg, const double* ubg, const orc_sqp_settings* ss,
                      const orc_qp_settings* qs, int pivot, int max_qps, double* H, double* h, double* A, double* al,
                      double* au, double* lx, double* ux) {
    int nrec = 0;
    DISPATCH_MODEL(model, sqp_trace_impl, &nrec, P, S, t0, tf, mparams, n_mparams, x_guess, lam_guess, d, lbx, ubx, lbg,
                   ubg, ss, qs, pivot, max_qps, H, h, A, al, au, lx, ux);
    return nrec;
}
void orc_nlp_solve(int problem, const double* x0, const double* lam0, const double* lbx, const double* ubx,
                   const double* lbg, const double* ubg, const orc_sqp_settings* ss, const orc_qp_settings* qs,
                   int pivot, double* x, double* lam, orc_sqp_info* info) {
    switch (problem) {
        case ORC_NLP_CONSTRAINED_ROSENBROCK: nlp_impl<ConstrainedRosenbrockDef>(x0, lam0, lbx, ubx, lbg, ubg, ss, qs, pivot, x, lam, info); break;
        case ORC_NLP_ROSENBROCK: nlp_impl<RosenbrockDef>(x0, lam0, lbx, ubx, lbg, ubg, ss, qs, pivot, x, lam, info); break;
        case ORC_NLP_SIMPLE: nlp_impl<SimpleNLPDef>(x0, lam0, lbx, ubx, lbg, ubg, ss, qs, pivot, x, lam, info); break;
        case ORC_NLP_HS071: nlp_impl<HS071Def>(x0, lam0, lbx, ubx, lbg, ubg, ss, qs, pivot, x, lam, info); break;
        default: break;
    }
}

}  // extern "C"
