/* ORACLE — TEST INFRASTRUCTURE ONLY (CPU restatement of the reference algorithm).
 * C entry points of the oracle shared library, bound with ctypes from tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg. Never linked by polympc_amd. */
#ifndef ORACLE_CAPI_H
#define ORACLE_CAPI_H
#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    double eps_rel, eps_abs;
    int max_iter;
    double rho, sigma, alpha;
    int check_termination;
    int adaptive_rho;
    double adaptive_rho_tolerance;
    int adaptive_rho_interval;
} orc_qp_settings;

typedef struct {
    int status, iter, rho_updates, flags;   /* flags: 2 = the conditioning gate of the condensed / constraint-first orders tripped */
    double rho_estimate, res_prim, res_dual;
} orc_qp_info;

typedef struct {
    double tau, eta, rho, eps_prim, eps_dual;
    int max_iter, line_search_max_iter;
    int regularisation;            /* 0 none, 1 eigenvalue mirroring, 2 Gershgorin */
    int exact_hessian_every_iter;  /* 0 = damped BFGS (default) */
    int preconditioner;            /* 0 identity (default), 1 Ruiz equilibration */
    int hessian_update;            /* 0 dense damped BFGS (default), 1 block BFGS of ContinuousOCP */
    int qp_solver;                 /* 0 boxADMM (default), 1 ADMM (OSQP form) */
    int line_search;               /* 0 l1-merit backtracking (default), 1 filter line search (LSFilter) */
    int filter_max_depth;          /* LSFilter::max_depth (10) */
    double filter_beta;            /* LSFilter::beta (1e-5) */
    double* filter_state;          /* NULL: every solve starts with an empty filter; else B x ORC_FILTER_STATE_DOUBLES, read before and
                                    * written after the solve (the solver member that outlives solve()): [count, cost0, viol0, cost1, ...] */
    double* iteration_trace;       /* NULL, or B x iteration_trace_capacity x ORC_TRACE_DOUBLES: [iter, alpha, primal_norm, dual_norm, cost, qp iterations,
                                    * qp status, max violation] per SQP iteration — what iteration_callback (sqp_base.hpp:33,685-686) could read */
    int iteration_trace_capacity;
} orc_sqp_settings;
enum { ORC_FILTER_STATE_DOUBLES = 21, ORC_TRACE_DOUBLES = 8 };

typedef struct {
    int iter, qp_solver_iter, status, flags;   /* flags: 2 = a QP gave up at the conditioning gate and the instance was re-solved in the full KKT form */
    double primal_norm, dual_norm, max_violation, cost;
} orc_sqp_info;

enum { ORC_MODEL_ROBOT = 0, ORC_MODEL_CSTR = 1, ORC_MODEL_PARKING = 2, ORC_MODEL_ROBOT_NG = 3, ORC_MODEL_KITE_STANDIN = 4, ORC_MODEL_PARKING_NG = 5 };
enum { ORC_NLP_CONSTRAINED_ROSENBROCK = 0, ORC_NLP_ROSENBROCK = 1, ORC_NLP_SIMPLE = 2, ORC_NLP_HS071 = 3 };

/* sin / cos / exp used by the model evaluations: 0 (default) = pmpc::detmath, the IEEE-only restatement the HIP kernels share
 * (GPU-vs-oracle comparisons are bit for bit); 1 = glibc, what the reference binary calls. Process-wide; returns the old value. */
int  orc_set_libm(int use_libm);
double orc_set_schur_refine_gate(double gate);   /* experiment switch (round 6): PIVOT_SCHUR's refinement step only above this conditioning estimate (0: always); returns the previous value */
void orc_schur_refine_counts(long long* out2, int reset);   /* solves with / without the refinement step since the last reset */
int  orc_set_hx_identity(int on);   /* experiment switch (round 6): H x of boxADMM's dual residual from the KKT identity instead of a mat-vec; returns the previous value */
/* kind 0 sin, 1 cos, 2 exp; impl 0 detmath, 1 glibc: y[i] = f(x[i]) */
void orc_math_eval(int kind, int impl, int count, const double* x, double* y);

void orc_qp_default_settings(orc_qp_settings* s);       /* qp_base.hpp:17-53 defaults */
void orc_sqp_qp_default_settings(orc_qp_settings* s);   /* + SQP-ctor overrides sqp_base.hpp:83-90 */
void orc_sqp_default_settings(orc_sqp_settings* s);

void orc_cheb(int P, double* nodes, double* weights, double* D);
int  orc_classify(double lb, double ub);
void orc_bfgs(int n, double* B, const double* s, const double* y);
void orc_regularise(int kind, int n, double* H);
void orc_ldlt_solve(int n, const double* K, const double* b, int pivot, double* x);

/* one KKT solve in the order of `pivot` (every policy): K (n+m)^2 column-major, lower triangle; rho_vec: the m step sizes of the constraint rows */
void orc_kkt_solve(int n, int m, const double* K, const double* rho_vec, const double* rhs, int pivot, double* sol);

/* batch of B QPs, instance-major, column-major matrices; x0/y0 may be NULL (zero guesses). threads<=1: serial */
void orc_qp_solve_batch(int B, int n, int m, const double* H, const double* h, const double* A, const double* Alb,
                        const double* Aub, const double* xlb, const double* xub, const double* x0, const double* y0,
                        const orc_qp_settings* s, int pivot, int threads, double* x, double* y, orc_qp_info* info);

/* PIVOT_SCHUR (7) at the QP entry: the collocation structure of the QPs of the next orc_qp_solve_batch calls — variables [x_0..x_{nn-1} | u_0..u_{nn-1}],
 * equality rows (node, state), P intervals per segment (the SQP entry points take it from the problem) */
void orc_set_schur_structure(int nx, int nu, int nn, int P);
/* the same with np = 1 parameter behind the node variables (n = (nx + nu) nn + np): the bordered form of the block-structured kernel */
void orc_set_schur_structure_np(int nx, int nu, int nn, int P, int np);

/* boxADMM<N, M, float> (box_admm_test.cpp:85-115): float arrays; pivot = PIVOT_EIGEN or PIVOT_STATIC; the info's floats are widened */
void orc_qp_solve_batch_f32(int B, int n, int m, const float* H, const float* h, const float* A, const float* Alb, const float* Aub,
                            const float* xlb, const float* xub, const float* x0, const float* y0, const orc_qp_settings* s, int pivot,
                            float* x, float* y, orc_qp_info* info);

/* ADMM<N, M, float> (admm_solver_test.cpp:84-113), same arguments */
void orc_qp_admm_solve_batch_f32(int B, int n, int m, const float* H, const float* h, const float* A, const float* Alb, const float* Aub,
                                 const float* xlb, const float* xub, const float* x0, const float* y0, const orc_qp_settings* s, int pivot,
                                 float* x, float* y, orc_qp_info* info);

/* the OSQP-style ADMM solver (admm.hpp) on the same batch layout; y has m+n entries per instance ([general | box]) */
void orc_qp_admm_solve_batch(int B, int n, int m, const double* H, const double* h, const double* A, const double* Alb,
                             const double* Aub, const double* xlb, const double* xub, const double* x0, const double* y0,
                             const orc_qp_settings* s, int pivot, int threads, double* x, double* y, orc_qp_info* info);

/* Ruiz equilibration of B QPs in place (qp_preconditioners.hpp:160-233); outputs D (B x n), E (B x m), c (B) */
void orc_ruiz_compute_batch(int B, int n, int m, double* H, double* h, double* A, double* Al, double* Au, double* l, double* u,
                            double* D, double* E, double* c);
/* x <- x.*D, y <- (1/c) [y_A.*E ; y_box./D]  (qp_preconditioners.hpp:359-364) */
void orc_ruiz_unscale_solution_batch(int B, int n, int m, const double* D, const double* E, const double* c, double* x, double* y);

/* sizes of the transcription */
void orc_ocp_dims(int model, int P, int S, int* nx, int* nu, int* np, int* nd, int* ng, int* n, int* m_eq, int* m_ineq);
void orc_ocp_time_nodes(int model, int P, int S, double t0, double tf, double* tn);

/* evaluate every collocation quantity at one point. Any output pointer may be NULL. */
void orc_ocp_eval(int model, int P, int S, double t0, double tf, const double* mparams, int n_mparams, const double* var,
                  const double* d, const double* lam, double* cost, double* c_eq, double* g_ineq, double* jac,
                  double* cost_grad, double* cost_hess, double* lag_grad, double* lag_hess);

/* batch SQP: B instances; lbx/ubx/lbg/ubg/x_guess/lam_guess/d are per-instance (instance-major). */
void orc_sqp_solve_batch(int model, int P, int S, double t0, double tf, const double* mparams, int n_mparams, int B,
                         const double* x_guess, const double* lam_guess, const double* d, const double* lbx,
                         const double* ubx, const double* lbg, const double* ubg, const orc_sqp_settings* ss,
                         const orc_qp_settings* qs, int pivot, int threads, double* x, double* lam, orc_sqp_info* info);

/* run ONE instance and dump every QP it hands to the QP solver (at most max_qps). Returns the number recorded. */
int orc_sqp_trace_qps(int model, int P, int S, double t0, double tf, const double* mparams, int n_mparams,
                      const double* x_guess, const double* lam_guess, const double* d, const double* lbx,
                      const double* ubx, const double* lbg, const double* ubg, const orc_sqp_settings* ss,
                      const orc_qp_settings* qs, int pivot, int max_qps, double* H, double* h, double* A, double* al,
                      double* au, double* lx, double* ux);

/* generic NLP known-answer problems (sqp_test_autodiff.cpp) */
void orc_nlp_solve(int problem, const double* x0, const double* lam0, const double* lbx, const double* ubx,
                   const double* lbg, const double* ubg, const orc_sqp_settings* ss, const orc_qp_settings* qs,
                   int pivot, double* x, double* lam, orc_sqp_info* info);

#ifdef __cplusplus
}
#endif
#endif
