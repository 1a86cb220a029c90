/* polympc_amd — C ABI of the MI355X-native batched SQP / box-ADMM engine.
 *
 * This is the ONLY boundary between host code and HIP. Plain C types, plain pointers and sizes, integer
 * return codes, no exceptions. The reference (PREDICT-EPFL/polympc) has no FFI: its boundary is compile-time
 * CRTP, so each entry point below names the reference member functions it replaces (file:line under
 * /root/reference). A maintainer binds these from the reference's C++ with the stub shown in INTEGRATION.md.
 *
 * Data layout (all fp64, instance-major, every matrix column-major exactly as an Eigen default matrix):
 *   H[b]  n*n     Hessian of QP b                 (qp_hessian_t,    qp_base.hpp:114-115)
 *   A[b]  m*n     general-constraint matrix       (qp_constraint_t, qp_base.hpp:111-112)
 *   h[b] n | Alb[b],Aub[b] m | xlb[b],xub[b] n     (qp_var_t / qp_dual_a_t)
 *   x[b]  n primal, y[b] m+n dual = [general | box]  (QPBase::m_x / m_y, qp_base.hpp:129-130, :251)
 * OCP variable layout: var = [x_0..x_{nn-1} | u_0..u_{nn-1} | p], node 0 = t_stop (continuous_ocp.hpp:757-765);
 * dual lam = [lam_eq | lam_ineq | lam_box] (continuous_ocp.hpp:1919).
 * +-inf bounds are passed as IEEE infinities, exactly as the reference does (sqp_base.hpp:75-78).
 */
#ifndef POLYMPC_AMD_H
#define POLYMPC_AMD_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pmpc_context pmpc_context; /* one per host thread / device; owns a HIP stream + workspaces */

typedef enum {
    PMPC_OK = 0,
    PMPC_ERR_INVALID_ARGUMENT = 1,
    PMPC_ERR_NO_DEVICE = 2,
    PMPC_ERR_HIP = 3,
    PMPC_ERR_UNSUPPORTED_SIZE = 4,
    PMPC_ERR_UNKNOWN_MODEL = 5,
    PMPC_ERR_ABI_MISMATCH = 6      /* the loaded library was built from another version of this header (C++ mirror / Python binding check) */
} pmpc_status;

/* status_t of qp_base.hpp:55-62 (same numeric values) */
typedef enum {
    PMPC_QP_SOLVED = 0,
    PMPC_QP_MAX_ITER_EXCEEDED = 1,
    PMPC_QP_UNSOLVED = 2,
    PMPC_QP_UNINITIALIZED = 3,
    PMPC_QP_INFEASIBLE = 4,
    PMPC_QP_INCONSISTENT = 5
} pmpc_qp_status;

/* qp_solver_settings_t, ADMM-related members (qp_base.hpp:17-53). pmpc_qp_settings_default() fills the
 * reference defaults; pmpc_qp_settings_sqp_default() additionally applies the SQPBase constructor overrides
 * (sqp_base.hpp:83-90). */
typedef struct {
    double eps_rel, eps_abs;
    int max_iter;
    double rho, sigma, alpha;
    int check_termination;
    int adaptive_rho;
    double adaptive_rho_tolerance;
    int adaptive_rho_interval;
    int linear_solver;   /* boxADMM's LinearSolver template argument (box_admm.hpp:25-27, default Eigen::LDLT, helpers.hpp:38-43):
                          * 0 (default) the static-order kernels — K is symmetric quasi-definite whenever H is positive semi-definite, so every pivot
                          *   is non-zero without permutations; register- / LDS- / HBM-resident by size;
                          * 1 LDL^T with Eigen::LDLT's pivoting (largest remaining |diagonal| first, D^+ solve with its zero-pivot rule), operation for
                          *   operation the CPU restatement's Eigen-style policy; LDS-resident kernel only (PMPC_ERR_UNSUPPORTED_SIZE beyond ~190 KKT
                          *   rows), an order of magnitude slower: for indefinite Hessians used without regularisation, and for cross-checks.
                          * (occupies what was tail padding: the struct size is unchanged) */
} pmpc_qp_settings;

/* qp_solver_info_t (qp_base.hpp:64-72), one per instance. rho_updates counts rho_vec_update calls of THIS solve
 * (= number of KKT factorisations), the reference never resets its counter (box_admm.hpp:395). */
typedef struct {
    int status, iter, rho_updates;
    int flags;   /* PMPC_FLAG_NONFINITE: a non-finite value reached the QP solution — the static-order (unpivoted) factorisation met a zero
                  * pivot (indefinite Hessian without regularisation). The reference has no such report: Eigen::LDLT pivots and "simply
                  * proceeds" (SURVEY Appendix B); status follows the reference's rules, this word says the numbers are not to be trusted. */
    double rho_estimate, res_prim, res_dual;
} pmpc_qp_info;
#define PMPC_FLAG_NONFINITE 1
/* PMPC_FLAG_ILLCOND: information, not an error — "this QP / instance was solved in the full KKT form". The default kernels eliminate boxADMM's diagonal
 * constraint block first: they invert / factorise S = H + sigma I + rho_box + A' diag(rho) A (or, block-structured kernel, 1/rho + A Q A') instead of the
 * (n + m)-row KKT matrix of box_admm.hpp:209-223. While the directions A leaves free are BOUNDED variables cond(S) stays ~1e5 whatever rho is and these
 * kernels follow the exact-arithmetic ADMM more closely than the reference's pivoted LDL^T does; when unbounded variables span them it grows with rho.
 * Three mechanisms hand such work to a full-form solve, each restated rule for rule by the CPU checker; all RESTART from the guesses (the QP from x0 / y0,
 * the SQP instance from x_guess / lam_guess — nothing of the abandoned attempt is used):
 *   (1) bounds rule — fused SQP kernels with a register-resident QP (one KKT row per lane; condensed register kernel): decided ONCE per instance, BEFORE
 *       any work, from its bounds: a control or parameter that is unbounded on both sides (|bound| > 1e10, qp_base.hpp:195-222) sends the instance to the
 *       redo launch (LDS-resident static LDL^T resp. the two-rows-per-lane full inverse). These kernels do NOT estimate cond(S) (a numeric gate cost them
 *       4 .. 10 %); a batch whose controls are all free is therefore solved entirely by the redo kernel, at that kernel's (lower) speed.
 *   (2) numeric gate, 1e10 — the QP entry point's one-row-per-lane kernels (max S_ii * max |(S^-1)_ii| off the swept tiles) and the large-instance kernel
 *       in its condensed mode (max S_ii / min |d_k| of its LDL^T), at every factorisation: beyond 1e10 the QP (entry point: re-solved by the LDS-resident
 *       kernel) resp. the instance (redo launch of the same kernel with kkt_form = 1) is given up.
 *   (3) numeric gate, 1e7 (PMPC_SCHUR_COND_GATE) — the block-structured kernel, whose range-space solve loses accuracy like that estimate SQUARED: the
 *       instance is re-solved by the LDS-resident static LDL^T.
 * pmpc_sqp_last_route reports the kernel family of the PRIMARY launch, also for instances that were re-solved. A developer's PMPC_NO_REDO_LAUNCH=1 (timing
 * only) skips the second launch: instances that gave up then keep status 4 (internal: "redo") with x / lam equal to their guesses.
 * pmpc_sqp_settings::kkt_form = 1 asks for the full form from the start. */
#define PMPC_FLAG_ILLCOND 2

/* sqp_settings_t (sqp_base.hpp:24-47) + the two override points the reference's tests use:
 * regularisation: 0 none (default hook, sqp_base.hpp:305), 1 eigenvalue mirroring (sqp_test_autodiff.cpp:29-45; Jacobi iteration in LDS, needs
 *                 16 n^2 bytes of LDS per instance: PMPC_ERR_UNSUPPORTED_SIZE beyond that; a round-robin Jacobi iteration on one wavefront in the restatement's
 *                 order — the reference uses it on a 2-variable NLP; on an OCP it costs ~0.6 ms (n = 35) .. ~1.7 ms (n = 55) .. ~4.5 ms (n = 80) per SQP iteration of a
 *                 lone instance), 2 Gershgorin shift (dense_sparse_compare.cpp:109-122)
 * exact_hessian_every_iter: update_linearisation_dense_impl overridden to linearisation_dense_impl
 *                           (codegen_test.cpp:381-398) instead of damped BFGS (bfgs.hpp:23-52). */
typedef struct {
    double tau, eta, rho, eps_prim, eps_dual;
    int max_iter, line_search_max_iter;
    int regularisation;
    int exact_hessian_every_iter;
    int preconditioner;   /* SQPBase's Preconditioner argument (sqp_base.hpp:64-68): 0 IdentityPreconditioner (default),
                           * 1 RuizEquilibration (qp_preconditioners.hpp:114-385), applied around every QP (sqp_base.hpp:605-611) */
    int hessian_update;   /* SQPBase::hessian_update_impl: 0 damped BFGS on the whole matrix (bfgs.hpp:23-52, the DENSE default),
                           * 1 the sparsity-preserving block BFGS of ContinuousOCP (continuous_ocp.hpp:2304-2431) that the reference's
                           * MPC tests plug in (mpc_wrapper_test.cpp:100-105); register-resident specialisations for 7- and 5-node grids (one KKT row per lane) and 9…13-node grids (two rows per lane) like the default, LDS-resident / HBM-factor kernels otherwise */
    int qp_solver;        /* SQPBase's QPSolver argument: 0 boxADMM (box_admm.hpp, default), 1 ADMM (admm.hpp, OSQP form: (2n+m)-row KKT);
                           * 1 is served by the LDS-resident kernels */
    int line_search;      /* SQPBase::step_size_selection_impl: 0 l1-merit backtracking (sqp_base.hpp:380-419, default), 1 the filter line
                           * search on LSFilter (src/solvers/line_search.hpp:31-98) that valet_parking_mpc_test.cpp:116-158 plugs in;
                           * 1 is served by the LDS-resident kernels */
    int filter_max_depth; /* LSFilter::max_depth (line_search.hpp:38; default 10 = PMPC_FILTER_MAX_DEPTH, the largest accepted) */
    double filter_beta;   /* LSFilter::beta (line_search.hpp:39; default 1e-5) */
    double* filter_state; /* the solver member `filter`, which outlives solve(): NULL (default) = every call starts from an empty filter;
                           * else a DEVICE buffer of B x PMPC_FILTER_STATE_DOUBLES, read when the solve starts and written back when it
                           * ends — per instance [count, cost_0, violation_0, cost_1, violation_1, ...], newest pair first
                           * (pmpc_filter_state_create / _clear / _destroy manage one for hosts without device pointers) */
    double* iteration_trace;      /* counterpart of sqp_settings_t::iteration_callback (sqp_base.hpp:33, called at :685-686 once per iteration
                                   * from the second one on, after the step and its norms): a fused kernel cannot call back into the host, it RECORDS
                                   * what the callback could read. NULL (default) = nothing recorded; else a DEVICE buffer of
                                   * B x iteration_trace_capacity x PMPC_TRACE_DOUBLES: record (iter - 1) of instance b =
                                   * [iter, alpha, primal_norm, dual_norm, cost, qp iterations, qp status, max constraint violation] of SQP iteration
                                   * `iter` (1-based; the first iteration is recorded too — the reference's callback starts at 2), written after that
                                   * iteration's termination test; iterations beyond the capacity are not recorded, records of iterations that
                                   * did not run keep what the buffer held (pmpc_iteration_trace_create / _clear zero it). */
    int iteration_trace_capacity; /* records per instance (ignored when iteration_trace is NULL) */
    int kkt_form;         /* large instances (KKT factor in HBM): 0 (default) condensed — the diagonal constraint block of the KKT matrix is eliminated in
                           * closed form and the n x n matrix H + sigma I + rho_box + A' diag(rho) A is factorised (same solution in exact arithmetic,
                           * box_admm.hpp:209-223 / :123 restated as PIVOT_CONDENSED); 1 the (n+m) x (n+m) KKT matrix as the reference builds it
                           * (grids of 65 .. 128 KKT rows: the two-rows-per-lane full inverse instead of the condensed register kernel, no conditioning
                           * rule; grids of at most 64 rows keep their constraint-first sweep and its bounds rule, whose redo launch IS the full form);
                           * 2 (round 6) the block-structured range-space form wherever a specialisation is compiled, INCLUDING the grids on which
                           * it is not the default because their instances are expected to meet its conditioning gate (parking, NP = 1, 11 nodes:
                           * the bordered form — such instances are re-solved by the redo launch, see PMPC_FLAG_ILLCOND); elsewhere the same as 0.
                           * (Occupies former tail padding: the struct size is unchanged.) */
} pmpc_sqp_settings;
#define PMPC_FILTER_MAX_DEPTH 10
#define PMPC_FILTER_STATE_DOUBLES (1 + 2 * PMPC_FILTER_MAX_DEPTH)
#define PMPC_TRACE_DOUBLES 8

/* sqp_status_t (sqp_base.hpp:49-55) */
typedef enum { PMPC_SQP_SOLVED = 0, PMPC_SQP_MAX_ITER_EXCEEDED = 1, PMPC_SQP_INVALID_SETTINGS = 2 } pmpc_sqp_status;

/* sqp_info_t (sqp_base.hpp:57-61) + the getters primal_norm/dual_norm/constr_violation/cost (:192-195) */
typedef struct {
    int iter, qp_solver_iter, status;
    int flags;   /* PMPC_FLAG_NONFINITE: OR of the pmpc_qp_info::flags of every QP of the solve, and set whenever the returned x or lam holds a
                  * non-finite value, whatever produced it (e.g. a warm start from an unconverged iterate that diverges);
                  * PMPC_FLAG_ILLCOND: a QP of the solve tripped the conditioning gate (see above) */
    double primal_norm, dual_norm, max_violation, cost;
} pmpc_sqp_info;

/* built-in OCP definitions (user OCPs are added with PMPC_REGISTER_OCP, see include/polympc/register_ocp.hpp) */
typedef enum {
    PMPC_MODEL_ROBOT = 0,        /* tests/control/mpc_wrapper_test.cpp:33-80   NX=3 NU=2 NP=0 ND=1 NG=0 */
    PMPC_MODEL_CSTR = 1,         /* tests/control/cstr_control_test.cpp:30-113 NX=4 NU=2 */
    PMPC_MODEL_PARKING = 2,      /* tests/control/dense_sparse_compare.cpp:22-55 NX=3 NU=2 NP=1 ND=1 */
    PMPC_MODEL_ROBOT_NG = 3,     /* robot + path constraint g = x0^2+x1^2 (NG=1) */
    PMPC_MODEL_KITE_STANDIN = 4, /* SYNTHETIC 13-state/3-input dimension stand-in (kiteNMPF.h is not in the reference) */
    PMPC_MODEL_PARKING_NG = 5    /* tests/control/nonlinear_constraints_test.cpp:31-75  parking + g = u0^2 cos(u1): NP=1, NG=1 */
} pmpc_model;

/* ------------------------------------------------------------------------------------------------------------ */
/* ABI version of THIS header. It changes whenever a struct above changes size or meaning or an entry point changes its signature
 * (1: round 1; 2: linear_solver, flags words, iteration_trace, fp32 QP entries; 3: this query, pmpc_sqp_last_route, multi-device batches;
 * 4: pmpc_sqp_settings::kkt_form).
 * A host must compare pmpc_abi_version() — what the loaded library was built from — with the PMPC_ABI_VERSION it was compiled against, and
 * pmpc_struct_size() with its own sizeof, before passing a settings struct: the *_default() functions write the whole struct of the
 * LIBRARY's layout. Settings structs must always be initialised with pmpc_*_settings_default() and then edited field by field (a struct
 * filled by hand leaves iteration_trace / filter_state / linear_solver undefined; the entry points reject what they can detect — unknown
 * enum values, a trace pointer with a capacity < 1 — with PMPC_ERR_INVALID_ARGUMENT, but a garbage pointer cannot be detected). */
#define PMPC_ABI_VERSION 4
int pmpc_abi_version(void);
/* sizeof of the library's own struct: which = 0 pmpc_qp_settings, 1 pmpc_qp_info, 2 pmpc_sqp_settings, 3 pmpc_sqp_info; 0 for anything else */
unsigned long pmpc_struct_size(int which);
const char* pmpc_version(void);
const char* pmpc_status_string(pmpc_status s);

/* Create a context on HIP device `device` (its own stream). Fails with PMPC_ERR_NO_DEVICE when no GPU is
 * visible: there is no CPU fallback. `stream` may be NULL (context creates one) or an existing hipStream_t. */
pmpc_status pmpc_create(int device, void* stream, pmpc_context** ctx);
pmpc_status pmpc_destroy(pmpc_context* ctx);
pmpc_status pmpc_synchronize(pmpc_context* ctx);
/* Profiling aid (PMPC_PHASE_PROFILE=1 at pmpc_create): shader-clock cycles summed over instances since the last reset:
 * [0] linearisation (+Hessian update) [1] QP [2] line search [3] termination test [4] whole SQP loop [5] BFGS
 * [6] KKT build + factorisation [7] QP residuals [8..23] finer slices (see tests/tools_phase_profile.py). 24 values. */
pmpc_status pmpc_debug_phase_cycles(pmpc_context* ctx, unsigned long long* out24, int reset);
/* Developer harness against uninitialised reads (also PMPC_POISON=1 at pmpc_create): when on, every launch of this context is preceded by a fill of
 * the HBM workspace, the staging buffers of the host-buffer entry points, every compute unit's LDS, every SIMD's register file and the low private
 * segment with signalling NaNs, so that a kernel which reads what it never wrote returns NaN (PMPC_FLAG_NONFINITE) instead of a plausible stale
 * value. The product kernels are not modified: the binaries under test are the shipped ones. Costs about 0.2 ms per launch. */
pmpc_status pmpc_debug_set_poison(pmpc_context* ctx, int on);

/* Which kernel family served the last pmpc_sqp_solve_batch[_dev] / pmpc_mpc_step_batch_dev call of this context (the reference has one code path;
 * here the size and the policy hooks select one of several, with different speed — a caller can log it instead of guessing):
 *   PMPC_ROUTE_REG1  register-resident QP, one KKT row per lane (n + m <= 64, grids with a compiled specialisation)
 *   PMPC_ROUTE_REG2  register-resident QP, two KKT rows per lane (65..128 rows)
 *   PMPC_ROUTE_LDS   KKT factor in LDS (any size that fits; every policy hook)
 *   PMPC_ROUTE_HBM   blocked tile LDL^T with the factor in an HBM workspace (large instances)
 *   PMPC_ROUTE_SCHUR block-structured kernel: Hessian block diagonal per node (hessian_update = 1 or exact Hessians, NP = NG = 0, kkt_form = 0) on a
 *                    grid with a compiled specialisation — per-node blocks in LDS, QP through the m x m Schur complement (m <= 64)
 *   PMPC_ROUTE_CONDREG condensed register-resident QP: 65..128 KKT rows with at most 112 variables and 64 constraint rows (at most one parameter; path constraints included) on a grid with a compiled specialisation,
 *                    kkt_form = 0 — only H + sigma I + rho_box + A' diag(rho) A is inverted (n instead of n + m rows); with the policy hooks (Ruiz preconditioner, filter line search,
 *                    eigenvalue mirroring) on the 7- / 11- / 16-node grids
 * PMPC_ROUTE_NONE before the first call. */
typedef enum { PMPC_ROUTE_NONE = 0, PMPC_ROUTE_REG1 = 1, PMPC_ROUTE_REG2 = 2, PMPC_ROUTE_LDS = 3, PMPC_ROUTE_HBM = 4, PMPC_ROUTE_SCHUR = 5, PMPC_ROUTE_CONDREG = 6 } pmpc_route;
int pmpc_sqp_last_route(pmpc_context* ctx);

void pmpc_qp_settings_default(pmpc_qp_settings* s);      /* qp_base.hpp:17-53 */
void pmpc_qp_settings_sqp_default(pmpc_qp_settings* s);  /* + sqp_base.hpp:83-90 */
void pmpc_sqp_settings_default(pmpc_sqp_settings* s);    /* sqp_base.hpp:24-47 */

/* Chebyshev–Gauss–Lobatto constants (replaces Chebyshev<P>::compute_nodes / compute_int_weights /
 * compute_diff_matrix, src/polynomials/ebyshev.hpp:111-214). nodes, weights: P+1; D: (P+1)^2 column-major. */
pmpc_status pmpc_chebyshev(int P, double* nodes, double* weights, double* D);

/* Batched boxADMM::solve (replaces QPBase::solve -> boxADMM::solve_impl, qp_base.hpp:161-175,
 * box_admm.hpp:81-205). Host buffers; copies in, solves on the GPU, copies out, synchronises.
 * x0 / y0 may be NULL (the 7-argument form: zero guesses, box_admm.hpp:81-86).
 * H: the KKT matrix is built from the LOWER triangle of H only, as Eigen::LDLT<Lower> reads it (helpers.hpp:38-43) — on every kernel family; the
 * residuals use the full matrix, as the reference's H * x does (qp_base.hpp:240-252).
 * Any size: n + m <= 64 and 65..112 rows have register-resident specialisations for the built-in shapes, systems below 112 rows are factorised in LDS,
 * larger ones (the reference's kite size, n + m = 464, included) keep a tiled factor in a per-QP HBM workspace the context owns
 * (about 2 (n+m)^2 x 8 bytes per QP). linear_solver = 1 (pivoted) exists in LDS only: PMPC_ERR_UNSUPPORTED_SIZE beyond ~190 rows. */
pmpc_status pmpc_qp_boxadmm_solve_batch(pmpc_context* ctx, int B, int n, int m, const double* H, const double* h,
                                        const double* A, const double* Alb, const double* Aub, const double* xlb,
                                        const double* xub, const double* x0, const double* y0,
                                        const pmpc_qp_settings* settings, double* x, double* y, pmpc_qp_info* info);

/* Same with DEVICE pointers (inputs already resident in HBM); asynchronous on the context's stream. */
pmpc_status pmpc_qp_boxadmm_solve_batch_dev(pmpc_context* ctx, int B, int n, int m, const double* H, const double* h,
                                            const double* A, const double* Alb, const double* Aub, const double* xlb,
                                            const double* xub, const double* x0, const double* y0,
                                            const pmpc_qp_settings* settings, double* x, double* y, pmpc_qp_info* info);

/* boxADMM<N, M, float>::solve — the single-precision instantiation of the QP solver (QPBase<..., Scalar = float>, qp_base.hpp:94-130; the reference
 * tests it in tests/solvers/qp/box_admm_test.cpp:85-115). float arrays in the layouts above; every quantity of the algorithm is a float as in the
 * reference's templates (settings narrowed to float as qp_solver_settings_t<float> stores them; DIV_BY_ZERO_REGUL = regulariser<float>::value,
 * qp_base.hpp:84-86); the info's floats are returned widened. linear_solver must be 0 (static order). The KKT matrix lives in LDS:
 * PMPC_ERR_UNSUPPORTED_SIZE beyond n + m = 128 or the LDS budget. A plain one-wavefront-per-QP kernel (the SQP solver computes in fp64 only, like every
 * SQP test of the reference). Host buffers / device pointers (asynchronous on the context's stream). */
pmpc_status pmpc_qp_boxadmm_solve_batch_f32(pmpc_context* ctx, int B, int n, int m, const float* H, const float* h, const float* A, const float* Alb,
                                            const float* Aub, const float* xlb, const float* xub, const float* x0, const float* y0,
                                            const pmpc_qp_settings* settings, float* x, float* y, pmpc_qp_info* info);
pmpc_status pmpc_qp_boxadmm_solve_batch_f32_dev(pmpc_context* ctx, int B, int n, int m, const float* H, const float* h, const float* A, const float* Alb,
                                                const float* Aub, const float* xlb, const float* xub, const float* x0, const float* y0,
                                                const pmpc_qp_settings* settings, float* x, float* y, pmpc_qp_info* info);

/* ADMM<N, M, float>::solve — the single-precision instantiation of the OSQP-style solver (admm.hpp; tests/solvers/qp/admm_solver_test.cpp:84-113):
 * as pmpc_qp_boxadmm_solve_batch_f32 with the stacked (2n+m)-row KKT system in LDS (PMPC_ERR_UNSUPPORTED_SIZE beyond 2n + m = 128). */
pmpc_status pmpc_qp_admm_solve_batch_f32(pmpc_context* ctx, int B, int n, int m, const float* H, const float* h, const float* A, const float* Alb,
                                         const float* Aub, const float* xlb, const float* xub, const float* x0, const float* y0,
                                         const pmpc_qp_settings* settings, float* x, float* y, pmpc_qp_info* info);
pmpc_status pmpc_qp_admm_solve_batch_f32_dev(pmpc_context* ctx, int B, int n, int m, const float* H, const float* h, const float* A, const float* Alb,
                                             const float* Aub, const float* xlb, const float* xub, const float* x0, const float* y0,
                                             const pmpc_qp_settings* settings, float* x, float* y, pmpc_qp_info* info);

/* Batched ADMM::solve — the reference's OSQP-style solver (replaces QPBase::solve -> ADMM::solve_impl, admm.hpp:104-212: box
 * constraints stacked under the general ones, one (2n+m)-row KKT system). Same arguments, layouts and dual ordering
 * [general (m) | box (n)] as pmpc_qp_boxadmm_solve_batch. Host buffers / device pointers. */
pmpc_status pmpc_qp_admm_solve_batch(pmpc_context* ctx, int B, int n, int m, const double* H, const double* h, const double* A,
                                     const double* Alb, const double* Aub, const double* xlb, const double* xub, const double* x0,
                                     const double* y0, const pmpc_qp_settings* settings, double* x, double* y, pmpc_qp_info* info);
pmpc_status pmpc_qp_admm_solve_batch_dev(pmpc_context* ctx, int B, int n, int m, const double* H, const double* h, const double* A,
                                         const double* Alb, const double* Aub, const double* xlb, const double* xub, const double* x0,
                                         const double* y0, const pmpc_qp_settings* settings, double* x, double* y, pmpc_qp_info* info);

/* Batched RuizEquilibration<Scalar,N,M,DENSE>::compute (qp_preconditioners.hpp:160-233): scales H, h, A, Alb, Aub, xlb, xub
 * of B QPs IN PLACE and returns the accumulated scalings D (B*n), E (B*m) and the cost scaling c (B). Host buffers. */
pmpc_status pmpc_qp_ruiz_compute_batch(pmpc_context* ctx, int B, int n, int m, double* H, double* h, double* A, double* Alb,
                                       double* Aub, double* xlb, double* xub, double* D, double* E, double* c);
/* Same with DEVICE pointers; asynchronous on the context's stream. */
pmpc_status pmpc_qp_ruiz_compute_batch_dev(pmpc_context* ctx, int B, int n, int m, double* H, double* h, double* A, double* Alb,
                                           double* Aub, double* xlb, double* xub, double* D, double* E, double* c);
/* RuizEquilibration::unscale(x, y) (qp_preconditioners.hpp:359-364): x <- x.*D, y <- (1/c) [y_A.*E ; y_box./D], in place.
 * Host buffers / device pointers. */
pmpc_status pmpc_qp_ruiz_unscale_batch(pmpc_context* ctx, int B, int n, int m, const double* D, const double* E, const double* c,
                                       double* x, double* y);
pmpc_status pmpc_qp_ruiz_unscale_batch_dev(pmpc_context* ctx, int B, int n, int m, const double* D, const double* E, const double* c,
                                           double* x, double* y);

/* Dimensions of the transcription of `model` with Spline<Chebyshev<P>,S> (continuous_ocp.hpp:69-98). */
pmpc_status pmpc_ocp_dims(int model, int P, int S, int* nx, int* nu, int* np, int* nd, int* ng, int* var_size,
                          int* num_eq, int* num_ineq);

/* Batched collocation assembly at given points (replaces ContinuousOCP::lagrangian_gradient_hessian<DENSE> and
 * its callees equalities_linearised / cost_gradient_hessian / cost, continuous_ocp.hpp:797-878,1182-1367,
 * 2100-2174). Host buffers. Any output may be NULL. var: B*n, d: B*ND, lam: B*(m+n) (NULL = zeros).
 * mparams: optional model parameters (robot: {q, r, qn} diagonal weights), may be NULL. */
pmpc_status pmpc_ocp_linearise_batch(pmpc_context* ctx, int model, int P, int S, double t0, double tf,
                                     const double* mparams, int n_mparams, int B, const double* var, const double* d,
                                     const double* lam, double* cost, double* constr, double* jac, double* cost_grad,
                                     double* lag_grad, double* lag_hess);

/* Batched SQPBase::solve (replaces Solver<OCP>::solve() = SQPBase::solve, sqp_base.hpp:569-696, with
 * linearisation :310-318, update_linearisation :490-504 + BFGS_update bfgs.hpp:23-52, step_size_selection
 * :380-419, termination :524-529, and the QP of box_admm.hpp:88-205 fused into one kernel per instance).
 * Host buffers. x_guess/lam_guess may be NULL (zeros, sqp_base.hpp:80-81); lbg/ubg may be NULL when NG == 0.
 * d: B*ND static parameters (Solver::parameters()). lbx/ubx: B*n (Solver::lower/upper_bound_x()). */
pmpc_status pmpc_sqp_solve_batch(pmpc_context* ctx, int model, int P, int S, double t0, double tf,
                                 const double* mparams, int n_mparams, int B, const double* x_guess,
                                 const double* lam_guess, const double* d, const double* lbx, const double* ubx,
                                 const double* lbg, const double* ubg, const pmpc_sqp_settings* sqp_settings,
                                 const pmpc_qp_settings* qp_settings, double* x, double* lam, pmpc_sqp_info* info);

/* The same batch sharded over several contexts — normally one per GPU of the node (instances are independent, the partition is
 * contiguous ranges [k B / n_ctx, (k+1) B / n_ctx) of the instance-major arrays, no collective): one host thread per context stages its shard in,
 * launches on that context's stream and stages the results out, all shards concurrently; the call returns when every shard is back in the host
 * arrays. Arguments as pmpc_sqp_solve_batch. Per-instance device state (filter_state, iteration_trace) belongs to ONE context and is therefore
 * rejected here (PMPC_ERR_INVALID_ARGUMENT). Returns the first error of any shard. Two contexts on the same device are allowed (testing); the
 * SAME context twice is rejected (PMPC_ERR_INVALID_ARGUMENT: one context serves one call at a time). Each context keeps its host thread from its
 * first sharded call until pmpc_destroy, so a control loop pays no thread creation per step. */
pmpc_status pmpc_sqp_solve_batch_multi(pmpc_context* const* ctxs, int n_ctx, int model, int P, int S, double t0, double tf,
                                       const double* mparams, int n_mparams, int B, const double* x_guess,
                                       const double* lam_guess, const double* d, const double* lbx, const double* ubx,
                                       const double* lbg, const double* ubg, const pmpc_sqp_settings* sqp_settings,
                                       const pmpc_qp_settings* qp_settings, double* x, double* lam, pmpc_sqp_info* info);

/* Same with DEVICE pointers; asynchronous on the context's stream. */
pmpc_status pmpc_sqp_solve_batch_dev(pmpc_context* ctx, int model, int P, int S, double t0, double tf,
                                     const double* mparams, int n_mparams, int B, const double* x_guess,
                                     const double* lam_guess, const double* d, const double* lbx, const double* ubx,
                                     const double* lbg, const double* ubg, const pmpc_sqp_settings* sqp_settings,
                                     const pmpc_qp_settings* qp_settings, double* x, double* lam, pmpc_sqp_info* info);

/* LSFilter state of B solver objects in device memory (zeroed = empty filters) for pmpc_sqp_settings::filter_state.
 * clear is LSFilter::clear() for all B (line_search.hpp:52), asynchronous on the context's stream; download copies the B x
 * PMPC_FILTER_STATE_DOUBLES values to the host after synchronising the stream. */
pmpc_status pmpc_filter_state_create(pmpc_context* ctx, int B, double** filter_state);
pmpc_status pmpc_filter_state_clear(pmpc_context* ctx, int B, double* filter_state);
pmpc_status pmpc_filter_state_download(pmpc_context* ctx, int B, const double* filter_state, double* host_out);
pmpc_status pmpc_filter_state_destroy(pmpc_context* ctx, double* filter_state);

/* Iteration records of B solver objects in device memory (zeroed) for pmpc_sqp_settings::iteration_trace — what the reference hands to
 * sqp_settings_t::iteration_callback (sqp_base.hpp:33,685-686), kept per iteration instead of called back. download: B x capacity x PMPC_TRACE_DOUBLES. */
pmpc_status pmpc_iteration_trace_create(pmpc_context* ctx, int B, int capacity, double** trace);
pmpc_status pmpc_iteration_trace_clear(pmpc_context* ctx, int B, int capacity, double* trace);
pmpc_status pmpc_iteration_trace_download(pmpc_context* ctx, int B, int capacity, const double* trace, double* host_out);
pmpc_status pmpc_iteration_trace_destroy(pmpc_context* ctx, double* trace);

/* One receding-horizon step of B MPC<OCP> controllers, everything resident on the device (replaces the caller's loop around
 * MPC::initial_conditions(x0) + MPC::solve() + MPC::solution_u_at(t_start), mpc_wrapper.hpp:89-93, :298, :241-244):
 *   1. pins x0 (B*NX) on the LAST nx entries of the x block of lbx / ubx (in place),
 *   2. solves every OCP warm-started from the CURRENT contents of x (B*n) and lam (B*(m+n)) — zeros for a cold start —
 *      and overwrites them with the new primal / dual solution (Solver::solve() keeps m_x / m_lam between calls),
 *   3. writes the control to apply now, u(t_start) = the last node of the u block, to u0 (B*NU; may be NULL).
 * Device pointers, asynchronous on the context's stream: the plant model / next x0 can be produced on the same stream
 * without any host round trip. */
pmpc_status pmpc_mpc_step_batch_dev(pmpc_context* ctx, int model, int P, int S, double t0, double tf, const double* mparams,
                                    int n_mparams, int B, const double* x0, const double* d, double* lbx, double* ubx,
                                    const double* lbg, const double* ubg, const pmpc_sqp_settings* sqp_settings,
                                    const pmpc_qp_settings* qp_settings, double* x, double* lam, pmpc_sqp_info* info, double* u0);

/* The same for callers that hold no device pointers (plain C / C++ hosts): an opaque batch of B controllers that keeps bounds,
 * static parameters, primal / dual iterate and results in HBM between steps. create uploads the problem data once (x_guess /
 * lam_guess may be NULL: zeros); step uploads only x0 (B*NX), runs pmpc_mpc_step_batch_dev and downloads u(t_start) (B*NU) and,
 * if wanted, the per-instance info; solution downloads the current primal / dual iterate (either pointer may be NULL). */
typedef struct pmpc_mpc_batch pmpc_mpc_batch;
pmpc_status pmpc_mpc_batch_create(pmpc_context* ctx, int model, int P, int S, double t0, double tf, const double* mparams, int n_mparams,
                                  int B, const double* d, const double* lbx, const double* ubx, const double* lbg, const double* ubg,
                                  const double* x_guess, const double* lam_guess, pmpc_mpc_batch** batch);
pmpc_status pmpc_mpc_batch_step(pmpc_mpc_batch* batch, const double* x0, const pmpc_sqp_settings* sqp_settings,
                                const pmpc_qp_settings* qp_settings, double* u0, pmpc_sqp_info* info);
pmpc_status pmpc_mpc_batch_solution(pmpc_mpc_batch* batch, double* x, double* lam);
pmpc_status pmpc_mpc_batch_destroy(pmpc_mpc_batch* batch);

/* ---- user-defined OCPs ---------------------------------------------------------------------------------------------
 * A user's OCP class (the reference's CRTP class with dynamics_impl / lagrange_term_impl / mayer_term_impl /
 * inequality_constraints_impl, continuous_ocp.hpp:191-288) is compiled for the GPU by hipcc in the user's own
 * translation unit with PMPC_REGISTER_OCP(Name) (include/polympc/register_ocp.hpp). The macro emits one C symbol
 * pmpc_user_sqp_dev_<Name>, a pmpc_sqp_dev_fn working on device buffers. pmpc_sqp_solve_batch_user is the matching
 * host-buffer wrapper. `model` points to the user's OCP parameter object (copied by value into the kernel). */
typedef pmpc_status (*pmpc_sqp_dev_fn)(pmpc_context* ctx, const void* model, int P, int S, double t0, double tf, int B,
                                       const double* x_guess, const double* lam_guess, const double* d, const double* lbx,
                                       const double* ubx, const double* lbg, const double* ubg,
                                       const pmpc_sqp_settings* sqp_settings, const pmpc_qp_settings* qp_settings, double* x,
                                       double* lam, pmpc_sqp_info* info);
pmpc_status pmpc_sqp_solve_batch_user(pmpc_context* ctx, pmpc_sqp_dev_fn fn, const void* model, int nx, int nu, int np, int nd,
                                      int ng, int P, int S, double t0, double tf, int B, const double* x_guess,
                                      const double* lam_guess, const double* d, const double* lbx, const double* ubx,
                                      const double* lbg, const double* ubg, const pmpc_sqp_settings* sqp_settings,
                                      const pmpc_qp_settings* qp_settings, double* x, double* lam, pmpc_sqp_info* info);

#ifdef __cplusplus
}
#endif
#endif /* POLYMPC_AMD_H */
