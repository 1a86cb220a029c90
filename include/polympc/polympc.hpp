// polympc_amd — host-side C++ mirror of PolyMPC's OCP / Solver / MPC surface over the C ABI (include/polympc_amd.h).
//
// Header-only, plain C++14, any host compiler; no Eigen. Same names, call shapes and error behaviour (status codes, no
// exceptions on the solve path) as the reference, so code written against the reference reads the same:
//
//   reference                                                             here
//   ---------------------------------------------------------------------------------------------------------------
//   POLYMPC_FORWARD_DECLARATION(Name,NX,NU,NP,ND,NG,T) continuous_ocp.hpp:23-31   same macro (traits only)
//   polympc::Chebyshev<P, GAUSS_LOBATTO, double>       ebyshev.hpp:27-36          same tag type
//   polympc::Spline<Poly, S>                            splines.hpp:22-46          same tag type (NUM_NODES = P*S+1)
//   class Robot : public ContinuousOCP<Robot, Approx, DENSE>  continuous_ocp.hpp:41-98   same CRTP base: enum sizes,
//                                                                                  set_time_limits(), time grid
//   Solver<OCP>::solve(), solve(x_guess, lam_guess), settings(), qp_settings(), info(), primal_solution(),
//   dual_solution(), lower/upper_bound_x(), lower/upper_bound_g(), parameters(), primal_norm(), dual_norm(),
//   constr_violation(), cost(), get_problem()            sqp_base.hpp:159-195,368-374   polympc::Solver<OCP>
//   QPBase::solve(H,h,A,Alb,Aub,xlb,xub[,x0,y0]) -> status_t, primal_solution(), dual_solution(), info(), settings()
//                                                        qp_base.hpp:148-175         polympc::boxADMM<N,M>
//   MPC<OCP,Solver>: initial_conditions, control_bounds, state_bounds, ... solution_x_at(k|t)  mpc_wrapper.hpp:77-298
//                                                                                  polympc::MPC<OCP>
// plus the batch forms this engine exists for: polympc::BatchSolver<OCP> (B independent instances per solve()).
//
// Where the device code of an OCP comes from: the built-in OCPs (polympc::models::*) are compiled into
// libpolympc_amd.so; a user OCP is compiled by hipcc with PMPC_REGISTER_OCP (register_ocp.hpp) and bound here with
// POLYMPC_USE_REGISTERED_OCP(Name). There is no CPU fallback: without a GPU, context creation fails and every solve
// reports the error through polympc::last_error().
#pragma once
#include <algorithm>
#include <array>
#include <cmath>
#include <cstddef>
#include <limits>
#include <memory>
#include <string>
#include <thread>
#include <utility>
#include <type_traits>
#include <vector>
#include "../polympc_amd.h"

namespace polympc {

enum { DENSE = 0, SPARSE = 1 };                      // utils/helpers.hpp:18-21 (only DENSE is on the hot path)
enum collocation_scheme { GAUSS, GAUSS_RADAU, GAUSS_LOBATTO };

template <int PolyOrder, collocation_scheme Qtype = GAUSS_LOBATTO, typename Scalar = double>
struct Chebyshev {
    enum { POLY_ORDER = PolyOrder, NUM_NODES = PolyOrder + 1 };
    using scalar_t = Scalar;
};
template <typename Polynomial, int NumSegments>
struct Spline {
    enum { POLY_ORDER = Polynomial::POLY_ORDER, NUM_SEGMENTS = NumSegments, NUM_NODES = POLY_ORDER * NUM_SEGMENTS + 1 };
    using scalar_t = typename Polynomial::scalar_t;
};

// fixed-size vector with the Eigen accessors the reference's call sites use (Scalar = double, or float for boxADMM<N, M, float>)
template <int N, typename Scalar = double>
struct Vector {
    std::array<Scalar, (N > 0 ? N : 1)> v{};
    static Vector Constant(Scalar c) { Vector r; r.v.fill(c); return r; }
    static Vector Zero() { return Constant(Scalar(0)); }
    Scalar& operator()(int i) { return v[i]; }
    Scalar operator()(int i) const { return v[i]; }
    Scalar& operator[](int i) { return v[i]; }
    Scalar operator[](int i) const { return v[i]; }
    Scalar* data() { return v.data(); }
    const Scalar* data() const { return v.data(); }
    static constexpr int size() { return N; }
    void setZero() { v.fill(Scalar(0)); }
    template <int K> Vector<K, Scalar> segment(int start) const { Vector<K, Scalar> r; for (int i = 0; i < K; ++i) r(i) = v[start + i]; return r; }
    template <int K> void set_segment(int start, const Vector<K, Scalar>& s) { for (int i = 0; i < K; ++i) v[start + i] = s(i); }
    template <int K> Vector<K, Scalar> head() const { return segment<K>(0); }
    template <int K> Vector<K, Scalar> tail() const { return segment<K>(N - K); }
    Scalar lpNormInf() const { Scalar r = 0; for (int i = 0; i < N; ++i) r = std::max(r, std::fabs(v[i])); return r; }
    bool isApprox(const Vector& o, Scalar prec) const {   // Eigen: ||a-b|| <= prec * min(||a||, ||b||)
        Scalar d = 0, a = 0, b = 0;
        for (int i = 0; i < N; ++i) { d += (v[i] - o.v[i]) * (v[i] - o.v[i]); a += v[i] * v[i]; b += o.v[i] * o.v[i]; }
        return std::sqrt(d) <= prec * std::min(std::sqrt(a), std::sqrt(b));
    }
};
// column-major fixed-size matrix (Eigen default)
template <int R, int C, typename Scalar = double>
struct Matrix {
    std::vector<Scalar> v = std::vector<Scalar>((size_t)(R > 0 ? R : 0) * (C > 0 ? C : 0), Scalar(0));
    Scalar& operator()(int i, int j) { return v[i + (size_t)j * R]; }
    Scalar operator()(int i, int j) const { return v[i + (size_t)j * R]; }
    Scalar* data() { return v.data(); }
    const Scalar* data() const { return v.data(); }
};

// status enums with the reference's names and values (qp_base.hpp:55-62, sqp_base.hpp:49-55)
typedef enum { SOLVED = 0, MAX_ITER_EXCEEDED = 1, UNSOLVED = 2, UNINITIALIZED = 3, INFEASIBLE = 4, INCONSISTENT = 5 } status_t;
struct sqp_status_t { enum { SOLVED = 0, MAX_ITER_EXCEEDED = 1, INVALID_SETTINGS = 2 } value; };
struct sqp_info_t { int iter = 0; int qp_solver_iter = 0; sqp_status_t status{sqp_status_t::MAX_ITER_EXCEEDED}; };
struct qp_solver_info_t { status_t status = UNINITIALIZED; int iter = 0; int rho_updates = 0; double rho_estimate = 0, res_prim = 1, res_dual = 1; };

// ---------------------------------------------------------------------------------------------------------------------
// one GPU context per host thread (created on first use); errors are sticky and queryable, never thrown
inline pmpc_status& last_error() { static thread_local pmpc_status e = PMPC_OK; return e; }
struct ContextHolder {
    pmpc_context* ctx = nullptr;
    ~ContextHolder() { if (ctx) pmpc_destroy(ctx); }
};
// the loaded library must come from the header this translation unit was compiled against (struct layouts cross the C ABI by value)
inline bool abi_matches() {
    return pmpc_abi_version() == PMPC_ABI_VERSION && pmpc_struct_size(0) == sizeof(pmpc_qp_settings) && pmpc_struct_size(1) == sizeof(pmpc_qp_info) &&
           pmpc_struct_size(2) == sizeof(pmpc_sqp_settings) && pmpc_struct_size(3) == sizeof(pmpc_sqp_info);
}
inline pmpc_context* context(int device = 0) {
    static thread_local ContextHolder h;
    if (!h.ctx) {
        if (!abi_matches()) { last_error() = PMPC_ERR_ABI_MISMATCH; return nullptr; }
        last_error() = pmpc_create(device, nullptr, &h.ctx);
    }
    return h.ctx;
}

// ---------------------------------------------------------------------------------------------------------------------
// OCP traits + CRTP base (sizes and time grid only: the dynamics live in device code)
template <typename Derived> struct polympc_traits;
#define POLYMPC_FORWARD_DECLARATION(cNAME, cNX, cNU, cNP, cND, cNG, TYPE) \
    class cNAME;                                                          \
    template <> struct polympc::polympc_traits<cNAME> {                   \
        using Scalar = TYPE;                                              \
        enum { NX = cNX, NU = cNU, NP = cNP, ND = cND, NG = cNG };        \
    };

// how an OCP class reaches its device code
template <typename OCP> struct device_binding;   // specialised by POLYMPC_USE_BUILTIN_OCP / POLYMPC_USE_REGISTERED_OCP

template <typename OCP, typename Approximation, int MatrixFormat = DENSE>
class ContinuousOCP {
public:
    static_assert(MatrixFormat == DENSE, "the GPU hot path implements the DENSE members of ContinuousOCP");
    enum {
        NX = polympc_traits<OCP>::NX, NU = polympc_traits<OCP>::NU, NP = polympc_traits<OCP>::NP, ND = polympc_traits<OCP>::ND,
        NG = polympc_traits<OCP>::NG,
        NUM_NODES = Approximation::NUM_NODES, POLY_ORDER = Approximation::POLY_ORDER, NUM_SEGMENTS = Approximation::NUM_SEGMENTS,
        VARX_SIZE = NX * NUM_NODES, VARU_SIZE = NU * NUM_NODES, VARP_SIZE = NP, VARD_SIZE = ND,
        VAR_SIZE = VARX_SIZE + VARU_SIZE + VARP_SIZE, NUM_EQ = VARX_SIZE, NUM_INEQ = NG * NUM_NODES, NUM_BOX = VAR_SIZE,
        DUAL_SIZE = NUM_EQ + NUM_INEQ + NUM_BOX, is_sparse = 0, is_dense = 1, MATRIXFMT = MatrixFormat
    };
    using scalar_t = double;
    using nlp_variable_t = Vector<VAR_SIZE>;
    using nlp_dual_t = Vector<DUAL_SIZE>;
    using nlp_ineq_constraints_t = Vector<NUM_INEQ>;
    using static_parameter_t = Vector<ND>;
    using time_t = Vector<NUM_NODES>;

    double t_start{0}, t_stop{1};
    time_t time_nodes;   // descending: node 0 = t_stop (continuous_ocp.hpp:50-55)

    ContinuousOCP() { set_time_limits(0.0, 1.0); }
    void set_time_limits(const double& t0, const double& tf) noexcept {   // continuous_ocp.hpp:147-159
        t_start = t0; t_stop = tf;
        const int P = POLY_ORDER, S = NUM_SEGMENTS;
        const double t_length = (t_stop - t_start) / S, t_shift = t_length / 2;
        for (int i = 0; i < S; ++i)
            for (int j = 0; j <= P; ++j)
                time_nodes(i * P + j) = (t_length / 2) * std::cos(double(P - j) * (M_PI / P)) + (t_start + t_shift + i * t_length);
        std::reverse(time_nodes.v.begin(), time_nodes.v.begin() + NUM_NODES);
    }
};

struct sqp_settings_t {   // sqp_base.hpp:24-47 (+ the two override points as flags)
    double tau = 0.5, eta = 0.25, rho = 0.5, eps_prim = 1e-3, eps_dual = 1e-3;
    int max_iter = 100, line_search_max_iter = 100;
    int regularisation = 0;            // 0: default no-op hook (sqp_base.hpp:305); 2: Gershgorin (dense_sparse_compare.cpp:109-122)
    bool exact_hessian_every_iter = false;
    int preconditioner = 0;            // SQPBase's Preconditioner template argument: 0 IdentityPreconditioner, 1 RuizEquilibration
    int hessian_update = 0;            // hessian_update_impl: 0 dense damped BFGS, 1 ContinuousOCP's block BFGS
    int qp_solver = 0;                 // QPSolver template argument: 0 boxADMM, 1 ADMM (OSQP form)
    int line_search = 0;               // step_size_selection_impl: 0 l1-merit backtracking (sqp_base.hpp:380-419), 1 the filter line search
                                       // of valet_parking_mpc_test.cpp:116-158 on LSFilter (line_search.hpp:31-98)
    int kkt_form = 0;                  // pmpc_sqp_settings::kkt_form: 0 the kernels' default (large instances solve the condensed n x n system), 1 the
                                       // reference's quasi-definite (n + m)-row KKT matrix of box_admm.hpp:209-223 on every route, 2 the block-structured
                                       // range-space form wherever it is compiled — including the bordered NP = 1 form (parking, 11 nodes), which is not a default
    void (*iteration_callback)(void* solver) = nullptr;   // sqp_base.hpp:33, called at :685-686 once per iteration from the second one on. The fused
                                       // kernel records what the callback can read (pmpc_sqp_settings::iteration_trace); Solver<OCP>::solve() then
                                       // calls it once per recorded iteration with info().iter, primal_norm(), dual_norm(), cost() and
                                       // constr_violation() of THAT iteration (the iterates themselves are not retained: primal_solution() is the final one)
};

// LSFilter's tunables under the reference's member names (`solver.filter.beta = 0.1`, valet_parking_mpc_test.cpp:192); the list itself
// lives in HBM, one per solver object of the batch, and is carried from one solve() to the next as the reference's member is.
struct LSFilterHandle {
    int max_depth = PMPC_FILTER_MAX_DEPTH;
    double beta = 1e-5;
    LSFilterHandle() = default;
    LSFilterHandle(const LSFilterHandle& o) : max_depth(o.max_depth), beta(o.beta) {}   // a copied solver starts with empty filters
    // assignment: the settings of `o`, and EMPTY filters — the device list this object held (sized for ITS batch, filled by ITS solves) is released, so
    // that the next bind() creates one for the batch the assigned-to solver now has (a list kept across an assignment from a larger batch was indexed
    // out of bounds by the kernel)
    LSFilterHandle& operator=(const LSFilterHandle& o) { if (this != &o) { release(); max_depth = o.max_depth; beta = o.beta; } return *this; }
    ~LSFilterHandle() { release(); }
    void clear() noexcept { if (m_dev) pmpc_filter_state_clear(m_ctx, m_B, m_dev); }    // LSFilter::clear(), line_search.hpp:52
    // number of pairs and the pairs (cost, violation), newest first, of instance b (downloaded on demand)
    std::vector<std::pair<double, double>> entries(int b) const {
        std::vector<std::pair<double, double>> out;
        if (!m_dev || b < 0 || b >= m_B) return out;
        std::vector<double> st((size_t)m_B * PMPC_FILTER_STATE_DOUBLES);
        if (pmpc_filter_state_download(m_ctx, m_B, m_dev, st.data()) != PMPC_OK) return out;
        const double* f = &st[(size_t)b * PMPC_FILTER_STATE_DOUBLES];
        for (int i = 0; i < (int)f[0]; ++i) out.emplace_back(f[1 + 2 * i], f[2 + 2 * i]);
        return out;
    }
    // fills the C settings; allocates the device list on first use
    pmpc_status bind(pmpc_context* ctx, int B, int line_search, pmpc_sqp_settings& ss) noexcept {
        ss.line_search = line_search; ss.filter_max_depth = max_depth; ss.filter_beta = beta; ss.filter_state = nullptr;
        if (line_search != 1) return PMPC_OK;
        if (m_dev && (m_B != B || m_ctx != ctx)) release();   // (another batch size or context: a fresh, empty list)
        if (!m_dev) {
            const pmpc_status st = pmpc_filter_state_create(ctx, B, &m_dev);
            if (st != PMPC_OK) return st;
            m_ctx = ctx; m_B = B;
        }
        ss.filter_state = m_dev;
        return PMPC_OK;
    }
private:
    void release() noexcept { if (m_dev) pmpc_filter_state_destroy(m_ctx, m_dev); m_dev = nullptr; m_ctx = nullptr; m_B = 0; }
    double* m_dev = nullptr; pmpc_context* m_ctx = nullptr; int m_B = 0;
};
using qp_solver_settings_t = pmpc_qp_settings;   // same member names as qp_base.hpp:17-53 (ADMM subset)

// ---------------------------------------------------------------------------------------------------------------------
// B independent instances of Solver<OCP> solved by one kernel launch
template <typename OCP>
class BatchSolver {
public:
    enum { VAR_SIZE = OCP::VAR_SIZE, NUM_EQ = OCP::NUM_EQ, NUM_INEQ = OCP::NUM_INEQ, DUAL_SIZE = OCP::DUAL_SIZE, ND = OCP::ND };
    explicit BatchSolver(int batch) : B(batch) {
        const double INF = std::numeric_limits<double>::infinity();
        m_x.assign((size_t)B * VAR_SIZE, 0.0); m_lam.assign((size_t)B * DUAL_SIZE, 0.0);
        m_lbx.assign((size_t)B * VAR_SIZE, -INF); m_ubx.assign((size_t)B * VAR_SIZE, INF);
        m_lbg.assign((size_t)B * NUM_INEQ, -INF); m_ubg.assign((size_t)B * NUM_INEQ, INF);
        m_p.assign((size_t)B * (ND > 0 ? ND : 1), 0.0);
        m_info.resize(B);
        pmpc_qp_settings_sqp_default(&m_qp_settings);   // SQPBase constructor overrides, sqp_base.hpp:83-90
    }
    // The reference's SQPBase owns its state by value and is copyable (sqp_base.hpp:127-152). A copy takes the problem, the settings, the bounds, the
    // iterates and the infos; it does NOT take the device contexts of set_devices() (they own streams and workspaces, and two solver objects must not
    // share them): the copy solves on the calling thread's context until it is given devices of its own, and starts with empty filters.
    BatchSolver(const BatchSolver& o)
        : problem(o.problem), B(o.B), m_x(o.m_x), m_lam(o.m_lam), m_lbx(o.m_lbx), m_ubx(o.m_ubx), m_lbg(o.m_lbg), m_ubg(o.m_ubg), m_p(o.m_p),
          m_info(o.m_info), m_trace(o.m_trace), m_trace_capacity(o.m_trace_capacity), m_settings(o.m_settings), m_qp_settings(o.m_qp_settings),
          filter(o.filter) {}
    BatchSolver& operator=(const BatchSolver& o) {
        if (this != &o) {
            problem = o.problem; B = o.B; m_x = o.m_x; m_lam = o.m_lam; m_lbx = o.m_lbx; m_ubx = o.m_ubx; m_lbg = o.m_lbg; m_ubg = o.m_ubg; m_p = o.m_p;
            m_info = o.m_info; m_trace = o.m_trace; m_trace_capacity = o.m_trace_capacity; m_settings = o.m_settings; m_qp_settings = o.m_qp_settings;
            filter = o.filter; m_multi.clear();
        }
        return *this;
    }
    BatchSolver(BatchSolver&&) = default;
    BatchSolver& operator=(BatchSolver&&) = default;
    int batch() const { return B; }
    // SURVEY 8e — several devices: the batch is split into contiguous shards, one per entry of `devices` (one context, stream and host thread each;
    // listing a device twice gives it two shards: useful on a one-GPU box and in tests), no collective. An empty list returns to the calling
    // thread's own context. Per-instance device state of one context (the carried LSFilter, iteration records) is not available in this mode.
    pmpc_status set_devices(const std::vector<int>& devices) noexcept {
        m_multi.clear();
        if (!devices.empty() && !abi_matches()) return last_error() = PMPC_ERR_ABI_MISMATCH;
        for (int dv : devices) {
            m_multi.emplace_back(new ContextHolder());
            const pmpc_status st = pmpc_create(dv, nullptr, &m_multi.back()->ctx);
            if (st != PMPC_OK) { m_multi.clear(); return last_error() = st; }
        }
        return PMPC_OK;
    }
    int num_shards() const noexcept { return m_multi.empty() ? 1 : (int)m_multi.size(); }
    OCP& get_problem() noexcept { return problem; }
    sqp_settings_t& settings() noexcept { return m_settings; }
    qp_solver_settings_t& qp_settings() noexcept { return m_qp_settings; }
    double* primal_solution(int b) noexcept { return &m_x[(size_t)b * VAR_SIZE]; }
    double* dual_solution(int b) noexcept { return &m_lam[(size_t)b * DUAL_SIZE]; }
    double* lower_bound_x(int b) noexcept { return &m_lbx[(size_t)b * VAR_SIZE]; }
    double* upper_bound_x(int b) noexcept { return &m_ubx[(size_t)b * VAR_SIZE]; }
    double* lower_bound_g(int b) noexcept { return &m_lbg[(size_t)b * NUM_INEQ]; }
    double* upper_bound_g(int b) noexcept { return &m_ubg[(size_t)b * NUM_INEQ]; }
    double* parameters(int b) noexcept { return &m_p[(size_t)b * (ND > 0 ? ND : 1)]; }
    const pmpc_sqp_info& info(int b) const noexcept { return m_info[b]; }

    // SQPBase::solve for every instance; the current primal/dual arrays are the initial guess (sqp_base.hpp:368-374)
    pmpc_status solve() noexcept {
        if (!m_multi.empty()) return solve_sharded();
        pmpc_context* ctx = context();
        if (!ctx) return last_error();
        pmpc_sqp_settings ss;
        pmpc_sqp_settings_default(&ss);
        ss.tau = m_settings.tau; ss.eta = m_settings.eta; ss.rho = m_settings.rho; ss.eps_prim = m_settings.eps_prim;
        ss.eps_dual = m_settings.eps_dual; ss.max_iter = m_settings.max_iter; ss.line_search_max_iter = m_settings.line_search_max_iter;
        ss.regularisation = m_settings.regularisation; ss.exact_hessian_every_iter = m_settings.exact_hessian_every_iter ? 1 : 0;
        ss.preconditioner = m_settings.preconditioner; ss.hessian_update = m_settings.hessian_update; ss.qp_solver = m_settings.qp_solver;
        ss.kkt_form = m_settings.kkt_form;
        { const pmpc_status fs = filter.bind(ctx, B, m_settings.line_search, ss); if (fs != PMPC_OK) return last_error() = fs; }
        double* trace_dev = nullptr;
        const int cap = m_settings.max_iter;
        if (m_settings.iteration_callback != nullptr && cap > 0) {   // records for the callbacks (replayed by the caller of solve())
            const pmpc_status ts = pmpc_iteration_trace_create(ctx, B, cap, &trace_dev);
            if (ts != PMPC_OK) return last_error() = ts;
            ss.iteration_trace = trace_dev; ss.iteration_trace_capacity = cap;
        }
        m_trace.clear(); m_trace_capacity = 0;
        std::vector<double> xo(m_x.size()), lo(m_lam.size());
        const pmpc_status st = device_binding<OCP>::solve(ctx, problem, OCP::POLY_ORDER, OCP::NUM_SEGMENTS, problem.t_start, problem.t_stop, B,
                                                          m_x.data(), m_lam.data(), m_p.data(), m_lbx.data(), m_ubx.data(),
                                                          (NUM_INEQ > 0) ? m_lbg.data() : nullptr, (NUM_INEQ > 0) ? m_ubg.data() : nullptr, &ss,
                                                          &m_qp_settings, xo.data(), lo.data(), m_info.data());
        last_error() = st;
        if (st == PMPC_OK) { m_x.swap(xo); m_lam.swap(lo); }
        if (trace_dev) {
            if (st == PMPC_OK) {
                m_trace.assign((size_t)B * cap * PMPC_TRACE_DOUBLES, 0.0);
                if (pmpc_iteration_trace_download(ctx, B, cap, trace_dev, m_trace.data()) == PMPC_OK) m_trace_capacity = cap; else m_trace.clear();
            }
            pmpc_iteration_trace_destroy(ctx, trace_dev);
        }
        return st;
    }
    // the same solve with the batch in contiguous shards [k B / N, (k+1) B / N) over the N contexts of set_devices(), one host thread per shard
    pmpc_status solve_sharded() noexcept {
        // per-instance device state of ONE context cannot follow the shards: no iteration records, and the filter line search runs with a
        // filter that lives for the solve only (filter_state = nullptr — the C entry's rule), not the list carried between solve() calls
        if (m_settings.iteration_callback != nullptr) return last_error() = PMPC_ERR_INVALID_ARGUMENT;
        pmpc_sqp_settings ss;
        pmpc_sqp_settings_default(&ss);
        ss.tau = m_settings.tau; ss.eta = m_settings.eta; ss.rho = m_settings.rho; ss.eps_prim = m_settings.eps_prim;
        ss.eps_dual = m_settings.eps_dual; ss.max_iter = m_settings.max_iter; ss.line_search_max_iter = m_settings.line_search_max_iter;
        ss.regularisation = m_settings.regularisation; ss.exact_hessian_every_iter = m_settings.exact_hessian_every_iter ? 1 : 0;
        ss.preconditioner = m_settings.preconditioner; ss.hessian_update = m_settings.hessian_update; ss.qp_solver = m_settings.qp_solver;
        ss.kkt_form = m_settings.kkt_form;
        ss.line_search = m_settings.line_search; ss.filter_max_depth = filter.max_depth; ss.filter_beta = filter.beta; ss.filter_state = nullptr;
        m_trace.clear(); m_trace_capacity = 0;
        const int N = (int)m_multi.size();
        std::vector<double> xo(m_x.size()), lo(m_lam.size());
        std::vector<pmpc_status> st((size_t)N, PMPC_OK);
        std::vector<std::thread> th;
        for (int k = 0; k < N; ++k) {
            const long long b0 = (long long)B * k / N, b1 = (long long)B * (k + 1) / N;
            if (b1 <= b0) continue;
            th.emplace_back([&, k, b0, b1]() {
                const size_t o = (size_t)b0;
                st[k] = device_binding<OCP>::solve(m_multi[k]->ctx, problem, OCP::POLY_ORDER, OCP::NUM_SEGMENTS, problem.t_start, problem.t_stop, (int)(b1 - b0),
                                                   &m_x[o * VAR_SIZE], &m_lam[o * DUAL_SIZE], &m_p[o * (ND > 0 ? ND : 1)], &m_lbx[o * VAR_SIZE], &m_ubx[o * VAR_SIZE],
                                                   (NUM_INEQ > 0) ? &m_lbg[o * NUM_INEQ] : nullptr, (NUM_INEQ > 0) ? &m_ubg[o * NUM_INEQ] : nullptr, &ss, &m_qp_settings,
                                                   &xo[o * VAR_SIZE], &lo[o * DUAL_SIZE], &m_info[o]);
            });
        }
        for (auto& t : th) t.join();
        for (int k = 0; k < N; ++k) if (st[k] != PMPC_OK) return last_error() = st[k];
        m_x.swap(xo); m_lam.swap(lo);
        return last_error() = PMPC_OK;
    }
    // record of SQP iteration `iter` (1-based) of instance b: [iter, alpha, primal_norm, dual_norm, cost, qp iterations, qp status, max violation];
    // null when nothing was recorded (no iteration_callback set, or the iteration did not run)
    const double* iteration_record(int b, int iter) const noexcept {
        if (iter < 1 || iter > m_trace_capacity || b < 0 || b >= B) return nullptr;
        const double* r = &m_trace[((size_t)b * m_trace_capacity + (iter - 1)) * PMPC_TRACE_DOUBLES];
        return r[0] == (double)iter ? r : nullptr;
    }

    OCP problem;
    int B;
    std::vector<double> m_x, m_lam, m_lbx, m_ubx, m_lbg, m_ubg, m_p;
    std::vector<pmpc_sqp_info> m_info;
    std::vector<double> m_trace; int m_trace_capacity = 0;
    std::vector<std::unique_ptr<ContextHolder>> m_multi;   // set_devices(): one context per shard
    sqp_settings_t m_settings;
    qp_solver_settings_t m_qp_settings;
    LSFilterHandle filter;   // `MySolver::filter` of valet_parking_mpc_test.cpp:114 for every instance (used when settings().line_search == 1)
};

// ---------------------------------------------------------------------------------------------------------------------
// Solver<OCP>: the single-instance surface of SQPBase (sqp_base.hpp:159-195, :368-374)
template <typename OCP>
class Solver {
public:
    enum { VAR_SIZE = OCP::VAR_SIZE, NUM_EQ = OCP::NUM_EQ, NUM_INEQ = OCP::NUM_INEQ, NUM_CONSTR = OCP::DUAL_SIZE };
    using scalar_t = double;
    using nlp_variable_t = typename OCP::nlp_variable_t;
    using nlp_dual_t = typename OCP::nlp_dual_t;
    using nlp_ineq_constraints_t = typename OCP::nlp_ineq_constraints_t;
    using parameter_t = typename OCP::static_parameter_t;
    using nlp_settings_t = sqp_settings_t;
    using nlp_info_t = sqp_info_t;

    Solver() : m_batch(1) {
        const double INF = std::numeric_limits<double>::infinity();
        m_lbx = nlp_variable_t::Constant(-INF); m_ubx = nlp_variable_t::Constant(INF);
        m_lbg = nlp_ineq_constraints_t::Constant(-INF); m_ubg = nlp_ineq_constraints_t::Constant(INF);
        m_x.setZero(); m_lam.setZero(); m_p.setZero();
    }
    const OCP& get_problem() const noexcept { return m_batch.problem; }
    OCP& get_problem() noexcept { return m_batch.problem; }
    const nlp_variable_t& primal_solution() const noexcept { return m_x; }
    nlp_variable_t& primal_solution() noexcept { return m_x; }
    const nlp_dual_t& dual_solution() const noexcept { return m_lam; }
    nlp_dual_t& dual_solution() noexcept { return m_lam; }
    nlp_settings_t& settings() noexcept { return m_batch.settings(); }
    qp_solver_settings_t& qp_settings() noexcept { return m_batch.qp_settings(); }
    const sqp_info_t& info() const noexcept { return m_info; }
    nlp_variable_t& lower_bound_x() noexcept { return m_lbx; }
    nlp_variable_t& upper_bound_x() noexcept { return m_ubx; }
    nlp_ineq_constraints_t& lower_bound_g() noexcept { return m_lbg; }
    nlp_ineq_constraints_t& upper_bound_g() noexcept { return m_ubg; }
    parameter_t& parameters() noexcept { return m_p; }
    double primal_norm() const noexcept { return m_primal_norm; }
    double dual_norm() const noexcept { return m_dual_norm; }
    double constr_violation() const noexcept { return m_max_violation; }
    double cost() const noexcept { return m_cost; }
    /** not in the reference: the device's information word for the last solve — PMPC_FLAG_NONFINITE (a non-finite value went through: do not trust the numbers) and
     *  PMPC_FLAG_ILLCOND (information only: the instance was solved again in the full KKT form behind a conditioning gate), include/polympc_amd.h */
    int info_flags() const noexcept { return m_batch.info(0).flags; }

    void solve() noexcept {
        std::copy(m_x.data(), m_x.data() + VAR_SIZE, m_batch.primal_solution(0));
        std::copy(m_lam.data(), m_lam.data() + OCP::DUAL_SIZE, m_batch.dual_solution(0));
        std::copy(m_lbx.data(), m_lbx.data() + VAR_SIZE, m_batch.lower_bound_x(0));
        std::copy(m_ubx.data(), m_ubx.data() + VAR_SIZE, m_batch.upper_bound_x(0));
        std::copy(m_lbg.data(), m_lbg.data() + NUM_INEQ, m_batch.lower_bound_g(0));
        std::copy(m_ubg.data(), m_ubg.data() + NUM_INEQ, m_batch.upper_bound_g(0));
        std::copy(m_p.data(), m_p.data() + OCP::ND, m_batch.parameters(0));
        m_info.status.value = sqp_status_t::MAX_ITER_EXCEEDED;
        if (m_batch.solve() != PMPC_OK) { m_info.status.value = sqp_status_t::INVALID_SETTINGS; return; }
        std::copy(m_batch.primal_solution(0), m_batch.primal_solution(0) + VAR_SIZE, m_x.data());
        std::copy(m_batch.dual_solution(0), m_batch.dual_solution(0) + OCP::DUAL_SIZE, m_lam.data());
        const pmpc_sqp_info& i = m_batch.info(0);
        if (m_batch.settings().iteration_callback != nullptr) {   // sqp_base.hpp:685-686: from the second iteration on, before that iteration's termination test
            int qp_sum = 0;
            if (const double* r1 = m_batch.iteration_record(0, 1)) qp_sum = (int)r1[5];
            for (int k = 2; k <= i.iter; ++k) {
                const double* r = m_batch.iteration_record(0, k);
                if (!r) break;
                qp_sum += (int)r[5];
                m_info.iter = k; m_info.qp_solver_iter = qp_sum;
                m_primal_norm = r[2]; m_dual_norm = r[3]; m_cost = r[4]; m_max_violation = r[7];
                m_batch.settings().iteration_callback(this);
            }
        }
        m_info.iter = i.iter; m_info.qp_solver_iter = i.qp_solver_iter;
        m_info.status.value = i.status == PMPC_SQP_SOLVED ? sqp_status_t::SOLVED : sqp_status_t::MAX_ITER_EXCEEDED;
        m_primal_norm = i.primal_norm; m_dual_norm = i.dual_norm; m_max_violation = i.max_violation; m_cost = i.cost;
    }
    void solve(const nlp_variable_t& x_guess, const nlp_dual_t& lam_guess) noexcept { m_x = x_guess; m_lam = lam_guess; solve(); }
    // copies (the reference's SQPBase is copyable): `filter` below is a reference into this object's own m_batch — a memberwise copy would leave
    // the copy's reference on the ORIGINAL's filter, and a reference member deletes the implicit assignment
    Solver(const Solver& o)
        : m_x(o.m_x), m_lbx(o.m_lbx), m_ubx(o.m_ubx), m_lam(o.m_lam), m_lbg(o.m_lbg), m_ubg(o.m_ubg), m_p(o.m_p), m_info(o.m_info),
          m_primal_norm(o.m_primal_norm), m_dual_norm(o.m_dual_norm), m_max_violation(o.m_max_violation), m_cost(o.m_cost), m_batch(o.m_batch) {}
    Solver& operator=(const Solver& o) {
        if (this != &o) {
            m_x = o.m_x; m_lbx = o.m_lbx; m_ubx = o.m_ubx; m_lam = o.m_lam; m_lbg = o.m_lbg; m_ubg = o.m_ubg; m_p = o.m_p; m_info = o.m_info;
            m_primal_norm = o.m_primal_norm; m_dual_norm = o.m_dual_norm; m_max_violation = o.m_max_violation; m_cost = o.m_cost; m_batch = o.m_batch;
        }
        return *this;
    }

    nlp_variable_t m_x, m_lbx, m_ubx;
    nlp_dual_t m_lam;
    nlp_ineq_constraints_t m_lbg, m_ubg;
    parameter_t m_p;
    sqp_info_t m_info;
    double m_primal_norm = 0, m_dual_norm = 0, m_max_violation = 0, m_cost = 0;
private:
    BatchSolver<OCP> m_batch;
public:
    LSFilterHandle& filter = m_batch.filter;   // `solver.filter.beta = 0.1`, valet_parking_mpc_test.cpp:192 (settings().line_search = 1)
};

// ---------------------------------------------------------------------------------------------------------------------
// boxADMM<N, M>: the QP seam (qp_base.hpp:148-175, box_admm.hpp:81-91). Matrices column-major.
// Scalar = float selects the single-precision instantiation the reference tests (box_admm_test.cpp:85-115): pmpc_qp_boxadmm_solve_batch_f32.
template <int N, int M, typename Scalar> class ADMM;
template <int N, int M, typename Scalar = double>
class boxADMM {
    static_assert(std::is_same<Scalar, double>::value || std::is_same<Scalar, float>::value, "boxADMM: Scalar is double or float");
public:
    using scalar_t = Scalar;
    using qp_var_t = Vector<N, Scalar>; using qp_dual_t = Vector<N + M, Scalar>; using qp_dual_a_t = Vector<M, Scalar>;
    using qp_hessian_t = Matrix<N, N, Scalar>; using qp_constraint_t = Matrix<M, N, Scalar>;
    using settings_t = qp_solver_settings_t; using info_t = qp_solver_info_t;
    boxADMM() { pmpc_qp_settings_default(&m_settings); }
    settings_t& settings() noexcept { return m_settings; }
    const info_t& info() const noexcept { return m_info; }
    const qp_var_t& primal_solution() const noexcept { return m_x; }
    const qp_dual_t& dual_solution() const noexcept { return m_y; }
    int iter{0};

    status_t solve(const qp_hessian_t& H, const qp_var_t& h, const qp_constraint_t& A, const qp_dual_a_t& Alb, const qp_dual_a_t& Aub,
                   const qp_var_t& xlb, const qp_var_t& xub) noexcept {
        return solve_(H, h, A, Alb, Aub, xlb, xub, nullptr, nullptr);
    }
    status_t solve(const qp_hessian_t& H, const qp_var_t& h, const qp_constraint_t& A, const qp_dual_a_t& Alb, const qp_dual_a_t& Aub,
                   const qp_var_t& xlb, const qp_var_t& xub, const qp_var_t& x_guess, const qp_dual_t& y_guess) noexcept {
        return solve_(H, h, A, Alb, Aub, xlb, xub, x_guess.data(), y_guess.data());
    }
private:
    status_t solve_(const qp_hessian_t& H, const qp_var_t& h, const qp_constraint_t& A, const qp_dual_a_t& Alb, const qp_dual_a_t& Aub,
                    const qp_var_t& xlb, const qp_var_t& xub, const Scalar* x0, const Scalar* y0) noexcept {
        pmpc_context* ctx = context();
        m_info.status = UNINITIALIZED;
        if (!ctx) return m_info.status;
        pmpc_qp_info qi;
        const pmpc_status st = entry_(m_osqp_form, ctx, H.data(), h.data(), A.data(), Alb.data(), Aub.data(), xlb.data(), xub.data(), x0, y0, &m_settings,
                                      m_x.data(), m_y.data(), &qi);
        last_error() = st;
        if (st != PMPC_OK) return m_info.status;
        m_info.status = (status_t)qi.status; m_info.iter = qi.iter; m_info.rho_updates = qi.rho_updates;
        m_info.rho_estimate = qi.rho_estimate; m_info.res_prim = qi.res_prim; m_info.res_dual = qi.res_dual;
        iter = qi.iter;
        return m_info.status;
    }
    // the C-ABI entry for the scalar type (overloads on the pointer type: the header stays C++14)
    static pmpc_status entry_(bool osqp, pmpc_context* ctx, const double* H, const double* h, const double* A, const double* Alb, const double* Aub,
                              const double* xlb, const double* xub, const double* x0, const double* y0, const pmpc_qp_settings* s, double* x, double* y,
                              pmpc_qp_info* qi) noexcept {
        return (osqp ? pmpc_qp_admm_solve_batch : pmpc_qp_boxadmm_solve_batch)(ctx, 1, N, M, H, h, A, Alb, Aub, xlb, xub, x0, y0, s, x, y, qi);
    }
    static pmpc_status entry_(bool osqp, pmpc_context* ctx, const float* H, const float* h, const float* A, const float* Alb, const float* Aub,
                              const float* xlb, const float* xub, const float* x0, const float* y0, const pmpc_qp_settings* s, float* x, float* y,
                              pmpc_qp_info* qi) noexcept {
        return (osqp ? pmpc_qp_admm_solve_batch_f32 : pmpc_qp_boxadmm_solve_batch_f32)(ctx, 1, N, M, H, h, A, Alb, Aub, xlb, xub, x0, y0, s, x, y, qi);
    }
    settings_t m_settings; info_t m_info; qp_var_t m_x; qp_dual_t m_y;
    bool m_osqp_form{false};
    friend class ADMM<N, M, Scalar>;
};
// ADMM<N, M>: the reference's OSQP-style solver (admm.hpp) — same seam, the stacked (2N+M)-row KKT system on the device
template <int N, int M, typename Scalar = double>
class ADMM : public boxADMM<N, M, Scalar> {
public:
    ADMM() { this->m_osqp_form = true; }
};

// ---------------------------------------------------------------------------------------------------------------------
// RuizEquilibration<N, M> (qp_preconditioners.hpp:114-385, DENSE): compute() scales the QP data in place on the device,
// unscale(x, y) maps a solution of the scaled QP back (usage: tests/solvers/qp/box_admm_test.cpp:47-83).
template <int N, int M>
class RuizEquilibration {
public:
    Vector<N> D; Vector<M> E;
    RuizEquilibration() { for (int i = 0; i < N; ++i) D(i) = 1.0; for (int i = 0; i < M; ++i) E(i) = 1.0; }
    const double& c() const noexcept { return m_c; }
    void compute(Matrix<N, N>& H, Vector<N>& h, Matrix<M, N>& A, Vector<M>& Al, Vector<M>& Au, Vector<N>& l, Vector<N>& u) noexcept {
        pmpc_context* ctx = context();
        if (!ctx) return;
        last_error() = pmpc_qp_ruiz_compute_batch(ctx, 1, N, M, H.data(), h.data(), A.data(), Al.data(), Au.data(), l.data(), u.data(),
                                                  D.data(), E.data(), &m_c);
    }
    void unscale(Vector<N>& x, Vector<N + M>& y) const noexcept {
        pmpc_context* ctx = context();
        if (!ctx) return;
        last_error() = pmpc_qp_ruiz_unscale_batch(ctx, 1, N, M, D.data(), E.data(), &m_c, x.data(), y.data());
    }
private:
    double m_c{1.0};
};

// ---------------------------------------------------------------------------------------------------------------------
// MPC<OCP>: the controller façade (mpc_wrapper.hpp:18-300) incl. Lagrange interpolation of the solution
// (LagrangeSpline, splines.hpp:101-139)
template <typename OCP>
class MPC {
    using nlp_solver_t = Solver<OCP>;
    nlp_solver_t m_solver;
public:
    static constexpr int nx = OCP::NX, nu = OCP::NU, np = OCP::NP, nd = OCP::ND, ng = OCP::NG;
    static constexpr int var_size = OCP::VAR_SIZE, varx_size = OCP::VARX_SIZE, varu_size = OCP::VARU_SIZE, dual_size = OCP::DUAL_SIZE;
    static constexpr int num_nodes = OCP::NUM_NODES, num_segms = OCP::NUM_SEGMENTS, num_ineq = OCP::NUM_INEQ, npn = OCP::POLY_ORDER + 1;
    using state_t = Vector<nx>; using control_t = Vector<nu>; using parameter_t = Vector<np>; using static_param = Vector<nd>;
    using constraint_t = Vector<ng>; using traj_state_t = Vector<varx_size>; using traj_control_t = Vector<varu_size>;
    using dual_var_t = Vector<dual_size>;

    MPC() { update_grid(); }
    void set_time_limits(const double& t0, const double& tf) noexcept { m_solver.get_problem().set_time_limits(t0, tf); update_grid(); }
    // initial condition = equality box on the LAST nx entries of the x block: node 0 is the final time (mpc_wrapper.hpp:89-93)
    void initial_conditions(const state_t& x0) noexcept { initial_conditions(x0, x0); }
    void initial_conditions(const state_t& x0_lb, const state_t& x0_ub) noexcept {
        for (int i = 0; i < nx; ++i) { m_solver.lower_bound_x()(varx_size - nx + i) = x0_lb(i); m_solver.upper_bound_x()(varx_size - nx + i) = x0_ub(i); }
    }
    void state_bounds(const state_t& xlb, const state_t& xub) noexcept {
        for (int k = 0; k < num_nodes - 1; ++k) for (int i = 0; i < nx; ++i) { m_solver.lower_bound_x()(k * nx + i) = xlb(i); m_solver.upper_bound_x()(k * nx + i) = xub(i); }
    }
    void final_state_bounds(const state_t& xlb, const state_t& xub) noexcept {
        for (int i = 0; i < nx; ++i) { m_solver.lower_bound_x()(i) = xlb(i); m_solver.upper_bound_x()(i) = xub(i); }
    }
    void control_bounds(const control_t& lb, const control_t& ub) noexcept {
        for (int k = 0; k < num_nodes; ++k) for (int i = 0; i < nu; ++i) { m_solver.lower_bound_x()(varx_size + k * nu + i) = lb(i); m_solver.upper_bound_x()(varx_size + k * nu + i) = ub(i); }
    }
    void constraints_bounds(const constraint_t& lbg, const constraint_t& ubg) noexcept {
        for (int k = 0; k < num_nodes; ++k) for (int i = 0; i < ng; ++i) { m_solver.lower_bound_g()(k * ng + i) = lbg(i); m_solver.upper_bound_g()(k * ng + i) = ubg(i); }
    }
    void parameters_bounds(const parameter_t& lbp, const parameter_t& ubp) noexcept {
        for (int i = 0; i < np; ++i) { m_solver.lower_bound_x()(var_size - np + i) = lbp(i); m_solver.upper_bound_x()(var_size - np + i) = ubp(i); }
    }
    void set_static_parameters(const static_param& param) noexcept { m_solver.parameters() = param; }
    void x_guess(const traj_state_t& g) noexcept { for (int i = 0; i < varx_size; ++i) m_solver.m_x(i) = g(i); }
    void u_guess(const traj_control_t& g) noexcept { for (int i = 0; i < varu_size; ++i) m_solver.m_x(varx_size + i) = g(i); }
    void lam_guess(const dual_var_t& g) noexcept { m_solver.m_lam = g; }
    void p_guess(const parameter_t& g) noexcept { for (int i = 0; i < np; ++i) m_solver.m_x(var_size - np + i) = g(i); }

    sqp_settings_t& settings() noexcept { return m_solver.settings(); }
    qp_solver_settings_t& qp_settings() noexcept { return m_solver.qp_settings(); }
    const sqp_info_t& info() const noexcept { return m_solver.info(); }
    nlp_solver_t& solver() noexcept { return m_solver; }
    OCP& ocp() noexcept { return m_solver.get_problem(); }
    double primal_norm() const noexcept { return m_solver.primal_norm(); }
    double dual_norm() const noexcept { return m_solver.dual_norm(); }
    double constr_violation() const noexcept { return m_solver.constr_violation(); }
    double cost() const noexcept { return m_solver.cost(); }

    traj_state_t solution_x() const noexcept { return m_solver.primal_solution().template segment<varx_size>(0); }
    traj_control_t solution_u() const noexcept { return m_solver.primal_solution().template segment<varu_size>(varx_size); }
    parameter_t solution_p() const noexcept { return m_solver.primal_solution().template tail<np>(); }
    dual_var_t solution_dual() const noexcept { return m_solver.dual_solution(); }
    // k-th collocation point counted FORWARD in time (mpc_wrapper.hpp:241-244, :267-270)
    state_t solution_x_at(const int& k) const noexcept { return m_solver.primal_solution().template segment<nx>(varx_size - (k + 1) * nx); }
    control_t solution_u_at(const int& k) const noexcept { return m_solver.primal_solution().template segment<nu>(varx_size + varu_size - (k + 1) * nu); }
    // Lagrange interpolation inside the segment containing t (mpc_wrapper.hpp:245-281)
    state_t solution_x_at(const double& t) const noexcept { return interpolate<nx>(t, 0, [this](int k) { return solution_x_at(k); }); }
    control_t solution_u_at(const double& t) const noexcept { return interpolate<nu>(t, 0, [this](int k) { return solution_u_at(k); }); }

    void solve() noexcept { m_solver.solve(); }

private:
    std::array<double, OCP::POLY_ORDER + 1> m_seg_nodes{};   // forward-time nodes of the first segment
    double sgm_length{1.0};
    void update_grid() {
        const OCP& o = m_solver.get_problem();
        for (int j = 0; j < npn; ++j) m_seg_nodes[j] = o.time_nodes(num_nodes - 1 - j);
        sgm_length = (o.t_stop - o.t_start) / num_segms;
    }
    template <int K, class F> Vector<K> interpolate(double t, int, F at) const {
        int idx = (int)std::floor(t / sgm_length);
        idx = std::max(0, std::min(idx, num_segms - 1));
        const double tl = t - idx * sgm_length;     // the reference evaluates the first-segment basis at the local time
        Vector<K> r = Vector<K>::Zero();
        for (int i = 0; i < npn; ++i) {
            double li = 1.0;
            for (int j = 0; j < npn; ++j) if (j != i) li *= (tl - m_seg_nodes[j]) / (m_seg_nodes[i] - m_seg_nodes[j]);
            const Vector<K> vi = at(idx * OCP::POLY_ORDER + i);
            for (int c = 0; c < K; ++c) r(c) += li * vi(c);
        }
        return r;
    }
};

// ---------------------------------------------------------------------------------------------------------------------
// BatchMPC<OCP>: B independent MPC<OCP> controllers whose bounds, static parameters and primal / dual iterate stay in HBM between
// steps (pmpc_mpc_batch_*). Per step only the measured states go up and the controls to apply come down. Built-in OCPs.
template <typename OCP>
class BatchMPC {
public:
    static constexpr int nx = OCP::NX, nu = OCP::NU, nd = OCP::ND, var_size = OCP::VAR_SIZE, varx_size = OCP::VARX_SIZE,
                         num_nodes = OCP::NUM_NODES, num_ineq = OCP::NUM_INEQ, dual_size = OCP::DUAL_SIZE;
    explicit BatchMPC(int batch) : B(batch) {
        const double INF = std::numeric_limits<double>::infinity();
        m_lbx.assign((size_t)B * var_size, -INF); m_ubx.assign((size_t)B * var_size, INF);
        m_lbg.assign((size_t)B * (num_ineq > 0 ? num_ineq : 1), -INF); m_ubg.assign((size_t)B * (num_ineq > 0 ? num_ineq : 1), INF);
        m_p.assign((size_t)B * (nd > 0 ? nd : 1), 0.0);
        m_info.resize(B);
        pmpc_qp_settings_sqp_default(&m_qp_settings);
    }
    ~BatchMPC() { if (m_batch) pmpc_mpc_batch_destroy(m_batch); }
    BatchMPC(const BatchMPC&) = delete; BatchMPC& operator=(const BatchMPC&) = delete;
    OCP& ocp() noexcept { return problem; }
    sqp_settings_t& settings() noexcept { return m_settings; }
    qp_solver_settings_t& qp_settings() noexcept { return m_qp_settings; }
    void set_time_limits(const double& t0, const double& tf) noexcept { problem.set_time_limits(t0, tf); }
    // the setters of mpc_wrapper.hpp:103-187 for instance b; they are uploaded by the first step()
    void control_bounds(int b, const Vector<nu>& lb, const Vector<nu>& ub) noexcept {
        for (int k = 0; k < num_nodes; ++k) for (int i = 0; i < nu; ++i) { lbx(b)[varx_size + k * nu + i] = lb(i); ubx(b)[varx_size + k * nu + i] = ub(i); }
    }
    void state_bounds(int b, const Vector<nx>& lb, const Vector<nx>& ub) noexcept {
        for (int k = 0; k < num_nodes - 1; ++k) for (int i = 0; i < nx; ++i) { lbx(b)[k * nx + i] = lb(i); ubx(b)[k * nx + i] = ub(i); }
    }
    void final_state_bounds(int b, const Vector<nx>& lb, const Vector<nx>& ub) noexcept { for (int i = 0; i < nx; ++i) { lbx(b)[i] = lb(i); ubx(b)[i] = ub(i); } }
    void set_static_parameters(int b, const Vector<nd>& p) noexcept { for (int i = 0; i < nd; ++i) m_p[(size_t)b * (nd > 0 ? nd : 1) + i] = p(i); }
    // initial_conditions(x0) + solve() + solution_u_at(t_start) for every controller: x0 holds B*nx states, u0 receives B*nu controls
    pmpc_status step(const double* x0, double* u0) noexcept {
        pmpc_context* ctx = context();
        if (!ctx) return last_error();
        if (!m_batch) {
            const std::vector<double> mp = problem.model_params();
            const pmpc_status st = pmpc_mpc_batch_create(ctx, device_binding<OCP>::model_id, OCP::POLY_ORDER, OCP::NUM_SEGMENTS, problem.t_start, problem.t_stop,
                                                         mp.empty() ? nullptr : mp.data(), (int)mp.size(), B, m_p.data(), m_lbx.data(), m_ubx.data(),
                                                         num_ineq > 0 ? m_lbg.data() : nullptr, num_ineq > 0 ? m_ubg.data() : nullptr, nullptr, nullptr, &m_batch);
            if (st != PMPC_OK) return last_error() = st;
        }
        pmpc_sqp_settings ss;
        pmpc_sqp_settings_default(&ss);
        ss.tau = m_settings.tau; ss.eta = m_settings.eta; ss.rho = m_settings.rho; ss.eps_prim = m_settings.eps_prim; ss.eps_dual = m_settings.eps_dual;
        ss.max_iter = m_settings.max_iter; ss.line_search_max_iter = m_settings.line_search_max_iter; ss.regularisation = m_settings.regularisation;
        ss.exact_hessian_every_iter = m_settings.exact_hessian_every_iter ? 1 : 0; ss.preconditioner = m_settings.preconditioner;
        ss.hessian_update = m_settings.hessian_update; ss.qp_solver = m_settings.qp_solver;
        { const pmpc_status fs = filter.bind(ctx, B, m_settings.line_search, ss); if (fs != PMPC_OK) return last_error() = fs; }
        return last_error() = pmpc_mpc_batch_step(m_batch, x0, &ss, &m_qp_settings, u0, m_info.data());
    }
    const pmpc_sqp_info& info(int b) const noexcept { return m_info[b]; }
    // current primal solution of every controller (B * var_size), downloaded on demand
    pmpc_status solution(std::vector<double>& x) noexcept {
        if (!m_batch) return PMPC_ERR_INVALID_ARGUMENT;
        x.resize((size_t)B * var_size);
        return last_error() = pmpc_mpc_batch_solution(m_batch, x.data(), nullptr);
    }
private:
    double* lbx(int b) noexcept { return &m_lbx[(size_t)b * var_size]; }
    double* ubx(int b) noexcept { return &m_ubx[(size_t)b * var_size]; }
    OCP problem; int B;
    std::vector<double> m_lbx, m_ubx, m_lbg, m_ubg, m_p;
    std::vector<pmpc_sqp_info> m_info;
    sqp_settings_t m_settings; qp_solver_settings_t m_qp_settings;
    pmpc_mpc_batch* m_batch{nullptr};
public:
    LSFilterHandle filter;   // the solvers' LSFilter members (used when settings().line_search == 1)
};

// ---------------------------------------------------------------------------------------------------------------------
// device bindings
#define POLYMPC_USE_BUILTIN_OCP(Name, MODEL_ID)                                                                                       \
    template <> struct polympc::device_binding<Name> {                                                                                \
        static pmpc_status solve(pmpc_context* ctx, const Name& ocp, int P, int S, double t0, double tf, int B, const double* xg,     \
                                 const double* lg, const double* d, const double* lbx, const double* ubx, const double* lbg,          \
                                 const double* ubg, const pmpc_sqp_settings* ss, const pmpc_qp_settings* qs, double* x, double* lam,  \
                                 pmpc_sqp_info* info) {                                                                               \
            const std::vector<double> mp = ocp.model_params();                                                                        \
            return pmpc_sqp_solve_batch(ctx, MODEL_ID, P, S, t0, tf, mp.empty() ? nullptr : mp.data(), (int)mp.size(), B, xg, lg, d,  \
                                        lbx, ubx, lbg, ubg, ss, qs, x, lam, info);                                                    \
        }                                                                                                                             \
    };

// DeviceModel = the plain struct registered with PMPC_REGISTER_OCP in the .hip translation unit; the host OCP class exposes
// `DeviceModel device_model() const` returning the parameter object that is copied into the kernel.
#define POLYMPC_USE_REGISTERED_OCP(Name, DeviceModel)                                                                                 \
    extern "C" pmpc_status pmpc_user_sqp_dev_##DeviceModel(pmpc_context*, const void*, int, int, double, double, int, const double*,  \
                                                           const double*, const double*, const double*, const double*, const double*, \
                                                           const double*, const pmpc_sqp_settings*, const pmpc_qp_settings*, double*, \
                                                           double*, pmpc_sqp_info*);                                                  \
    template <> struct polympc::device_binding<Name> {                                                                                \
        static pmpc_status solve(pmpc_context* ctx, const Name& ocp, int P, int S, double t0, double tf, int B, const double* xg,     \
                                 const double* lg, const double* d, const double* lbx, const double* ubx, const double* lbg,          \
                                 const double* ubg, const pmpc_sqp_settings* ss, const pmpc_qp_settings* qs, double* x, double* lam,  \
                                 pmpc_sqp_info* info) {                                                                               \
            const auto dm = ocp.device_model();                                                                                       \
            return pmpc_sqp_solve_batch_user(ctx, pmpc_user_sqp_dev_##DeviceModel, &dm, Name::NX, Name::NU, Name::NP, Name::ND,       \
                                             Name::NG, P, S, t0, tf, B, xg, lg, d, lbx, ubx, lbg, ubg, ss, qs, x, lam, info);         \
        }                                                                                                                             \
    };

}  // namespace polympc

// ---------------------------------------------------------------------------------------------------------------------
// built-in OCPs (device code inside libpolympc_amd.so)
namespace polympc { namespace models {
template <typename Approximation> class MobileRobot;
template <typename Approximation> class CSTR;
}}
namespace polympc {
template <typename A> struct polympc_traits<models::MobileRobot<A>> { using Scalar = double; enum { NX = 3, NU = 2, NP = 0, ND = 1, NG = 0 }; };
template <typename A> struct polympc_traits<models::CSTR<A>> { using Scalar = double; enum { NX = 4, NU = 2, NP = 0, ND = 0, NG = 0 }; };
namespace models {
// tests/control/mpc_wrapper_test.cpp:33-80
template <typename Approximation>
class MobileRobot : public ContinuousOCP<MobileRobot<Approximation>, Approximation, DENSE> {
public:
    double q = 1.0, r = 1.0, qn = 1.0;
    void set_Q_coeff(const double& c) { q = c; }
    std::vector<double> model_params() const { return {q, r, qn}; }
};
// tests/control/cstr_control_test.cpp:30-113
template <typename Approximation>
class CSTR : public ContinuousOCP<CSTR<Approximation>, Approximation, DENSE> {
public:
    CSTR() { this->set_time_limits(0, 100); }
    std::vector<double> model_params() const { return {}; }
};
}  // namespace models
template <typename A> struct device_binding<models::MobileRobot<A>> {
    static constexpr int model_id = PMPC_MODEL_ROBOT;   // built-in device code: usable with BatchMPC
    static pmpc_status solve(pmpc_context* ctx, const models::MobileRobot<A>& ocp, int P, int S, double t0, double tf, int B, const double* xg,
                             const double* lg, const double* d, const double* lbx, const double* ubx, const double* lbg, const double* ubg,
                             const pmpc_sqp_settings* ss, const pmpc_qp_settings* qs, double* x, double* lam, pmpc_sqp_info* info) {
        const std::vector<double> mp = ocp.model_params();
        return pmpc_sqp_solve_batch(ctx, PMPC_MODEL_ROBOT, P, S, t0, tf, mp.data(), (int)mp.size(), B, xg, lg, d, lbx, ubx, lbg, ubg, ss, qs, x, lam, info);
    }
};
template <typename A> struct device_binding<models::CSTR<A>> {
    static constexpr int model_id = PMPC_MODEL_CSTR;
    static pmpc_status solve(pmpc_context* ctx, const models::CSTR<A>&, int P, int S, double t0, double tf, int B, const double* xg,
                             const double* lg, const double* d, const double* lbx, const double* ubx, const double* lbg, const double* ubg,
                             const pmpc_sqp_settings* ss, const pmpc_qp_settings* qs, double* x, double* lam, pmpc_sqp_info* info) {
        return pmpc_sqp_solve_batch(ctx, PMPC_MODEL_CSTR, P, S, t0, tf, nullptr, 0, B, xg, lg, d, lbx, ubx, lbg, ubg, ss, qs, x, lam, info);
    }
};
}  // namespace polympc
