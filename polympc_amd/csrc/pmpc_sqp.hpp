// polympc_amd — fused per-instance SQP loop on the device (one wavefront per OCP instance).
//
// Replaces SQPBase::solve and its default policies (/root/reference/src/solvers/sqp_base.hpp): solve :569-696,
// linearisation_dense_impl :310-318, update_linearisation_dense_impl :490-504, solve_qp :533-565,
// step_size_selection_impl :380-419, constraints_violation_impl :423-444, max_constraints_violation_impl :448-474,
// termination_criteria_impl :524-529; BFGS_update (src/solvers/bfgs.hpp:23-52); the Gershgorin regulariser of
// tests/control/dense_sparse_compare.cpp:109-122. Linearisation, Hessian update, KKT factorisation, ADMM iterations,
// line search and step all run inside ONE kernel launch; the iterate, multipliers, bounds and QP vectors never leave
// LDS between SQP iterations. H (n x n) and the Jacobian (m x n) live in a per-instance HBM workspace that stays
// L2-resident (config A: 15.7 KB per instance).
#pragma once
#include <hip/hip_runtime.h>
#include "pmpc_ocp.hpp"
#include "pmpc_qp.hpp"
#include "pmpc_jview.hpp"
#include "pmpc_qp_reg.hpp"
#include "pmpc_qp_reg2.hpp"
#include "pmpc_qp_big.hpp"
#include "pmpc_qp_schur.hpp"
#include "pmpc_qp_cond.hpp"
#include "pmpc_ruiz.hpp"
#include "pmpc_admm.hpp"

#ifndef PMPC_EXPERIMENT_FORCE_SYMLOWER
#define PMPC_EXPERIMENT_FORCE_SYMLOWER 0   /* 1: developer experiment (round 6, EXPERIMENTS.md) — the headline kernel's KKT build reads the LOWER triangle of H only (one address select per load): what any
                                              half / packed storage of the BFGS matrix would cost on the read side */
#endif
namespace pmpc {

struct SqpLds {
    double *x, *lam, *lam_k, *h, *lg, *lgn, *al, *au, *lx, *ux, *lbx, *ubx, *lbg, *ubg, *step, *xs, *cb, *t1, *t2, *t3;
    __host__ __device__ static size_t doubles(int n, int m, int mi) {
        return 12 * (size_t)n + 2 * (size_t)(m + n) + 3 * (size_t)m + 2 * (size_t)mi + 3 * (size_t)(n + m) + 8;
    }
    __device__ __forceinline__ double* carve(double* p, int n, int m, int mi) {
        x = p; p += n; lam = p; p += m + n; lam_k = p; p += m + n; h = p; p += n; lg = p; p += n; lgn = p; p += n;
        al = p; p += m; au = p; p += m; lx = p; p += n; ux = p; p += n; lbx = p; p += n; ubx = p; p += n;
        lbg = p; p += mi; ubg = p; p += mi; step = p; p += n; xs = p; p += n; cb = p; p += m;
        t1 = p; p += n + m; t2 = p; p += n + m; t3 = p; p += n + m;
        return p;
    }
};

constexpr double DBL_EPS = 2.220446049250313e-16;
constexpr int RUIZ_MAX_NDER = 64;   // see SqpDevice::qp_and_step
constexpr int PMPC_SQP_IN_PROGRESS = 3;   // internal: the instance continues in the next iteration-slice launch
constexpr int PMPC_SQP_REDO = 4;          // internal: a QP of a condensed kernel gave up at its conditioning gate (PMPC_FLAG_ILLCOND) — the launcher's redo launch solves the instance again in the full KKT form
constexpr int PMPC_REDO_MODE = -0x7fffffff - 1;   // it_begin of that redo launch: only instances with status PMPC_SQP_REDO run, from their guesses

// NN, MM > 0: compile-time QP size -> register-resident QP: NN+MM <= 64 one KKT row per lane (pmpc_qp_reg.hpp, REG1), 65..112 two rows per
// lane (pmpc_qp_reg2.hpp, REG2: the QP alone is specialised, the other phases run the size-generic code with compile-time trip counts);
// 0 -> LDS-resident QP
// PROF: accumulate per-phase shader-clock cycles (separate kernel instantiation; costs 16+ VGPRs, off by default)
// HU: Hessian-update policy compiled into a register-resident specialisation (0 dense damped BFGS, 1 block BFGS); the LDS-resident
// kernels (NN == 0) select it at run time from settings.hessian_update
// BIG: large-instance mode — the KKT factor is the tiled HBM workspace of pmpc_qp_big.hpp
// POL: register-resident specialisation that carries the policy hooks the reference's tests install beside the default ones — the Ruiz
//      preconditioner (qp_preconditioners.hpp:114-220) and the filter line search (line_search.hpp:31-98); the LDS / HBM-resident kernels (NN == 0)
//      always carry them. A separate instantiation: the default register kernels stay free of the (cold) calls and their spills.
// PS: P * 256 + S of a BLOCK-STRUCTURED specialisation (pmpc_qp_schur.hpp; 0 = none): the Hessian is block diagonal per node (block BFGS or exact
//      Hessians, NG = 0; NP = 1: the arrow shape, `hbrd` holds the border row with the corner and the border column) and lives as per-node blocks in LDS (`hblk`), J as its per-node blocks (`ocp.jblk`) + the differentiation matrix: the
//      HBM workspace is not touched at all, and the QP is solved through the m x m Schur complement. NN, MM are the compile-time sizes there too.
//      PS = -1: the CONDENSED register specialisation (pmpc_qp_cond.hpp) of a two-rows-per-lane kernel — everything as REG2 except the QP.
template <class Model, int NN = 0, int MM = 0, bool PROF = false, int HU = 0, bool BIG = false, bool POL = false, int PS = 0>
struct SqpDevice {
    static constexpr bool SCH = PS > 0;
    static constexpr bool SLIM = PS == -2;   // large-instance mode compiled for two wavefronts per SIMD (256 registers)
    static constexpr int BIG_NW = (PS == -3) ? 4 : 1;   // large-instance mode on a workgroup of four wavefronts (BigTeam, pmpc_qp_big.hpp): this object lives on the first one
    void* big_mail = nullptr;                           // the team's mailbox (LDS)
    static constexpr bool CND = PS == -1;   // condensed register QP (pmpc_qp_cond.hpp): a two-rows-per-lane kernel (REG2) whose QP inverts S = H + sigma I + rho_box + A' diag(rho) A only
    static constexpr int SCH_P = PS / 256, SCH_S = PS % 256;
    static constexpr bool HOOKS = (NN == 0) || POL;
    using Dm = OcpDims<Model>;
    // Ruiz scaling is compiled into the kernels that carry the hooks (HOOKS: the LDS / HBM-resident kernels and the hook builds, POL, of the register kernels): in the default
    // register-resident kernels its three (cold, out-of-line) calls cost private-memory frames and call-ABI spills on the hot path.
    // History of the condensed kernels (PS = -1): until late round 6 the launcher never routed preconditioner = 1 to them (Ruiz rescales the workspace the per-node blocks of A mirror),
    // the calls were dead code there — and in round 4 / 5 they were what MISCOMPILED the hook build of the small condensed kernel (robot 11 nodes, 55 + 33, one row per lane: with the three
    // never-executed calls compiled in, the QP step of the primal-only lanes was lost — a VGPR -> AGPR copy of the lane id placed inside a partial-EXEC block, DESIGN.md hazard 3 /
    // EXPERIMENTS.md round 6), so round 5 compiled them out. -DPMPC_EXPERIMENT_CND_WITH_RUIZ restores exactly that round-5 build (developer switch: reproduces the fault on hipcc 7.2
    // when the allocator makes the same choice).
#ifdef PMPC_EXPERIMENT_CND_WITH_RUIZ
    static constexpr bool RUIZ_COMPILED = HOOKS && (int)Dm::NDER <= RUIZ_MAX_NDER;
#else
    // (late round 6: the hook builds of the condensed kernels carry the Ruiz calls again — their tables come from the scaled workspace, pmpc_qp_cond.hpp WS; the fault of round 5 is one
    //  the CPU suite now detects in the built code, tests/test_kernel_occupancy_cpu.py)
    static constexpr bool RUIZ_COMPILED = HOOKS && (int)Dm::NDER <= RUIZ_MAX_NDER && PS <= 0;   // (PS > 0: the block-structured kernel has no dense workspace to scale — of the hooks it carries the filter line search only)
#endif
    static constexpr bool REG1 = !SCH && NN > 0 && NN + MM <= WAVE;    // one KKT row per lane
    static constexpr bool REG2 = !SCH && NN > 0 && NN + MM > WAVE;     // two KKT rows per lane
    double *hblk = nullptr, *hbrd = nullptr /* NP = 1: border row + corner, border column (pmpc_qp_schur.hpp) */, *qblk = nullptr, *xsc = nullptr, *dsc = nullptr, *dtab = nullptr;   // SCH: Hessian blocks, Q blocks, the exchange vectors and the D~ tables of the QP (LDS)
    static constexpr int MEMCH = BIG ? BIG_MEM_BATCH : 8;      // loads in flight per lane in the row walks over the BFGS matrix in HBM
    Ocp<Model>& ocp;
    SqpLds& v;
    QpLds& qw;
    double* lsbuf = nullptr;   // LDS scratch of the side-by-side line search (aliases the MFMA staging, free outside the QP)
    bool ls_side_by_side = false;   // lsbuf holds G >= 2 candidates
    bool cb_valid = false;     // v.cb holds the constraint values of the CURRENT iterate (set by the line search)
    double* tr = nullptr;  // LDS transpose scratch of the register-resident QP (aliases the per-node AD staging, dead during the QP)
    double* Hw;  // H(i,j) = Hw[j*ldw + i]  — upper block of the stacked (n+m) x n HBM workspace [H ; J]
    double* Aw;  // J(r,j) = Aw[j*ldw + r]  — lower block (Aw = Hw + n)
    int ldw;     // n + m
    const pmpc_sqp_settings& ss;
    const pmpc_qp_settings& qs;
    int n, m, me, mi;
    // the same sizes as compile-time constants on the register path (NN, MM > 0): loops over them unroll, their uniform guards
    // fold away and the LDS loads of a reduction are issued in one batch
    static constexpr int NNODES_CT_ = (NN > 0) ? MM / (Model::NX + Model::NG) : 0;
    __device__ __forceinline__ int n_ct() const { if constexpr (NN > 0) return NN; else return n; }
    __device__ __forceinline__ int m_ct() const { if constexpr (NN > 0) return MM; else return m; }
    __device__ __forceinline__ int me_ct() const { if constexpr (NN > 0) return Model::NX * NNODES_CT_; else return me; }
    __device__ __forceinline__ int mi_ct() const { if constexpr (NN > 0) return Model::NG * NNODES_CT_; else return mi; }
    // block-sparse view of J (pmpc_jview.hpp) — register-resident kernels: the launcher carved ocp.jblk / ocp.gblk behind the staging
    using JV = JView<Model, (NNODES_CT_ > 0 ? NNODES_CT_ : 1)>;
    __device__ __forceinline__ JV jview() const { return JV{ocp.s.D, ocp.s.nsr, ocp.jblk, ocp.gblk, ocp.P}; }
    double cost_log = 0.0, primal_norm = 0.0, dual_norm = 0.0, max_violation = 0.0;
    double alpha_log = 0.0; int qp_iter_last = 0, qp_status_last = 0;   // for the iteration records
    double* trace = nullptr;   // this instance's records (pmpc_sqp_settings::iteration_trace), or null
    int qp_iter_total = 0;
    int qp_flags = 0;            // OR of the QP solves' flags (PMPC_FLAG_NONFINITE)
    static constexpr int FLAG_GAVE_UP = 0x100;   // internal bit of qp_flags: the QP that just ran gave up at its conditioning gate (never reported)
    long long cyc[PROF ? 24 : 1] = {0};
    __device__ __forceinline__ static long long now() { if constexpr (PROF) return clock64(); else return 0; }
    __device__ __forceinline__ void acc(int i, long long dt) { if constexpr (PROF) cyc[i] += dt; }   // shader-clock cycles: 0 linearise(+BFGS) 1 QP 2 line search 3 termination 4 total 5 BFGS 6 KKT build+factor 7 QP residuals 8 ls node evaluation 9 ls scalar sums 10 first-order staging 11 second-order staging 12 first-order assembly 13 Hessian assembly 14 Lagrangian gradient 16..20 KKT inverse: row loads+staging, panel moves, sweeps, MFMA updates, final conversion 21 ls prologue (mu, grad'p) 22 ls acceptance

#ifdef PMPC_EXPERIMENT_WG_STAMPS
    long long t_start_stamp = wall_clock64();
#endif
    __device__ SqpDevice(Ocp<Model>& o, SqpLds& v_, QpLds& q_, double* H_, double* A_, const pmpc_sqp_settings& s, const pmpc_qp_settings& q)
        : ocp(o), v(v_), qw(q_), Hw(H_), Aw(A_), ldw(o.dm.n + o.dm.m), ss(s), qs(q), n(o.dm.n), m(o.dm.m), me(o.dm.me), mi(o.dm.mi) {}

    // constraints_violation_impl :423-444 (sequential sums, reference association order)
    __device__ __forceinline__ double constraints_violation(const double* xx) {
        ocp.constraints(xx, v.cb);
        double cl1 = DBL_EPS;
        double s = 0.0;
        for (int i = 0; i < me; ++i) s += fabs(v.cb[i]);
        cl1 += s;
        s = 0.0; for (int i = 0; i < mi; ++i) s += fmax(v.lbg[i] - v.cb[me + i], 0.0); cl1 += s;
        s = 0.0; for (int i = 0; i < mi; ++i) s += fmax(v.cb[me + i] - v.ubg[i], 0.0); cl1 += s;
        s = 0.0; for (int i = 0; i < n; ++i) s += fmax(v.lbx[i] - xx[i], 0.0); cl1 += s;
        s = 0.0; for (int i = 0; i < n; ++i) s += fmax(xx[i] - v.ubx[i], 0.0); cl1 += s;
        wsync();
        return cl1;
    }
    // max_constraints_violation_impl :448-474
    __device__ __forceinline__ double max_constraints_violation(const double* xx) {
        if (!cb_valid) ocp.constraints(xx, v.cb);   // otherwise the accepted line-search candidate already evaluated them at this point
        const int ln = lane_id();
        const int n = n_ct(), me = me_ct(), mi = mi_ct();
        double c = 0.0, a = -INFINITY, b = -INFINITY, e = -INFINITY, f = -INFINITY;
        for (int i = ln; i < me; i += WAVE) c = fmax(c, fabs(v.cb[i]));
        for (int i = ln; i < mi; i += WAVE) { a = fmax(a, v.lbg[i] - v.cb[me + i]); b = fmax(b, v.cb[me + i] - v.ubg[i]); }
        for (int i = ln; i < n; i += WAVE) { e = fmax(e, v.lbx[i] - xx[i]); f = fmax(f, xx[i] - v.ubx[i]); }
        // one reduction of the per-lane maxima (max is exact and order-free)
        c = fmax(fmax(c, fmax(a, b)), fmax(e, f));
        c = wave_max(c);
        wsync();
        return c;
    }

    // ---- LSFilter, src/solvers/line_search.hpp:31-98, for the filter line search of valet_parking_mpc_test.cpp:116-158 (line_search = 1).
    // filt = [count, (cost, constraint violation) pairs, newest first] in LDS. Every lane evaluates the same acceptance test on the
    // same values; lane 0 alone edits the list. Compiled into the LDS-resident kernels only (the launcher routes line_search = 1 there).
    double* filt = nullptr;
    __device__ __forceinline__ bool filter_mode() const { if constexpr (HOOKS) return __builtin_amdgcn_readfirstlane(ss.line_search) == 1; else return false; }
    __device__ __forceinline__ bool filter_is_acceptable(double cost, double constraint) const {   // :65-74
        int cnt = (int)filt[0];
        if (cnt > PMPC_FILTER_MAX_DEPTH) cnt = PMPC_FILTER_MAX_DEPTH;
        const double beta = ss.filter_beta;
        bool ok = true;
        for (int i = 0; i < cnt; ++i) {
            const double f = filt[1 + 2 * i], c = filt[2 + 2 * i];
            const double bc = beta * c;
            if (((f - bc) <= cost) && ((c - bc) <= constraint)) ok = false;
        }
        return __builtin_amdgcn_readfirstlane((int)ok) != 0;
    }
    __device__ __forceinline__ void filter_add(double cost, double constraint) {                   // :76-92 (remove_if(dominated_by) :14-29 keeps the order)
        wsync();
        if (lane_id() == 0) {
            int cnt = (int)filt[0];
            if (cnt > PMPC_FILTER_MAX_DEPTH) cnt = PMPC_FILTER_MAX_DEPTH;
            if (cnt < ss.filter_max_depth) {
                int k = 0;
                for (int i = 0; i < cnt; ++i) {
                    const double f = filt[1 + 2 * i], c = filt[2 + 2 * i];
                    if (!((f >= cost) && (c >= constraint))) { filt[1 + 2 * k] = f; filt[2 + 2 * k] = c; ++k; }
                }
                cnt = k + 1;
            }
            // emplace_front; when the filter was full its last entry falls off (pop_back)
            for (int i = cnt - 1; i > 0; --i) { filt[1 + 2 * i] = filt[2 * i - 1]; filt[2 + 2 * i] = filt[2 * i]; }
            filt[1] = cost; filt[2] = constraint;
            for (int i = cnt; i < PMPC_FILTER_MAX_DEPTH; ++i) { filt[1 + 2 * i] = 0.0; filt[2 + 2 * i] = 0.0; }
            filt[0] = (double)cnt;
        }
        wsync();
    }

    // step_size_selection_impl :380-419 ; p = QP primal step in qw.x  (one trial point at a time)
    __device__ __forceinline__ double step_size_selection_serial() {
        const double* p = qw.x;
        const int ln = lane_id();
        const double constr_l1 = constraints_violation(v.x);
        const double mu = lds_inf_norm(v.lam_k, m + n);
        const double cost_1 = ocp.cost(v.x);
        const double phi_l1 = cost_1 + mu * constr_l1;
        const double Dp_phi_l1 = seq_dot(v.h, p, n) - mu * constr_l1;
        const bool fmode = filter_mode();
        if constexpr (HOOKS) { if (fmode && filter_is_acceptable(cost_1, constr_l1)) filter_add(cost_1, constr_l1); }
        double alpha = 1.0;
        cb_valid = false;
        for (int i = 1; i < ss.line_search_max_iter; ++i) {
            for (int j = ln; j < n; j += WAVE) { double t = alpha * p[j]; t += v.x[j]; v.xs[j] = t; }
            wsync();
            const double cost_step = ocp.cost(v.xs);
            cost_log = cost_step;
            const double constr_step = constraints_violation(v.xs);
            if constexpr (HOOKS) {
                if (fmode) {
                    if (filter_is_acceptable(cost_step, constr_step)) { filter_add(cost_step, constr_step); cb_valid = true; return alpha; }
                    alpha = ss.tau * alpha;
                    continue;
                }
            }
            const double phi_step = cost_step + mu * constr_step;
            if (__builtin_amdgcn_readfirstlane((int)(phi_step <= (phi_l1 + alpha * ss.eta * Dp_phi_l1)))) { cb_valid = true; return alpha; }
            alpha = ss.tau * alpha;
        }
        return alpha;
    }

    // The same line search with the trial points evaluated SIDE BY SIDE: a collocation grid has only NN nodes, so the
    // 64 lanes hold G = 64/NN candidate points at once — the current iterate (alpha = 0) and the next G-1 backtracking
    // trials alpha = 1, tau, tau^2, ... Each candidate goes through exactly the arithmetic of cost() / equalities() /
    // constraints_violation_impl; the acceptance test then walks the candidates in the reference's order, so the chosen
    // alpha and the logged cost are those of the sequential loop. The accepted candidate's constraint values are kept
    // for the termination test (x + alpha*p is the same floating-point vector).
    __device__ __forceinline__ double step_size_selection() {
        const int NNo = ocp.dm.NN;
        const int G = WAVE / NNo;
        if (G < 2 || !ls_side_by_side || ss.rho < 0.0) return step_size_selection_serial();   // (sqp rho is unused by the reference; negative = debug switch)
        constexpr int NX = Model::NX, NU = Model::NU, NP = Model::NP, NG = Model::NG;
        const double* p = qw.x;
        const int ln = lane_id();
        const int n = n_ct(), m = m_ct(), me = me_ct();
        const int P = ocp.P, S = ocp.S, VARX = ocp.dm.VARX, VARU = ocp.dm.VARU;
        double* cand_c = lsbuf;                    // [G][m]
        double* cand_L = cand_c + G * m;           // [G][NN]
        double* cand_viol = cand_L + G * NNo;      // [G]
        double* cand_cost = cand_viol + G;         // [G]
        double* cand_alpha = cand_cost + G;        // [G]
        const long long p0_ = now();
        const double mu = lds_inf_norm(v.lam_k, m + n);
        const double gp = seq_dot(v.h, p, n);
        acc(21, now() - p0_);
        double phi_l1 = 0.0, Dp_phi_l1 = 0.0;
        double alpha = 1.0;      // alpha of the next trial to be evaluated
        int trial = 1;           // index i of that trial in the reference loop (1 .. ls_max-1)
        bool first = true;
        cb_valid = false;
        while (true) {
            const int base = first ? 1 : 0;                            // candidate 0 of the first pass is the current iterate
            int ntr = ss.line_search_max_iter - trial;                 // trials still allowed
            if (ntr > G - base) ntr = G - base;
            if (ntr < 0) ntr = 0;
            const int ncand = base + ntr;
            {   // alpha of every candidate of this pass (the reference's running product alpha = tau * alpha)
                double al = alpha;
                for (int g = 0; g < ncand; ++g) {
                    if (first && g == 0) { if (ln == 0) cand_alpha[0] = 0.0; continue; }
                    if (ln == 0) cand_alpha[g] = al;
                    al = ss.tau * al;
                }
            }
            wsync();
            const long long e0 = now();
            const int g = ln / NNo, k = ln - g * NNo;
            if (g < ncand) {
                const bool is_base = first && g == 0;
                const double ag = cand_alpha[g];
                auto xat = [&](int idx) -> double { const double xv_ = v.x[idx]; double t = ag * p[idx]; t += xv_; return is_base ? xv_ : t; };
                double xk[NX > 0 ? NX : 1], uk[NU > 0 ? NU : 1], pk[NP > 0 ? NP : 1], f[NX > 0 ? NX : 1];
                for (int q = 0; q < NX; ++q) { xk[q] = xat(k * NX + q); f[q] = 0.0; }
                for (int q = 0; q < NU; ++q) uk[q] = xat(VARX + k * NU + q);
                for (int q = 0; q < NP; ++q) pk[q] = xat(VARX + VARU + q);
                const double tk = ocp.s.tn[k];
                ocp.model.template dynamics_impl<Value>(as_cvalues(xk), as_cvalues(uk), as_cvalues(pk), cref<double>(ocp.d), Value(tk), as_values(f));
                int seg, row; ocp.seg_row(k, seg, row);
                {   // D-row times the segment's states: loads of 4 nodes at a time (independent), then the ordered adds
                    double acc[NX > 0 ? NX : 1];
                    for (int q = 0; q < NX; ++q) acc[q] = 0.0;
                    for (int j0 = 0; j0 <= P; j0 += 4) {
                        double dv[4], xs[4][NX > 0 ? NX : 1];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int jj = (j0 + j <= P) ? j0 + j : 0;
                            dv[j] = ocp.s.D[row + jj * (P + 1)];
#pragma unroll
                            for (int q = 0; q < NX; ++q) xs[j][q] = xat((seg * P + jj) * NX + q);
                        }
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (j0 + j <= P) {
#pragma unroll
                                for (int q = 0; q < NX; ++q) acc[q] += dv[j] * xs[j][q];
                            }
                    }
#pragma unroll
                    for (int q = 0; q < NX; ++q) {
                        double cv = acc[q];
                        cv -= ocp.ts * f[q];
                        cand_c[g * m + k * NX + q] = cv;
                    }
                }
                if (NG > 0) {
                    double gg[NG > 0 ? NG : 1];
                    for (int q = 0; q < NG; ++q) gg[q] = 0.0;
                    ocp.model.template inequality_constraints_impl<Value>(as_cvalues(xk), as_cvalues(uk), as_cvalues(pk), cref<double>(ocp.d), tk, as_values(gg));
                    for (int q = 0; q < NG; ++q) cand_c[g * m + me + k * NG + q] = gg[q];
                }
                Value L(0.0);
                ocp.model.template lagrange_term_impl<Value>(as_cvalues(xk), as_cvalues(uk), as_cvalues(pk), cref<double>(ocp.d), tk, L);
                cand_L[g * NNo + k] = L.v;
            }
            wsync();
            const long long e1 = now();
            if (ln < ncand) {   // one lane per candidate: the scalar sums, in the reference's association order
                const int gc = ln;
                const bool is_base = first && gc == 0;
                const double ag = cand_alpha[gc];
                auto xat = [&](int idx) -> double { const double xv_ = v.x[idx]; double t = ag * p[idx]; t += xv_; return is_base ? xv_ : t; };
                // trip counts are compile-time constants on the register path: the loads are then issued in bulk and only
                // the (order-preserving) add chains remain serial; the lower / upper sums run as two interleaved chains
                constexpr int NNODES_CT = (NN > 0) ? MM / (NX + NG) : 0;
                const int n_ = (NN > 0) ? NN : n, m_ = (NN > 0) ? MM : m, me_ = (NN > 0) ? NX * NNODES_CT : me, mi_ = (NN > 0) ? NG * NNODES_CT : mi;
                double cl1 = DBL_EPS, sacc = 0.0;
                constexpr int CH = 8;   // loads in chunks of CH independent LDS reads, then the serial adds
                for (int i0 = 0; i0 < me_; i0 += CH) {
                    double cv[CH];
#pragma unroll
                    for (int i = 0; i < CH; ++i) cv[i] = cand_c[gc * m_ + ((i0 + i < me_) ? i0 + i : 0)];
#pragma unroll
                    for (int i = 0; i < CH; ++i) if (i0 + i < me_) sacc += fabs(cv[i]);
                }
                cl1 += sacc;
                double slo = 0.0, shi = 0.0;
                for (int i0 = 0; i0 < mi_; i0 += CH) {
                    double cv[CH], lb[CH], ub[CH];
#pragma unroll
                    for (int i = 0; i < CH; ++i) { const int ii = (i0 + i < mi_) ? i0 + i : 0; cv[i] = cand_c[gc * m_ + me_ + ii]; lb[i] = v.lbg[ii]; ub[i] = v.ubg[ii]; }
#pragma unroll
                    for (int i = 0; i < CH; ++i) if (i0 + i < mi_) { slo += fmax(lb[i] - cv[i], 0.0); shi += fmax(cv[i] - ub[i], 0.0); }
                }
                cl1 += slo; cl1 += shi;
                slo = 0.0; shi = 0.0;
                for (int i0 = 0; i0 < n_; i0 += CH) {
                    double pv[CH], xv[CH], lb[CH], ub[CH];
#pragma unroll
                    for (int i = 0; i < CH; ++i) { const int ii = (i0 + i < n_) ? i0 + i : 0; pv[i] = p[ii]; xv[i] = v.x[ii]; lb[i] = v.lbx[ii]; ub[i] = v.ubx[ii]; }
#pragma unroll
                    for (int i = 0; i < CH; ++i) if (i0 + i < n_) {
                        double t = ag * pv[i]; t += xv[i];
                        const double xi = is_base ? xv[i] : t;
                        slo += fmax(lb[i] - xi, 0.0); shi += fmax(xi - ub[i], 0.0);
                    }
                }
                cl1 += slo; cl1 += shi;
                cand_viol[gc] = cl1;
                double c = 0.0;
                for (int sg = 0; sg < S; ++sg)
                    for (int k0 = 0; k0 <= P; k0 += 4) {
                        double wv[4], Lv[4];
#pragma unroll
                        for (int kk = 0; kk < 4; ++kk) { const int kc = (k0 + kk <= P) ? k0 + kk : 0; wv[kk] = ocp.s.w[kc]; Lv[kk] = cand_L[gc * NNo + sg * P + kc]; }
#pragma unroll
                        for (int kk = 0; kk < 4; ++kk) if (k0 + kk <= P) c += ocp.ts * wv[kk] * Lv[kk];
                    }
                double x0[NX > 0 ? NX : 1], u0[NU > 0 ? NU : 1], p0[NP > 0 ? NP : 1];
                for (int q = 0; q < NX; ++q) x0[q] = xat(q);
                for (int q = 0; q < NU; ++q) u0[q] = xat(VARX + q);
                for (int q = 0; q < NP; ++q) p0[q] = xat(VARX + VARU + q);
                Value M(0.0);
                ocp.model.template mayer_term_impl<Value>(as_cvalues(x0), as_cvalues(u0), as_cvalues(p0), cref<double>(ocp.d), ocp.s.tn[0], M);
                c += M.v;
                cand_cost[gc] = c;
            }
            wsync();
            const long long e2 = now();
            acc(8, e1 - e0); acc(9, e2 - e1);
            const bool fmode = filter_mode();
            if (first) {
                const double constr_l1 = cand_viol[0];
                phi_l1 = cand_cost[0] + mu * constr_l1;
                Dp_phi_l1 = gp - mu * constr_l1;
                if constexpr (HOOKS) { if (fmode && filter_is_acceptable(cand_cost[0], constr_l1)) filter_add(cand_cost[0], constr_l1); }
            }
            int accepted = -1;
            for (int gc = base; gc < ncand; ++gc) {   // the reference's sequential acceptance order
                const double ag = cand_alpha[gc];
                const double cost_step = cand_cost[gc];
                cost_log = cost_step;
                bool ok;
                if constexpr (HOOKS) {
                    if (fmode) { ok = filter_is_acceptable(cost_step, cand_viol[gc]); if (ok) filter_add(cost_step, cand_viol[gc]); }
                    else ok = __builtin_amdgcn_readfirstlane((int)((cost_step + mu * cand_viol[gc]) <= (phi_l1 + ag * ss.eta * Dp_phi_l1))) != 0;
                } else {
                    const double phi_step = cost_step + mu * cand_viol[gc];
                    ok = __builtin_amdgcn_readfirstlane((int)(phi_step <= (phi_l1 + ag * ss.eta * Dp_phi_l1))) != 0;
                }
                if (ok) { accepted = gc; alpha = ag; break; }
                alpha = ss.tau * ag;
                ++trial;
            }
            if (accepted >= 0) {
                for (int i = ln; i < m; i += WAVE) v.cb[i] = cand_c[accepted * m + i];   // constraint values at x + alpha*p for the termination test
                cb_valid = true;
                wsync();
                acc(22, now() - e2);
                return alpha;
            }
            first = false;
            if (trial >= ss.line_search_max_iter) return alpha;
        }
    }

    // lag_grad = J^T lam[0:m] + cost_grad + lam_box  (continuous_ocp.hpp:2112-2114)
    __device__ __forceinline__ void lagrangian_gradient(double* out) {
        if constexpr (SCH) {   // J' lam from the per-node blocks and the differentiation matrix (pmpc_jview.hpp): the non-zero products of the dense chain, rows ascending
            const JV jv = jview();
#pragma unroll
            for (int e = 0; e < (NN + WAVE - 1) / WAVE; ++e) {
                const int col = lane_id() + WAVE * e;
                const int j = col < NN ? col : 0;
                double a = jv.coldot(j, v.lam);
                a += v.h[j];
                a += v.lam[MM + j];
                if (col < NN) out[j] = a;
            }
            wsync();
            return;
        }
        if constexpr (REG2) {   // (one KKT row per lane: the dense column loads below measured faster there — config A / D +6 % with the sparse form)
            // J' lam from the per-node blocks and the differentiation matrix in LDS: the non-zero products of the dense chain below in the
            // same ascending-row order (pmpc_jview.hpp). A non-finite multiplier takes the dense loops (0 * inf = NaN on the structural zeros).
            bool fin = true;
            for (int i = lane_id(); i < MM; i += WAVE) fin = fin && ((v.lam[i] - v.lam[i]) == 0.0);
            if (__builtin_amdgcn_ballot_w64(!fin) == 0) {
                const JV jv = jview();
#pragma unroll
                for (int e = 0; e < (NN + WAVE - 1) / WAVE; ++e) {
                    const int col = lane_id() + WAVE * e;
                    const int j = col < NN ? col : 0;
                    double a = jv.coldot(j, v.lam);
                    a += v.h[j];
                    a += v.lam[MM + j];
                    if (col < NN) out[j] = a;
                }
                wsync();
                return;
            }
        }
        if constexpr (REG1) {   // compile-time sizes: one batch of independent loads (column j of J), then the chain
            const int j = lane_id() < NN ? lane_id() : 0;
            const unsigned jo = (unsigned)j * (NN + MM) + opaque_zero();
            double col[MM];
#pragma unroll
            for (int i = 0; i < MM; ++i) col[i] = Aw[jo + (unsigned)i];
            double a = 0.0;
#pragma unroll
            for (int i = 0; i < MM; ++i) a += col[i] * v.lam[i];
            a += v.h[j];
            a += v.lam[MM + j];
            if (lane_id() < NN) out[j] = a;
            wsync();
            return;
        }
        if constexpr (REG2) {   // compile-time sizes, columns lane and lane + 64: the loads of a column in batches, then the same add chain
            constexpr bool FEW = NN > WAVE && NN - WAVE <= 4 && MM <= WAVE;   // a few columns in the second slot (config B: 64 and 65): lane i loads J(i, column)
            if constexpr (FEW) {                                              // and forms the product with lam_i, the products are added in ascending i on every lane
#pragma unroll                                                                // (v_readlane) — instead of MM loads per lane for two live lanes
                for (int t = 0; t < NN - WAVE; ++t) {
                    const int i = lane_id() < MM ? lane_id() : 0;
                    const unsigned o = (unsigned)(WAVE + t) * (NN + MM) + (unsigned)i + opaque_zero();
                    const double prod = Aw[o] * v.lam[i];
                    double a = 0.0;
#pragma unroll
                    for (int k = 0; k < MM; ++k) a += bcast_lane(prod, k);
                    a += v.h[WAVE + t];
                    a += v.lam[MM + WAVE + t];
                    if (lane_id() == t) out[WAVE + t] = a;
                }
            }
#pragma unroll
            for (int e = 0; e < ((NN > WAVE && !FEW) ? 2 : 1); ++e) {
                const int col = lane_id() + 64 * e;
                const int j = col < NN ? col : 0;
                const unsigned jo = (unsigned)j * (NN + MM) + opaque_zero();
                double a = 0.0;
#pragma unroll
                for (int i0 = 0; i0 < MM; i0 += 16) {
                    double colv[16];
#pragma unroll
                    for (int i = 0; i < 16; ++i) colv[i] = Aw[jo + (unsigned)((i0 + i < MM) ? i0 + i : 0)];
#pragma unroll
                    for (int i = 0; i < 16; ++i) if (i0 + i < MM) a += colv[i] * v.lam[i0 + i];
                }
                a += v.h[j];
                a += v.lam[MM + j];
                if (col < NN) out[j] = a;
            }
            wsync();
            return;
        }
        for (int j = lane_id(); j < n; j += WAVE) {
            double a = seq_dot_strided(Aw, 1, (size_t)ldw, j, m, v.lam);   // column j of J against lam: one add chain, rows ascending, eight loads in flight
            a += v.h[j];
            a += v.lam[m + j];
            out[j] = a;
        }
        wsync();
    }

    // Eigenvalue mirroring, tests/solvers/sqp/sqp_test_autodiff.cpp:29-45 (the regulariser the reference's SQP tests plug into
    // hessian_regularisation_dense_impl): H = V diag(w) V^T with every eigenvalue w <= 0 replaced by -w + 0.1, when the smallest one is not
    // positive. The reference calls Eigen::EigenSolver; the CPU restatement — and this routine, operation for operation — uses the cyclic
    // Jacobi method: rotations (p, q) in row-major order, t = sign(theta) / (|theta| + sqrt(theta^2 + 1)), the column pair, then the row pair,
    // then the vector pair updated as  c a - s b,  s a + c b  (products and one add, no fma), sweeps until the sum of squares of the strict
    // lower triangle drops below 1e-300 (at most 100). A (n x n) and V (n x n) live in LDS (`eig`, 2 n^2 doubles, allocated by the
    // launcher only when regularisation = 1); every lane evaluates the wave-uniform scalars redundantly, the O(n) updates run one entry per lane.
    double* eig = nullptr;
    // Eigenvalue mirroring (sqp_test_autodiff.cpp:29-45): H = V D V' with the non-positive eigenvalues w replaced by -w + 0.1. The reference calls Eigen::EigenSolver; here a Jacobi
    // iteration in LDS (A = eig[0, n^2), V behind it) — since late round 6 in the ROUND-ROBIN order: the n / 2 disjoint pairs of a round rotate together (their angles from the matrix
    // at the start of the round; then every pair mixes its two COLUMNS of A and V over all rows, then every pair mixes its two ROWS of A over all columns), n - 1 rounds per sweep,
    // so that all 64 lanes work, and the annihilated entry is written as an exact zero — the pair-by-pair cyclic order of rounds 2 .. 6 left its rounding residue there, never met its
    // stopping rule and ran all 100 sweeps: 7 (n = 35) .. 33 ms (n = 80) per SQP iteration. Restated by the test suite's CPU checker (jacobi_eig; it keeps the pair-by-pair iteration for n <= 8, the sizes of the reference's own NLP tests — no OCP grid is that small)
    // (each entry sees the same operations in the same order: the pairs of a round touch disjoint columns in the first phase and disjoint rows in the second).
    __device__ __forceinline__ void regularise_eig_mirror() {
        const int ln = lane_id();
        const int n = n_ct();
        double* A = eig; double* V = eig + (size_t)n * n;
        for (int e = ln; e < n * n; e += WAVE) { const int j = e / n, i = e - j * n; A[e] = Hw[(size_t)j * ldw + i]; V[e] = (i == j) ? 1.0 : 0.0; }
        wsync();
        const int np = n + (n & 1), m2 = np / 2, nr = np - 1;   // round-robin tournament over np players (np - 1 = n: a bye when n is odd)
        double* cs = v.t1;                                       // c of pair i at [i], s at [m2 + i] (m2 <= n / 2 + 1 entries each)
        auto pair_of = [&](int r, int i, int& p, int& q) {
            const int a = (i == 0) ? np - 1 : (r + i) % nr, b = (i == 0) ? r : (r - i + nr) % nr;
            p = a < b ? a : b; q = a < b ? b : a;
        };
        for (int sweep = 0; sweep < 100; ++sweep) {
            double amax = 0.0;
            for (int e = ln; e < n * n; e += WAVE) { const int j = e / n, i = e - j * n; if (i > j) amax = fmax(amax, fabs(A[e])); }
            amax = wave_max(amax);
            if (__builtin_amdgcn_readfirstlane((int)(amax * amax < 1e-300))) break;
            for (int r = 0; r < nr; ++r) {
                for (int i = ln; i < m2; i += WAVE) {
                    int p, q; pair_of(r, i, p, q);
                    double c = 1.0, sn = 0.0;
                    if (q < n) {
                        const double apq = A[p + q * n];
                        if (!(fabs(apq) < 1e-300)) {
                            const double theta = (A[q + q * n] - A[p + p * n]) / (2 * apq);
                            const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + ::sqrt(theta * theta + 1));
                            c = 1 / ::sqrt(t * t + 1); sn = t * c;
                        }
                    }
                    cs[i] = c; cs[m2 + i] = sn;
                }
                wsync();
                for (int it = ln; it < m2 * n; it += WAVE) {   // columns p, q of A and V, every row k
                    const int i = it / n, k = it - i * n;
                    int p, q; pair_of(r, i, p, q);
                    const double c = cs[i], sn = cs[m2 + i];
                    if (q < n && sn != 0.0) {
                        const double akp = A[k + p * n], akq = A[k + q * n]; A[k + p * n] = c * akp - sn * akq; A[k + q * n] = sn * akp + c * akq;
                        const double vkp = V[k + p * n], vkq = V[k + q * n]; V[k + p * n] = c * vkp - sn * vkq; V[k + q * n] = sn * vkp + c * vkq;
                    }
                }
                wsync();
                for (int it = ln; it < m2 * n; it += WAVE) {   // rows p, q of A, every column k
                    const int i = it / n, k = it - i * n;
                    int p, q; pair_of(r, i, p, q);
                    const double c = cs[i], sn = cs[m2 + i];
                    if (q < n && sn != 0.0) { const double apk = A[p + k * n], aqk = A[q + k * n]; A[p + k * n] = c * apk - sn * aqk; A[q + k * n] = sn * apk + c * aqk; }
                }
                wsync();
                // the rotated pair's off-diagonal entry is zero by construction; what the two mixes leave there is the rounding of the DIAGONAL entries (eps |a_pp|), which
                // never decays — written as the exact zero, the off-diagonal mass falls quadratically to underflow (~10 sweeps) instead of stalling at eps |A| for all 100
                for (int i = ln; i < m2; i += WAVE) {
                    int p, q; pair_of(r, i, p, q);
                    if (q < n && cs[m2 + i] != 0.0) { A[p + q * n] = 0.0; A[q + p * n] = 0.0; }
                }
                wsync();
            }
        }
        double mn = A[0];
        for (int i = 1; i < n; ++i) mn = fmin(mn, A[i + i * n]);
        if (__builtin_amdgcn_readfirstlane((int)(mn <= 0))) {
            wsync();
            for (int i = ln; i < n; i += WAVE) { const double w = A[i + i * n]; A[i + i * n] = (w <= 0) ? (-1 * w + 0.1) : w; }   // (the diagonal of A now holds the mirrored spectrum)
            wsync();
            for (int e = ln; e < n * n; e += WAVE) {
                const int j = e / n, i = e - j * n;
                double a = 0.0;
                for (int k = 0; k < n; ++k) a += (V[i + k * n] * A[k + k * n]) * V[j + k * n];
                Hw[(size_t)j * ldw + i] = a;
            }
            wfence();
        }
        wsync();
    }

    // Gershgorin shift, dense_sparse_compare.cpp:109-122
    __device__ __forceinline__ void regularise_gershgorin() {
        if constexpr (SCH) {   // column i of H = the node's block column: |entries| added rows ascending (x rows, then u rows), as the dense loop adds them
            constexpr int NB = Model::NX + Model::NU, N0 = NN - Model::NP;
            for (int i = lane_id(); i < NN; i += WAVE) {
                if (Model::NP > 0 && i >= N0) {   // the parameter's column: the border column rows ascending, the corner last
                    const double aii = hbrd[N0];
                    double ri = 0.0;
                    for (int r = 0; r < N0; ++r) ri += fabs(hbrd[NN + r]);
                    ri += fabs(aii);
                    ri -= fabs(aii);
                    if (aii - ri <= 0) hbrd[N0] = aii + ((ri - aii) + 0.01);
                    continue;
                }
                const bool isx = i < Model::NX * NNODES_CT_;
                const int k = isx ? i / Model::NX : (i - Model::NX * NNODES_CT_) / Model::NU;
                const int c = isx ? i - k * Model::NX : Model::NX + (i - Model::NX * NNODES_CT_) - k * Model::NU;
                const double* col = hblk + k * NB * NB + c * NB;
                const double aii = col[c];
                double ri = 0.0;
#pragma unroll
                for (int r = 0; r < NB; ++r) ri += fabs(col[r]);
                if constexpr (Model::NP > 0) ri += fabs(hbrd[i]);   // (row p of the column)
                ri -= fabs(aii);
                if (aii - ri <= 0) hblk[k * NB * NB + c * NB + c] = aii + ((ri - aii) + 0.01);
            }
            wsync();
            return;
        }
        for (int i = lane_id(); i < n; i += WAVE) {
            const double aii = Hw[(size_t)i * ldw + i];
            double ri = 0.0;
            for (int k = 0; k < n; ++k) ri += fabs(Hw[(size_t)i * ldw + k]);
            ri -= fabs(aii);
            if (aii - ri <= 0) Hw[(size_t)i * ldw + i] = aii + ((ri - aii) + 0.01);
        }
        wfence();
        wsync();
    }

    // linearisation_dense_impl :310-318 (exact = true) and update_linearisation_dense_impl :490-504 (exact = false) share ONE
    // inlined copy of the first-order stage (as a called function it kept the whole Ocp object in private memory and
    // spilled the live registers around every call).
    //   exact:  f, df, L, dL -> c, J, cost gradient; second-order stage -> H; lag_grad
    //   update: f, df, L, dL -> c, J (per-node blocks only), cost gradient; new lag_grad; damped BFGS on H
    __device__ __forceinline__ void linearise(bool exact, bool structure) {
        const long long l0 = now();
        if constexpr (BIG_NW > 1) {   // (node, direction) pairs dealt over the team (big_helper_loop)
            BigMail<JViewRT<Model>>* mail = (BigMail<JViewRT<Model>>*)big_mail;
            if (lane_id() == 0) { mail->op = BIG_OP_STAGE1; mail->var = v.x; mail->n = n; mail->m = m; }
            __syncthreads();
            ocp.template stage_first_order_part<BIG_NW>(v.x, 0);
            __syncthreads();
        } else
        ocp.stage_first_order(v.x);
        const long long l1 = now();
        acc(10, l1 - l0);
        if (exact) {
            if constexpr (BIG_NW > 1 && (int)Dm::NDER > Ocp<Model>::WIDE_MAX_NDER) {   // the team's helpers take three quarters of the entries (big_helper_loop)
                BigMail<JViewRT<Model>>* mail = (BigMail<JViewRT<Model>>*)big_mail;
                if (lane_id() == 0) { mail->op = BIG_OP_STAGE2; mail->var = v.x; mail->lam = v.lam; }
                __syncthreads();
                ocp.template stage_second_order_entry_part<BIG_NW>(v.x, v.lam, 0);
                __syncthreads();
            } else
            ocp.stage_second_order(v.x, v.lam);
            const long long l2 = now();
            if constexpr (SCH) ocp.template assemble_first_order<false, false>(v.al, nullptr, v.h, 0, false);
            else ocp.template assemble_first_order<false>(v.al, Aw, v.h, ldw, structure);
            const long long l3 = now();
            if constexpr (SCH) { if constexpr (Model::NP > 0) ocp.assemble_hessian_arrow(hblk, hbrd); else ocp.assemble_hessian_blocks(hblk); }
            else ocp.assemble_hessian(Hw, ldw);
            const long long l4 = now();
            lagrangian_gradient(v.lg);
            acc(11, l2 - l1); acc(12, l3 - l2); acc(13, l4 - l3); acc(14, now() - l4);
            if (ss.regularisation == 2) regularise_gershgorin();
            if constexpr (HOOKS) { if (ss.regularisation == 1) regularise_eig_mirror(); }   // (LDS / HBM-resident kernels and, since round 6, the hook builds of the register kernels: the launcher routes this policy there)
        } else {
            // J's zeros and D entries are already in place — unless the Ruiz preconditioner scaled and unscaled the workspace around the
            // last QP (sqp_base.hpp:605-609): the round trip leaves rounding noise on every entry, and the reference rebuilds J from
            // scratch at each linearisation
            bool rebuild = false;
            if constexpr (RUIZ_COMPILED) rebuild = __builtin_amdgcn_readfirstlane(ss.preconditioner) == 1;
            if constexpr (SCH) ocp.template assemble_first_order<false, false>(v.al, nullptr, v.h, 0, false);
            else ocp.template assemble_first_order<false>(v.al, Aw, v.h, ldw, rebuild);
            const long long l3 = now();
            lagrangian_gradient(v.lgn);
            acc(12, l3 - l1); acc(14, now() - l3);
            const long long b0 = now();
            if constexpr (SCH) bfgs_update_block();
            else if constexpr (REG1 && HU == 1) bfgs_update_block();
            else if constexpr (REG1) { double brow[NN > 0 ? NN : 1]; bfgs_load_row(brow); bfgs_update_reg(brow); }
            else if constexpr (REG2) { if (__builtin_amdgcn_readfirstlane(ss.hessian_update) == 1) bfgs_update_block(); else bfgs_update_reg2(); }
            else { if (__builtin_amdgcn_readfirstlane(ss.hessian_update) == 1) bfgs_update_block(); else bfgs_update(); }   // (the launcher routes hessian_update = 1 to these kernels)
            acc(5, now() - b0);
            for (int i = lane_id(); i < n; i += WAVE) v.lg[i] = v.lgn[i];
            wsync();
        }
    }

    // BFGS_update, bfgs.hpp:23-52 ; s = v.step, y = lgn - lg
    // register-row variant for compile-time n: lane i owns row i of B; ONE batch of loads, ONE batch of stores
    // brow: row i of B (loading it earlier, across the first-order staging, costs more in register pressure than the L2
    // round trip it hides — measured)
    // E: row slot — lane i owns row i + 64 E (two-rows-per-lane kernels: E = 1 serves the rows from 64 on)
    template <int E = 0>
    __device__ __forceinline__ void bfgs_load_row(double (&brow)[NN > 0 ? NN : 1]) {
        const int row = lane_id() + 64 * E;
        const unsigned i = (row < NN ? row : 0) + opaque_zero();
#pragma unroll
        for (int j = 0; j < NN; ++j) brow[j] = (Hw + (size_t)(j * (NN + MM)))[i];   // (uniform column base + ONE per-lane offset: scalar-base addressing, no address register pair per load)
    }
    // B s and y for the rows of slot E (into v.t1 / v.t3); brow stays in registers
    template <int E>
    __device__ __forceinline__ void bfgs_row_products(const double (&brow)[NN > 0 ? NN : 1]) {
        const int row = lane_id() + 64 * E;
        const int i = row < NN ? row : 0;
        double a = 0.0;
#pragma unroll
        for (int j = 0; j < NN; ++j) a += brow[j] * v.step[j];
        if (row < NN) { v.t1[i] = a; v.t3[i] = v.lgn[i] - v.lg[i]; }
    }
    // B(row, :) += -(Bs_row Bs')/sBs + (r_row r')/sr for the rows of slot E, then the row goes back to the workspace
    template <int E>
    __device__ __forceinline__ void bfgs_row_rank2(double (&brow)[NN > 0 ? NN : 1], const UniformDiv& by_sBs, const UniformDiv& by_sr) {
        const int row = lane_id() + 64 * E;
        const int i = row < NN ? row : 0;
        const double* Bs = v.t1; const double* r = v.t2;
        const double Bsi = Bs[i], ri = r[i];
#pragma unroll
        for (int j0 = 0; j0 < NN; j0 += 8) {
            double bsj[8], rj[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) { const int jj = (j0 + j < NN) ? j0 + j : 0; bsj[j] = Bs[jj]; rj[j] = r[jj]; }
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (j0 + j < NN) {
                    double b = brow[j0 + j];
                    b += by_sBs(-Bsi * bsj[j]);
                    b += by_sr(ri * rj[j]);
                    asm volatile("" : "+v"(b));   // (pins the chunk here — see bfgs_update_reg)
                    brow[j0 + j] = b;
                }
            asm volatile("" ::: "memory");
            sched_fence();
        }
        if (row < NN) {
            const unsigned io = (unsigned)i + opaque_zero();
#pragma unroll
            for (int j = 0; j < NN; ++j) (Hw + (size_t)(j * (NN + MM)))[io] = brow[j];
        }
    }
    // BFGS_update for the two-rows-per-lane kernels (compile-time n, up to 128 rows of B): the same operations on every entry as
    // bfgs_update_reg / bfgs_update (one add chain per row for B s, sequential scalar products, quotients by the wave-uniform divisors)
    __device__ __forceinline__ void bfgs_update_reg2() {
        const int ln = lane_id();
        double* Bs = v.t1; double* r = v.t2; double* y = v.t3;
        double brow[NN > 0 ? NN : 1];
        if constexpr (NN > WAVE) { bfgs_load_row<1>(brow); bfgs_row_products<1>(brow); }
        bfgs_load_row<0>(brow); bfgs_row_products<0>(brow);
        wsync();
        double sBs, sy;
        seq_dot_pair(v.step, Bs, y, sBs, sy);
        double sr;
        if (sy < 0.2 * sBs) {
            const double theta = 0.8 * sBs / (sBs - sy);
            for (int i = ln; i < NN; i += WAVE) r[i] = theta * y[i] + (1 - theta) * Bs[i];
            sr = theta * sy + (1 - theta) * sBs;
        } else {
            for (int i = ln; i < NN; i += WAVE) r[i] = y[i];
            sr = sy;
        }
        wsync();
        if (__builtin_amdgcn_readfirstlane((int)(sr < DBL_EPS))) return;
        const UniformDiv by_sBs(sBs), by_sr(sr);
        if (!(by_sBs.ok() && by_sr.ok())) { rank2_update_mem(Bs, r, sBs, sr); return; }   // divisor outside the window of UniformDiv (rare)
        bfgs_row_rank2<0>(brow, by_sBs, by_sr);
        if constexpr (NN > WAVE) { bfgs_load_row<1>(brow); bfgs_row_rank2<1>(brow, by_sBs, by_sr); }
        wfence();
        wsync();
    }
    // s'a and s'b for compile-time n, each the ascending chain of seq_dot, eight entries per scheduling fence: fully unrolled and unfenced, the
    // scheduler hoists all 3 n LDS reads above the two chains — 210 registers beside the row of B, whose head then goes to scratch right behind
    // its loads (one L2 round trip each: the update took 17.3 k instead of 9 k cycles per iteration)
    __device__ __forceinline__ void seq_dot_pair(const double* s, const double* a, const double* b, double& sa, double& sb) {
        sa = 0.0; sb = 0.0;
#pragma unroll
        for (int j0 = 0; j0 < NN; j0 += 8) {
            double ts[8], ta[8], tb[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) { const int jj = (j0 + j < NN) ? j0 + j : 0; ts[j] = s[jj]; ta[j] = a[jj]; tb[j] = b[jj]; }
#pragma unroll
            for (int j = 0; j < 8; ++j) if (j0 + j < NN) { sa += ts[j] * ta[j]; sb += ts[j] * tb[j]; }
            sched_fence();
        }
    }
    __device__ __forceinline__ void bfgs_update_reg(double (&brow)[NN > 0 ? NN : 1]) {
        const int ln = lane_id();
        const int i = ln < NN ? ln : 0;
        double* Bs = v.t1; double* r = v.t2; double* y = v.t3;
        {
            double a = 0.0;
#pragma unroll
            for (int j = 0; j < NN; ++j) a += brow[j] * v.step[j];
            if (ln < NN) { Bs[i] = a; y[i] = v.lgn[i] - v.lg[i]; }
        }
        wsync();
        double sBs, sy;
        seq_dot_pair(v.step, Bs, y, sBs, sy);
        double sr;
        if (sy < 0.2 * sBs) {
            const double theta = 0.8 * sBs / (sBs - sy);
            if (ln < NN) r[i] = theta * y[i] + (1 - theta) * Bs[i];
            sr = theta * sy + (1 - theta) * sBs;
        } else {
            if (ln < NN) r[i] = y[i];
            sr = sy;
        }
        wsync();
        if (__builtin_amdgcn_readfirstlane((int)(sr < DBL_EPS))) return;
        const double Bsi = Bs[i], ri = r[i];
        const UniformDiv by_sBs(sBs), by_sr(sr);   // the same quotients as "/ sBs", "/ sr", bit for bit
        if (by_sBs.ok() && by_sr.ok()) {
#pragma unroll
            for (int j0 = 0; j0 < NN; j0 += 8) {
                double bsj[8], rj[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) { const int jj = (j0 + j < NN) ? j0 + j : 0; bsj[j] = Bs[jj]; rj[j] = r[jj]; }
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (j0 + j < NN) {
                        double b = brow[j0 + j];
                        b += by_sBs(-Bsi * bsj[j]);
                        b += by_sr(ri * rj[j]);
                        asm volatile("" : "+v"(b));   // (pins the chunk here: the optimiser otherwise sinks the arithmetic into the store block below and leaves all 2 n LDS reads in front of it, beside the row of B — 256 registers and scratch)
                        brow[j0 + j] = b;
                    }
                asm volatile("" ::: "memory");
                sched_fence();
            }
        } else {   // divisor outside the window of UniformDiv: generic divisions, in place in the workspace (rare)
            rank2_update_mem(Bs, r, sBs, sr);
            return;
        }
        if (ln < NN) {
            const unsigned io = (unsigned)i + opaque_zero();
#pragma unroll
            for (int j = 0; j < NN; ++j) (Hw + (size_t)(j * (NN + MM)))[io] = brow[j];
        }
        wfence();
        wsync();
    }
    __device__ __forceinline__ void bfgs_update() {
        const int ln = lane_id();
        double* Bs = v.t1; double* r = v.t2; double* y = v.t3;
        if constexpr (BIG_NW > 1) {   // rows dealt over the team (big_helper_loop)
            BigMail<JViewRT<Model>>* mail = (BigMail<JViewRT<Model>>*)big_mail;
            if (ln == 0) { mail->op = BIG_OP_BFGS_BS; mail->n = n; mail->m = m; mail->Hw = Hw; mail->ldw = ldw; mail->pa[0] = v.step; mail->pa[1] = v.lgn; mail->pa[2] = v.lg; mail->pw[0] = Bs; mail->pw[1] = y; }
            __syncthreads();
            bfgs_rows_products<BIG_NW, MEMCH>(Hw, ldw, n, v.step, v.lgn, v.lg, Bs, y, 0);
            __syncthreads();
        } else {
        for (int i = ln; i < n; i += WAVE) {
            Bs[i] = seq_dot_strided<MEMCH>(Hw, (size_t)ldw, 1, i, n, v.step);   // row i of B times s: one add chain, columns ascending, eight loads in flight
            y[i] = v.lgn[i] - v.lg[i];
        }
        wsync();
        }
        const double sBs = seq_dot(v.step, Bs, n);
        const double sy = seq_dot(v.step, y, n);
        double sr;
        if (sy < 0.2 * sBs) {
            const double theta = 0.8 * sBs / (sBs - sy);
            for (int i = ln; i < n; i += WAVE) r[i] = theta * y[i] + (1 - theta) * Bs[i];
            sr = theta * sy + (1 - theta) * sBs;
        } else {
            for (int i = ln; i < n; i += WAVE) r[i] = y[i];
            sr = sy;
        }
        wsync();
        if (sr < DBL_EPS) return;
        if constexpr (BIG_NW > 1) {   // rows dealt over the team
            const UniformDiv da(sBs), db(sr);
            const bool fast = da.ok() && db.ok();
            BigMail<JViewRT<Model>>* mail = (BigMail<JViewRT<Model>>*)big_mail;
            if (ln == 0) { mail->op = BIG_OP_BFGS_R2; mail->n = n; mail->m = m; mail->Hw = Hw; mail->ldw = ldw; mail->pa[3] = Bs; mail->pa[4] = r; mail->f[0] = sBs; mail->f[1] = sr; mail->fast = fast ? 1 : 0; }
            __syncthreads();
            if (fast) bfgs_rows_rank2<BIG_NW, MEMCH, true>(Hw, ldw, n, Bs, r, sBs, sr, 0); else bfgs_rows_rank2<BIG_NW, MEMCH, false>(Hw, ldw, n, Bs, r, sBs, sr, 0);
            __syncthreads();
        } else
        rank2_update_mem(Bs, r, sBs, sr);
    }
    // Sparsity-preserving block BFGS: ContinuousOCP::hessian_update_impl<SPARSE>, continuous_ocp.hpp:2304-2431 (what the reference's
    // MPC tests plug into SQPBase::hessian_update_impl, mpc_wrapper_test.cpp:100-105), on the dense workspace: per node k only the
    // (x_k, u_k) diagonal block gets the damped rank-2 update, with the global scalars s'Bs, s'y, s'r; NP > 0 adds the parameter
    // border and corner. One lane per updated entry; coefficients in the reference's association order
    //   (-scaling_inv * v_i) * v_j  then  += (c_inv * w_i) * w_j   (w = y or the damped r);  hes_xu = hes_ux'.
    __device__ __forceinline__ void bfgs_update_block() {
        constexpr int NX = Model::NX, NU = Model::NU, NP = Model::NP, NB = NX + NU;
        const int ln = lane_id();
        const int VARX = ocp.dm.VARX, VARU = ocp.dm.VARU, NNo = ocp.dm.NN;
        double* vv = v.t1; double* r = v.t2; double* y = v.t3;
        if constexpr (SCH) {   // row i of B times s from the node's block: the non-zero products of the dense chain, columns ascending (x columns, then u columns)
            for (int i = ln; i < n; i += WAVE) {
                if (NP > 0 && i >= VARX + VARU) {   // the parameter's row: the border row, columns ascending, the corner last
                    vv[i] = seq_dot(hbrd, v.step, n);
                    y[i] = v.lgn[i] - v.lg[i];
                    continue;
                }
                const bool isx = i < VARX;
                const int k = isx ? i / NX : (i - VARX) / NU;
                const int c = isx ? i - k * NX : NX + (i - VARX) - k * NU;
                double hb[NB], sv[NB];
#pragma unroll
                for (int cc = 0; cc < NB; ++cc) hb[cc] = hblk[k * NB * NB + cc * NB + c];
#pragma unroll
                for (int cc = 0; cc < NX; ++cc) sv[cc] = v.step[k * NX + cc];
#pragma unroll
                for (int cc = 0; cc < NU; ++cc) sv[NX + cc] = v.step[VARX + k * NU + cc];
                double a = 0.0;
#pragma unroll
                for (int cc = 0; cc < NB; ++cc) a += hb[cc] * sv[cc];
                if constexpr (NP > 0) a += hbrd[n + i] * v.step[VARX + VARU];   // (border column: the last product of the row's chain)
                vv[i] = a;
                y[i] = v.lgn[i] - v.lg[i];
            }
        } else
        for (int i = ln; i < n; i += WAVE) {
            vv[i] = seq_dot_strided<MEMCH>(Hw, (size_t)ldw, 1, i, n, v.step);   // row i of B times s: one add chain, columns ascending, eight loads in flight
            y[i] = v.lgn[i] - v.lg[i];
        }
        wsync();
        const double scaling = seq_dot(v.step, vv, n);
        const double scaling_inv = 1.0 / scaling;
        const double sy = seq_dot(v.step, y, n);
        const double sy_inv = 1.0 / sy;
        const double* w = y; double c_inv = sy_inv;
        if (!(sy >= 0.2 * scaling)) {
            const double theta = 0.8 * scaling / (scaling - sy);
            for (int i = ln; i < n; i += WAVE) r[i] = theta * y[i] + (1 - theta) * vv[i];
            wsync();
            c_inv = 1.0 / seq_dot(v.step, r, n);
            w = r;
        }
        auto term = [&](int i, int j) { double t = (-scaling_inv * vv[i]) * vv[j]; t += (c_inv * w[i]) * w[j]; return t; };
        auto gidx = [&](int k, int b) { return b < NX ? k * NX + b : VARX + k * NU + (b - NX); };
        for (int e = ln; e < NNo * NB * NB; e += WAVE) {
            const int k = e / (NB * NB), rem = e - k * (NB * NB), bj = rem / NB, bi = rem - bj * NB;
            const int gi = gidx(k, bi), gj = gidx(k, bj);
            const double t = (bi < NX && bj >= NX) ? term(gj, gi) : term(gi, gj);   // the xu block is the transpose of the ux block
            if constexpr (SCH) hblk[e] += t;   // (e = k NB^2 + bj NB + bi: the block layout itself)
            else Hw[(size_t)gj * ldw + gi] += t;
        }
        if constexpr (NP > 0) {
            const int a = VARX + VARU;
            for (int e = ln; e < a * NP; e += WAVE) {   // border: column block and its transposed copy in the rows
                const int j = e / a, i = e - j * a;
                const double t = term(i, a + j);
                if constexpr (SCH) { hbrd[n + i] += t; hbrd[i] += t; }
                else {
                Hw[(size_t)(a + j) * ldw + i] += t;
                Hw[(size_t)i * ldw + (a + j)] += t;
                }
            }
            for (int e = ln; e < NP * NP; e += WAVE) {
                const int j = e / NP, i = e - j * NP;
                if constexpr (SCH) hbrd[a] += term(a, a);
                else Hw[(size_t)(a + j) * ldw + (a + i)] += term(a + i, a + j);
            }
        }
        wfence();
        wsync();
    }
    // B += -(Bs Bs^T)/sBs + (r r^T)/sr, element by element in the HBM workspace: lane i walks row i, MEMCH columns per batch of loads
    // (every entry sees the same two operations as in the reference's expression, entries are independent of each other).
    // FAST: the two quotients per entry through UniformDiv (5 operations each instead of the ~40 of the generic expansion, the same bits) —
    // callers pass it when both divisors lie in its window.
    template <bool FAST>
    __device__ __forceinline__ void rank2_update_rows(const double* Bs, const double* r, double sBs, double sr) {
        const int ln = lane_id();
        constexpr int CH = MEMCH;
        const UniformDiv by_sBs(sBs), by_sr(sr);
        for (int i = ln; i < n; i += WAVE) {
            const double Bsi = Bs[i], ri = r[i];
            double* __restrict__ row = Hw + i;
            for (int j0 = 0; j0 < n; j0 += CH) {
                double b[CH];
#pragma unroll
                for (int u = 0; u < CH; ++u) b[u] = row[(size_t)((j0 + u < n) ? j0 + u : n - 1) * ldw];
#pragma unroll
                for (int u = 0; u < CH; ++u)
                    if (j0 + u < n) {
                        double t = b[u];
                        if constexpr (FAST) { t += by_sBs(-Bsi * Bs[j0 + u]); t += by_sr(ri * r[j0 + u]); }
                        else { t += (-Bsi * Bs[j0 + u]) / sBs; t += (ri * r[j0 + u]) / sr; }
                        row[(size_t)(j0 + u) * ldw] = t;
                    }
            }
        }
        wfence();
        wsync();
    }
    __device__ __forceinline__ void rank2_update_mem(const double* Bs, const double* r, double sBs, double sr) {
        if constexpr (NN > 0) rank2_update_rows<false>(Bs, r, sBs, sr);   // register-resident kernels: the rare fallback of their own UniformDiv paths
        else {
            const UniformDiv a(sBs), b(sr);
            if (a.ok() && b.ok()) rank2_update_rows<true>(Bs, r, sBs, sr); else rank2_update_rows<false>(Bs, r, sBs, sr);
        }
    }

    // QP bounds :588-593
    __device__ __forceinline__ void form_qp_bounds() {
        const int ln = lane_id();
        const int n = n_ct(), m = m_ct(), me = me_ct();
        for (int i = ln; i < m; i += WAVE) {
            double a = -v.al[i], b = a;
            if (i >= me) { a += v.lbg[i - me]; b += v.ubg[i - me]; }
            v.al[i] = a; v.au[i] = b;
        }
        for (int i = ln; i < n; i += WAVE) { v.lx[i] = v.lbx[i] - v.x[i]; v.ux[i] = v.ubx[i] - v.x[i]; }
        wsync();
    }

    // one SQP iteration after (update_)linearisation: QP, line search, step, norms  (:588-632 / :652-683)
    __device__ __forceinline__ void qp_and_step() {
        const int ln = lane_id();
        const int n = n_ct(), m = m_ct();
        const long long q0 = now();
        form_qp_bounds();
        pmpc_qp_info qi;
        // m_preconditioner.compute(m_H, m_h, m_A, m_al, m_au, m_lx, m_ux), sqp_base.hpp:605 / :661 — in place in the workspace
        // (RUIZ_COMPILED: a guard from the time when the 16-direction stand-in kernel — then with 7.9 KB of private AD arrays per
        // lane — returned NaNs as soon as this never-taken branch was compiled into it, hipcc 7.2; with the entry-per-lane
        // second-order stage that kernel has no such arrays and the fault no longer reproduces, so the guard admits every model)
        bool ruiz = false;
        const RuizScratch rz{v.t1, v.t1 + n, v.t2, v.t2 + n};
        double rz_c = 1.0;
        if constexpr (RUIZ_COMPILED) {
            ruiz = __builtin_amdgcn_readfirstlane(ss.preconditioner) == 1;
            if (ruiz) rz_c = ruiz_compute_wave(n, m, Hw, ldw, v.h, Aw, ldw, v.al, v.au, v.lx, v.ux, rz);
        }
        // 7-argument form: zero guesses (Q2)
        if constexpr (SCH) {
            boxadmm_solve_schur<Model, SCH_P, SCH_S>(hblk, hbrd, v.h, ocp.jblk, ocp.s.D, ocp.s.nsr, v.al, v.au, v.lx, v.ux, qs, qi, qw.x, qw.y, tr, qblk, xsc, dsc, dtab,
                                                     PROF ? &cyc[PROF ? 6 : 0] : nullptr, PROF ? &cyc[PROF ? 16 : 0] : nullptr);
            wsync();
        } else if constexpr (CND) {
            if constexpr (POL) {
                // the node blocks of the sparse view mirror the workspace: after the equilibration they are read back from it (E_r A(r, c) D_c, the entries the
                // dense orders see), and the D~ tables are then built from the workspace, one set per state index (ws_on; the launcher sized the staging for them)
                if (ruiz) {
                    constexpr int NXc = Model::NX, NUc = Model::NU, NDERc = Dm::NDER, JBSc = Dm::JBS, NNc = NNODES_CT_, VXc = NXc * NNc, P0c = (NXc + NUc) * NNc;
                    for (int e = ln; e < MM * NDERc; e += WAVE) {
                        const int r = e / NDERc, i = e - r * NDERc;
                        const int k = (r < VXc) ? r / NXc : (r - VXc) / (Model::NG > 0 ? Model::NG : 1);
                        const int c = (i < NXc) ? k * NXc + i : ((i < NXc + NUc) ? VXc + k * NUc + (i - NXc) : P0c + (i - NXc - NUc));
                        ocp.jblk[r * JBSc + i] = Aw[(size_t)c * ldw + r];
                    }
                    wsync();
                }
                boxadmm_solve_cond<NN, MM, JV, true>(Hw, v.h, v.al, v.au, v.lx, v.ux, qs, qi, qw.x, qw.y, tr, jview(), PROF ? &cyc[PROF ? 6 : 0] : nullptr, PROF ? &cyc[PROF ? 16 : 0] : nullptr, ruiz);
            } else
            boxadmm_solve_cond<NN, MM>(Hw, v.h, v.al, v.au, v.lx, v.ux, qs, qi, qw.x, qw.y, tr, jview(), PROF ? &cyc[PROF ? 6 : 0] : nullptr, PROF ? &cyc[PROF ? 16 : 0] : nullptr);
            wsync();
        } else if constexpr (REG2) {   // (lower-triangle read of H always: the Hessian update is a run-time choice in these kernels)
            // (POL: the Ruiz preconditioner may have rescaled the workspace, whose entries the blocks of the sparse view then no longer are: dense residuals)
            if constexpr (POL) boxadmm_solve_reg2<NN, MM, true, true>(Hw, v.h, Aw, v.al, v.au, v.lx, v.ux, nullptr, nullptr, qs, qi, qw.x, qw.y, tr, PROF ? &cyc[PROF ? 6 : 0] : nullptr, PROF ? &cyc[PROF ? 16 : 0] : nullptr);
            else boxadmm_solve_reg2<NN, MM, true, true>(Hw, v.h, Aw, v.al, v.au, v.lx, v.ux, nullptr, nullptr, qs, qi, qw.x, qw.y, tr, PROF ? &cyc[PROF ? 6 : 0] : nullptr, PROF ? &cyc[PROF ? 16 : 0] : nullptr, jview());
            wsync();
        } else if constexpr (REG1) { boxadmm_solve_reg<NN, MM, true, (HU == 1) || POL || PMPC_EXPERIMENT_FORCE_SYMLOWER>(   /* lower-triangle read of H: the block BFGS and a Ruiz-scaled H (D_i H_ij D_j) are not bitwise symmetric */ Hw, v.h, Aw, v.al, v.au, v.lx, v.ux, nullptr, nullptr, qs, qi, qw.x, qw.y, tr, PROF ? &cyc[PROF ? 6 : 0] : nullptr, PROF ? &cyc[PROF ? 16 : 0] : nullptr); wsync(); }
        else {
            // Solver<Problem, ADMM<...>>: the launcher sized the QP's LDS for the stacked (2n+m)-row system when qp_solver = 1
            if (RUIZ_COMPILED && __builtin_amdgcn_readfirstlane(ss.qp_solver) == 1) admm_solve(qw, n, m, Hw, ldw, v.h, Aw, ldw, v.al, v.au, v.lx, v.ux, nullptr, nullptr, qs, qi);
            else {
                long long tq[6] = {0, 0, 0, 0, 0, 0};
                if constexpr (BIG) {
                    // large instances: condensed linear algebra from the block-sparse view of J (n instead of n + m rows) — unless the Ruiz preconditioner
                    // rescaled the workspace, whose entries the per-node blocks of the view then no longer are
                    const JViewRT<Model> jvr{ocp.Dlds, ocp.s.nsr, ocp.jblk, ocp.gblk, ocp.P, ocp.dm.NN, ocp.dm.VARX, ocp.dm.VARU, ocp.dm.me, ocp.jtab};
                    boxadmm_solve<true, JViewRT<Model>, SLIM, BIG_NW>(qw, n, m, Hw, ldw, v.h, Aw, ldw, v.al, v.au, v.lx, v.ux, nullptr, nullptr, qs, qi, PROF ? tq : nullptr, jvr,
                                                        ocp.keep_blk && !ruiz && n <= BIG_COND_MAX_ROWS && m <= BIG_COND_MAX_ROWS && __builtin_amdgcn_readfirstlane(ss.kkt_form) == 0,
                                                        (BigMail<JViewRT<Model>>*)big_mail);
                } else
                boxadmm_solve<BIG>(qw, n, m, Hw, ldw, v.h, Aw, ldw, v.al, v.au, v.lx, v.ux, nullptr, nullptr, qs, qi, PROF ? tq : nullptr);
                acc(6, tq[0]); acc(7, tq[1]); acc(17, tq[2]); acc(16, tq[3]);   // (slots 16 / 17 double as "KKT build" / "substitutions" on the LDS path)
                if constexpr (BIG) { acc(18, tq[4]); acc(19, tq[5]); }   // (condensed mode — slot 18: the A'(rho r2) product, slot 19: the two triangular passes)
            }
        }
        qp_iter_total += qi.iter;
        qp_flags |= qi.flags;
        qp_iter_last = qi.iter; qp_status_last = qi.status;
        if constexpr (BIG || SCH) { if (__builtin_amdgcn_readfirstlane(qi.flags & PMPC_FLAG_ILLCOND) != 0) { qp_flags |= FLAG_GAVE_UP; return; } }   // the QP gave up at its conditioning gate: nothing of this solve is used (solve() ends it with PMPC_SQP_REDO)
        if constexpr (RUIZ_COMPILED) if (ruiz) {   // unscale(p, p_lambda); unscale(m_H, m_h, m_A, ...), sqp_base.hpp:608-609 / :664-665
            ruiz_unscale_solution_wave(n, m, rz.D, rz.E, rz_c, qw.x, qw.y);
            ruiz_unscale_problem_wave(n, m, Hw, ldw, v.h, Aw, ldw, v.al, v.au, v.lx, v.ux, rz.D, rz.E, rz_c);
        }
        // lam_k = p_lambda ; p_lambda -= lam
        for (int i = ln; i < m + n; i += WAVE) { v.lam_k[i] = qw.y[i]; qw.y[i] = qw.y[i] - v.lam[i]; }
        wsync();
        const long long q1 = now();
        const double alpha = step_size_selection();
        const long long q2 = now();
        acc(1, q1 - q0); acc(2, q2 - q1);
        const double pn = lds_inf_norm(qw.x, n), dn = lds_inf_norm(qw.y, m + n);
        for (int i = ln; i < n; i += WAVE) { const double st = alpha * qw.x[i]; v.x[i] += st; v.step[i] = st; }
        for (int i = ln; i < m + n; i += WAVE) v.lam[i] += alpha * qw.y[i];
        primal_norm = alpha * pn;
        dual_norm = alpha * dn;
        alpha_log = alpha;
        wsync();
    }
    __device__ __forceinline__ bool termination_criteria() {  // :524-529
        max_violation = max_constraints_violation(v.x);
        return (primal_norm <= ss.eps_prim) && (dual_norm <= ss.eps_dual) && (max_violation <= ss.eps_prim);
    }

    // SQP iterations it_begin+1 .. min(it_end, max_iter). status PMPC_SQP_IN_PROGRESS (internal) when the slice ends first.
    __device__ __forceinline__ void solve(pmpc_sqp_info& info, int it_begin, int it_end) {
        int status = PMPC_SQP_MAX_ITER_EXCEEDED;
        int iter = it_begin;
        // single code site for the (large) QP + line-search body: first pass = exact linearisation (:583), later = update (:649)
        while (true) {
            ++iter;
            const long long c0 = now();
            linearise(iter == 1 || ss.exact_hessian_every_iter, true);
            const long long c1 = now();
            qp_and_step();
            if constexpr (BIG || SCH) { if (qp_flags & FLAG_GAVE_UP) { status = PMPC_SQP_REDO; break; } }
            const long long c2 = now();
            const bool done = __builtin_amdgcn_readfirstlane((int)termination_criteria()) != 0;
            const long long c3 = now();
            acc(0, c1 - c0); acc(3, c3 - c2); acc(4, c3 - c0); (void)c2;
            if (trace != nullptr && iter <= ss.iteration_trace_capacity && lane_id() == 0) {   // what sqp_settings_t::iteration_callback could read (sqp_base.hpp:685-686)
                double* r = trace + (size_t)(iter - 1) * PMPC_TRACE_DOUBLES;
                r[0] = (double)iter; r[1] = alpha_log; r[2] = primal_norm; r[3] = dual_norm; r[4] = cost_log;
                r[5] = (double)qp_iter_last; r[6] = (double)qp_status_last; r[7] = max_violation;
#ifdef PMPC_EXPERIMENT_WG_STAMPS   /* developer build (tests/experiments/launch_timeline.py): wall-clock stamps (100 MHz) instead of alpha / the step norms — when this iteration ended, when the workgroup started, where it ran */
                r[1] = (double)wall_clock64(); r[2] = (double)t_start_stamp;
                { unsigned xcc_; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(xcc_)); r[3] = (double)(xcc_ & 7u); }
#endif
            }
            if (done) { status = PMPC_SQP_SOLVED; break; }
            if (iter >= ss.max_iter) break;
            if (iter >= it_end) { status = PMPC_SQP_IN_PROGRESS; break; }
        }
        {   // a non-finite iterate is reported whatever produced it (a QP solution, a diverging step of a warm start from an unconverged point, the model itself)
            bool bad = false;
            for (int i = lane_id(); i < n_ct(); i += WAVE) bad |= (v.x[i] - v.x[i]) != 0.0;
            for (int i = lane_id(); i < m_ct() + n_ct(); i += WAVE) bad |= (v.lam[i] - v.lam[i]) != 0.0;
            if (__builtin_amdgcn_ballot_w64(bad) != 0) qp_flags |= PMPC_FLAG_NONFINITE;
        }
        info.iter = iter; info.qp_solver_iter = qp_iter_total; info.status = status; info.flags = qp_flags & ~FLAG_GAVE_UP;
        info.primal_norm = primal_norm; info.dual_norm = dual_norm; info.max_violation = max_violation; info.cost = cost_log;
    }
};

}  // namespace pmpc
