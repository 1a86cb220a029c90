// ORACLE — TEST INFRASTRUCTURE ONLY (CPU restatement of the reference algorithm).
//
// Dense box-ADMM QP solver + the LDL^T it relies on.
// Follows /root/reference/src/solvers/qp_base.hpp (settings :17-53, status :55-62, info :64-72,
// parse_constraints_bounds :195-222, residuals :240-252, DIV_BY_ZERO_REGUL :79-82,126) and
// /root/reference/src/solvers/box_admm.hpp (solve_impl :88-205, construct_kkt_matrix :209-223,
// factorise :336-341, compute_kkt_rhs :351-355, rho_vec_update :357-396, residuals_update :398-415,
// eps_prim/eps_dual/termination :417-431, estimate_rho :433-445, update_kkt_rho :448-452).
//
// Third-party arithmetic: the reference factorises with Eigen::LDLT<Matrix,Lower>
// (src/utils/helpers.hpp:38-43; Eigen 3.3.7 pinned in ci/install-linux.sh:21). Eigen is NOT vendored in
// /root/reference; the two pivot policies below restate its published algorithm:
//   PIVOT_EIGEN  : symmetric max-|diag| pivoting, left-looking column update, D^+ solve (Eigen semantics)
//   PIVOT_SWEEP  : no factorisation at all — W = -K^{-1} by the symmetric sweep operator in blocks of 4 pivots
//                  (static order; since round 4 the diagonal constraint block is swept in closed form first, BoxADMM::factorise_sweep_cf — a bare
//                  matrix handed to LDLT::compute is swept pivot block by pivot block as before)
//                  and x = -(W b) as a mat-vec (block partial sums). This is the arithmetic of the register-resident HIP
//                  kernel (polympc_amd/csrc/pmpc_qp_reg.hpp), restated operation by operation so that the kernel can be
//                  checked bit for bit; it is tied to the reference only through PIVOT_EIGEN (tests/test_oracle_pins.py
//                  compares the two policies on the reference's QP fixtures and on the SQP workloads).
//   PIVOT_STATIC : identical arithmetic without the permutation, right-looking update order — the order
//                  the HIP kernels use, so a GPU-vs-oracle comparison isolates kernel bugs from pivot effects.
//   PIVOT_BLOCKED: PIVOT_STATIC's factor and forward substitution, backward substitution by column dot products in blocks of 16 (the blocked
//                  tile LDL^T of pmpc_qp_big.hpp, KKT systems that live in HBM)
//   PIVOT_SWEEP2 : PIVOT_SWEEP's blocked sweep for 65..128 rows with the mat-vec order of the two-rows-per-lane register kernel
//   PIVOT_SWEEP1 : the swept inverse of PIVOT_SWEEP one pivot at a time on the lower triangle (any size), x = -(W b) as one fma chain
//                  per row: accuracy evidence for the explicit-inverse route above 64 rows (no shipped kernel uses it this round).
//   PIVOT_EXACT  : (round 5) not a kernel order — PIVOT_EIGEN's factor used as a preconditioner for iterative refinement with residuals in
//                  long double (x87 extended, 64-bit mantissa) until the correction stalls: the linear solves of the ADMM to working accuracy,
//                  i.e. the trajectory of exact arithmetic as far as fp64 can state it. The yardstick that tells WHICH of two orders that
//                  disagree at a large penalty is the inexact one (tests/test_oracle_pins.py).
//   PIVOT_SCHUR  : (round 4) the order of the block-structured kernel (polympc_amd/csrc/pmpc_qp_schur.hpp) that serves a Hessian which is
//                  block diagonal per collocation node — what ContinuousOCP's block BFGS (continuous_ocp.hpp:2304-2431) and the exact Lagrangian
//                  Hessian keep. The per-node blocks of H + sigma I + rho_box are inverted one by one, the m x m Schur complement
//                  1/rho + A P^{-1} A' is swept like PIVOT_SWEEP, and a solve is two block products, two sparse products with A and one
//                  m x m mat-vec (BoxADMM::factorise_schur / kkt_solve_schur below).
// All matrices column-major.
#pragma once
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <limits>
#include <stdexcept>
#include <vector>
#include <atomic>

namespace oracle {

enum qp_status { QP_SOLVED = 0, QP_MAX_ITER_EXCEEDED = 1, QP_UNSOLVED = 2, QP_UNINITIALIZED = 3, QP_INFEASIBLE = 4, QP_INCONSISTENT = 5 };
enum pivot_policy { PIVOT_EIGEN = 0, PIVOT_STATIC = 1, PIVOT_SWEEP = 2, PIVOT_SWEEP1 = 3, PIVOT_SWEEP2 = 4, PIVOT_BLOCKED = 5, PIVOT_CONDENSED = 6, PIVOT_SCHUR = 7, PIVOT_CONDSWEEP = 8, PIVOT_EXACT = 9 };

struct qp_settings {  // qp_base.hpp:17-53 (ADMM-related subset)
    double eps_rel = 1e-3, eps_abs = 1e-3;
    int max_iter = 1000;
    bool warm_start = false;
    double rho = 1e-1, sigma = 1e-6, alpha = 1.0;
    int check_termination = 25;
    bool adaptive_rho = false;
    double adaptive_rho_tolerance = 5;
    int adaptive_rho_interval = 25;
};

struct qp_info {  // qp_base.hpp:64-72
    int status = QP_UNINITIALIZED;
    int iter = 0;
    int rho_updates = 0;
    double rho_estimate = 0;
    double res_prim = 1, res_dual = 1;
    int flags = 0;   // QP_FLAG_ILLCOND: the conditioning gate of the condensed / constraint-first orders tripped (BoxADMM::COND_GATE)
};
enum { QP_FLAG_ILLCOND = 2 };   // = PMPC_FLAG_ILLCOND of include/polympc_amd.h

// ---------------------------------------------------------------------------------------------
// LDL^T of a symmetric matrix given by its LOWER triangle (Appendix B of SURVEY.md)
struct LDLT {
    int n = 0;
    pivot_policy policy = PIVOT_EIGEN;
    std::vector<double> M;   // factor: unit-lower L below the diagonal, D on the diagonal
    std::vector<int> tr;     // transpositions
    std::vector<double> temp;
    double piv_min_abs = 0.0;   // smallest |pivot| of the static / swept orders (the conditioning gate of BoxADMM reads it)
    bool exact = false;         // PIVOT_EXACT: refine every solve against K0 in long double
    std::vector<double> K0;     // PIVOT_EXACT: the matrix itself, full symmetric storage

    void compute(const std::vector<double>& K, int n_, pivot_policy pol) {
        n = n_; policy = pol; M = K; tr.assign(n, 0); temp.assign(n, 0.0);
        exact = policy == PIVOT_EXACT;
        if (exact) {
            policy = PIVOT_EIGEN;
            K0 = K;
            for (int j = 0; j < n; ++j) for (int i = 0; i < j; ++i) K0[i + j * n] = K0[j + i * n];
        }
        if (policy == PIVOT_STATIC || policy == PIVOT_BLOCKED) { compute_static(); return; }
        if (policy == PIVOT_SWEEP1) { compute_sweep1(); return; }
        if (policy == PIVOT_SWEEP2) {  // the two-rows-per-lane register kernel (pmpc_qp_reg2.hpp): the same blocked sweep on 65..128 rows
            if (n > 128) throw std::invalid_argument("oracle: PIVOT_SWEEP2 restates the 128-row register kernel; use PIVOT_STATIC / PIVOT_EIGEN for larger systems");
            compute_sweep(); return;
        }
        if (policy == PIVOT_SWEEP) {   // mirrors the register-resident kernel, which exists for at most 64 KKT rows (4 column blocks of 16)
            if (n > 64) throw std::invalid_argument("oracle: PIVOT_SWEEP restates the 64-row register kernel; use PIVOT_STATIC / PIVOT_EIGEN for larger systems");
            compute_sweep(); return;
        }
        auto at = [&](int i, int j) -> double& { return M[i + j * n]; };
        for (int k = 0; k < n; ++k) {
            // largest remaining |diagonal| (first occurrence)
            int big = k; double bv = std::fabs(at(k, k));
            for (int i = k + 1; i < n; ++i) { double v = std::fabs(at(i, i)); if (v > bv) { bv = v; big = i; } }
            tr[k] = big;
            if (big != k) {
                for (int j = 0; j < k; ++j) std::swap(at(k, j), at(big, j));
                for (int i = big + 1; i < n; ++i) std::swap(at(i, k), at(i, big));
                std::swap(at(k, k), at(big, big));
                for (int i = k + 1; i < big; ++i) std::swap(at(i, k), at(big, i));
            }
            const int rs = n - k - 1;
            if (k > 0) {
                for (int j = 0; j < k; ++j) temp[j] = at(j, j) * at(k, j);
                double acc = 0.0;
                for (int j = 0; j < k; ++j) acc += at(k, j) * temp[j];
                at(k, k) -= acc;
                for (int i = k + 1; i < n; ++i) {
                    double a = 0.0;
                    for (int j = 0; j < k; ++j) a += at(i, j) * temp[j];
                    at(i, k) -= a;
                }
            }
            const double akk = at(k, k);
            const bool valid = std::fabs(akk) > 0.0;
            if (k == 0 && !valid) { for (int j = 0; j < n; ++j) tr[j] = j; return; }
            if (rs > 0 && valid) for (int i = k + 1; i < n; ++i) at(i, k) /= akk;
        }
    }

    // no pivoting; right-looking: after column k is scaled, a_ij -= a_ik(unscaled) * l_jk
    void compute_static() {
        auto at = [&](int i, int j) -> double& { return M[i + j * n]; };
        for (int k = 0; k < n; ++k) tr[k] = k;
        std::vector<double> col(n);
        piv_min_abs = std::numeric_limits<double>::infinity();
        for (int k = 0; k < n; ++k) {
            const double dk = at(k, k);
            piv_min_abs = std::fmin(piv_min_abs, std::fabs(dk));
            for (int i = k + 1; i < n; ++i) { col[i] = at(i, k); at(i, k) = col[i] / dk; }
            for (int j = k + 1; j < n; ++j) {
                const double ljk = at(j, k);
                for (int i = j; i < n; ++i) at(i, j) = std::fma(-col[i], ljk, at(i, j));
            }
        }
    }

    // Symmetric sweep operator, blocks of BK = 4 pivots, static order. After all sweeps M = -K^{-1} (full storage).
    // Block step on pivots kb..kb+BK-1 (panel p = M[:, block], Cold = its copy):
    //   in-panel scalar sweeps   t = 0..BK-1, k = kb+t:  r = 1/p[k][t];  l_i = p[i][t]*r;
    //                            u != t:  p[i][u] = fma(-l_i, p[k][u], p[i][u]) (i != k),  p[k][u] = p[k][u]*r;
    //                            p[i][t] = l_i (i != k),  p[k][t] = -r
    //   trailing update          i, j outside the block, i/16 >= j/16 (block-lower storage in 16x16 tiles; the other
    //                            entries are their mirror images):  M[i][j] = fma(-p[i][t], Cold[j][t], M[i][j]),  t ascending
    //   write-back               M[:, block] = p,  then M[block, :] = p^T
    // PIVOT_SWEEP1: W = -K^{-1} by the symmetric sweep operator one pivot at a time on the lower triangle (any size) — the order of the
    // LDS-resident inverse kernel: for pivot k, r = 1/K_kk, l_i = K_ik * r, every other lower entry (i >= j, i, j != k) becomes
    // fma(-l_i, K_jk, K_ij) with the UNSCALED column entries K_jk of before the step; then column k <- l, K_kk <- -r.
    void compute_sweep1() {
        auto lo = [&](int i, int j) -> double& { return i >= j ? M[i + j * n] : M[j + i * n]; };
        for (int k = 0; k < n; ++k) tr[k] = k;
        std::vector<double> c(n), l(n);
        for (int k = 0; k < n; ++k) {
            const double r = 1.0 / lo(k, k);
            for (int i = 0; i < n; ++i) { c[i] = lo(i, k); l[i] = c[i] * r; }
            for (int j = 0; j < n; ++j) {
                if (j == k) continue;
                for (int i = j; i < n; ++i) { if (i == k) continue; M[i + j * n] = std::fma(-l[i], c[j], M[i + j * n]); }
            }
            for (int i = 0; i < n; ++i) if (i != k) lo(i, k) = l[i];
            lo(k, k) = -r;
        }
        for (int j = 0; j < n; ++j) for (int i = 0; i < j; ++i) M[i + j * n] = M[j + i * n];   // mirror for solve()
    }

    // tiles_as_given: the diagonal 16 x 16 tiles are taken as they are (both triangles as the caller computed them — PIVOT_SCHUR forms its rows
    // one per lane and the two triangles of a diagonal tile differ in the last bit); tiles above the block diagonal are mirror images as always
    // npiv >= 0: only the pivots [0, npiv) are swept (the others were eliminated in closed form by the caller, BoxADMM::factorise_sweep_cf)
    void compute_sweep(bool tiles_as_given = false, int npiv = -1) {
        auto at = [&](int i, int j) -> double& { return M[i + j * n]; };
        for (int k = 0; k < n; ++k) tr[k] = k;
        for (int j = 0; j < n; ++j) for (int i = 0; i < j; ++i) if (!tiles_as_given || i / 16 < j / 16) at(i, j) = at(j, i);   // full symmetric storage
        const int BK = 4;   // block size of the kernel (RegKkt::BK)
        const int lim = npiv < 0 ? n : npiv;
        std::vector<double> p((size_t)n * BK), cold((size_t)n * BK), l(n);
        piv_min_abs = std::numeric_limits<double>::infinity();
        for (int kb = 0; kb < lim; kb += BK) {
            const int w = std::min(BK, lim - kb);
            for (int t = 0; t < w; ++t) for (int i = 0; i < n; ++i) { p[i * BK + t] = at(i, kb + t); cold[i * BK + t] = p[i * BK + t]; }
            for (int t = 0; t < w; ++t) {
                const int k = kb + t;
                piv_min_abs = std::fmin(piv_min_abs, std::fabs(p[k * BK + t]));
                const double r = 1.0 / p[k * BK + t];
                for (int i = 0; i < n; ++i) l[i] = p[i * BK + t] * r;
                for (int u = 0; u < w; ++u) {
                    if (u == t) continue;
                    const double rk = p[k * BK + u];
                    for (int i = 0; i < n; ++i) p[i * BK + u] = (i == k) ? rk * r : std::fma(-l[i], rk, p[i * BK + u]);
                }
                for (int i = 0; i < n; ++i) p[i * BK + t] = (i == k) ? -r : l[i];
            }
            // block-lower storage in 16x16 tiles: an entry (i, j) with i/16 < j/16 is not stored, it IS entry (j, i)
            for (int j = 0; j < n; ++j) {
                if (j >= kb && j < kb + w) continue;
                for (int i = 0; i < n; ++i) {
                    if ((i >= kb && i < kb + w) || i / 16 < j / 16) continue;
                    double a = at(i, j);
                    for (int t = 0; t < w; ++t) a = std::fma(-p[i * BK + t], cold[j * BK + t], a);
                    at(i, j) = a;
                }
            }
            for (int t = 0; t < w; ++t) for (int i = 0; i < n; ++i) at(i, kb + t) = p[i * BK + t];
            for (int t = 0; t < w; ++t) for (int j = 0; j < n; ++j) at(kb + t, j) = p[j * BK + t];
            for (int j = 0; j < n; ++j) for (int i = 0; i < j; ++i) if (i / 16 < j / 16) at(i, j) = at(j, i);
        }
    }

    void solve(const double* b, double* x) const {
        if (exact) {
            std::vector<long double> xl(n), r(n);
            std::vector<double> rd(n), dx(n);
            solve_plain(b, x);
            for (int i = 0; i < n; ++i) xl[i] = x[i];
            long double prev = -1.0L;
            for (int it = 0; it < 12; ++it) {
                long double rn = 0.0L;
                for (int i = 0; i < n; ++i) {
                    long double a = b[i];
                    for (int j = 0; j < n; ++j) a -= (long double)K0[i + j * n] * xl[j];
                    r[i] = a; rd[i] = (double)a; rn = std::fmax(rn, std::fabs(a));
                }
                if (rn == 0.0L || (prev >= 0.0L && rn >= prev)) break;   // the correction stalls
                prev = rn;
                solve_plain(rd.data(), dx.data());
                for (int i = 0; i < n; ++i) xl[i] += dx[i];
            }
            for (int i = 0; i < n; ++i) x[i] = (double)xl[i];
            return;
        }
        solve_plain(b, x);
    }
    void solve_plain(const double* b, double* x) const {
        auto at = [&](int i, int j) -> double { return M[i + j * n]; };
        if (policy == PIVOT_SWEEP1) {   // x = -(W b): one fma chain per row, columns ascending
            for (int i = 0; i < n; ++i) { double a = 0.0; for (int j = 0; j < n; ++j) a = std::fma(at(i, j), b[j], a); x[i] = -a; }
            return;
        }
        if (policy == PIVOT_SWEEP2) {  // x = -(W b): four fma chains per row, chain q over the columns j = q (mod 4) ascending (the columns one
                                       // 16-lane row of the wavefront owns in the accumulator-tile layout), combined as (P0+P2)+(P1+P3)
            for (int i = 0; i < n; ++i) {
                double acc[4] = {0.0, 0.0, 0.0, 0.0};
                for (int j = 0; j < n; ++j) acc[j & 3] = std::fma(at(i, j), b[j], acc[j & 3]);
                x[i] = -((acc[0] + acc[2]) + (acc[1] + acc[3]));
            }
            return;
        }
        if (policy == PIVOT_SWEEP) {   // x = -(W b): one fma chain per block of 16 columns (P_r, r = j/16), combined as (P0+P2)+(P1+P3)
            for (int i = 0; i < n; ++i) {
                double acc[4] = {0.0, 0.0, 0.0, 0.0};
                for (int j = 0; j < n; ++j) acc[j >> 4] = std::fma(at(i, j), b[j], acc[j >> 4]);
                x[i] = -((acc[0] + acc[2]) + (acc[1] + acc[3]));
            }
            return;
        }
        for (int i = 0; i < n; ++i) x[i] = b[i];
        if (policy == PIVOT_EIGEN) for (int k = 0; k < n; ++k) if (tr[k] != k) std::swap(x[k], x[tr[k]]);
        if (policy == PIVOT_BLOCKED) {
            // The large-instance kernel (pmpc_qp_big.hpp): factor and forward substitution exactly as PIVOT_STATIC; the BACKWARD substitution reads the
            // factor by columns — the layout the forward pass streams — instead of by rows: per block of 16 columns (descending) the contributions of
            // the rows below the block are column dot products, summed in the kernel's order — 64 partial sums per column (row r of the rows below goes
            // to partial (r - 16(J+1)) mod 64, rows ascending, fma), the partials added in four groups of 16 (index order inside a group, (S0+S1)+(S2+S3) across) — subtracted once, then the 16 x 16 triangle of the
            // block as in PIVOT_STATIC. (No second, row-ordered copy of L is stored or read.)
            for (int j = 0; j < n; ++j) for (int i = j + 1; i < n; ++i) x[i] = std::fma(-at(i, j), x[j], x[i]);
            for (int i = 0; i < n; ++i) x[i] = x[i] / at(i, i);
            const int nb = (n + 15) / 16;
            for (int J = nb - 1; J >= 0; --J) {
                const int lo = 16 * J, hi = std::min(lo + 16, n), below = 16 * (J + 1);
                for (int c = lo; c < hi; ++c) {
                    double P[64]; for (int l = 0; l < 64; ++l) P[l] = 0.0;
                    for (int r = below; r < n; ++r) P[(r - below) & 63] = std::fma(at(r, c), x[r], P[(r - below) & 63]);
                    double S[4];   // four groups of 16 partials, each added in index order, combined as (S0 + S1) + (S2 + S3)
                    for (int k = 0; k < 4; ++k) { double a = 0.0; for (int u = 0; u < 16; ++u) a += P[16 * k + u]; S[k] = a; }
                    const double sum = (S[0] + S[1]) + (S[2] + S[3]);
                    x[c] = x[c] - sum;
                }
                for (int j = hi - 1; j >= lo; --j) for (int i = j - 1; i >= lo; --i) x[i] = std::fma(-at(j, i), x[j], x[i]);
            }
            return;
        }
        if (policy == PIVOT_STATIC) {
            // column-oriented forward substitution, fma order of the HIP kernel
            for (int j = 0; j < n; ++j) for (int i = j + 1; i < n; ++i) x[i] = std::fma(-at(i, j), x[j], x[i]);
            for (int i = 0; i < n; ++i) x[i] = x[i] / at(i, i);
            for (int j = n - 1; j >= 0; --j) for (int i = j - 1; i >= 0; --i) x[i] = std::fma(-at(j, i), x[j], x[i]);
            return;
        }
        for (int i = 0; i < n; ++i) { double a = x[i]; for (int j = 0; j < i; ++j) a -= at(i, j) * x[j]; x[i] = a; }
        const double tol = 1.0 / std::numeric_limits<double>::max();
        for (int i = 0; i < n; ++i) { if (std::fabs(at(i, i)) > tol) x[i] /= at(i, i); else x[i] = 0.0; }
        for (int i = n - 1; i >= 0; --i) { double a = x[i]; for (int j = i + 1; j < n; ++j) a -= at(j, i) * x[j]; x[i] = a; }
        for (int k = n - 1; k >= 0; --k) if (tr[k] != k) std::swap(x[k], x[tr[k]]);
    }
};

// ---------------------------------------------------------------------------------------------
struct BoxADMM {
    static constexpr double RHO_MIN = 1e-6, RHO_MAX = 1e+6, RHO_EQ_FACTOR = 1e+3;   // box_admm.hpp:56-59
    static constexpr double LOOSE_BOUNDS_THRESH = 1e+10, EQ_TOL = 1e-4;            // qp_base.hpp:124-125
    static constexpr double DIV_BY_ZERO_REGUL = 10e-10;                            // qp_base.hpp:79-82
    enum ctype { INEQUALITY_CONSTRAINT = 0, EQUALITY_CONSTRAINT = 1, LOOSE_BOUNDS = 2 };

    int N, M;
    qp_settings settings;
    qp_info info;
    pivot_policy pivot = PIVOT_EIGEN;
    // PIVOT_SCHUR: the collocation structure of the QP (variables [x_0 .. x_{nn-1} | u_0 .. u_{nn-1}], equality rows (node, state), P intervals per
    // segment — continuous_ocp.hpp:757-765, :797-878); set by the SQP driver from the problem's dimensions, by tests through orc_set_schur_structure
    struct SchurStruct { int nx = 0, nu = 0, nn = 0, P = 0, np = 0; } schur;   // np = 1: one parameter behind the node variables (bordered system, below)
    std::vector<double> x, y;  // primal N, dual M+N ([general | box])
    std::vector<double> x_tilde, q, z, z_tilde, z_prev, rho_vec, rho_inv_vec, rho_box, rho_box_inv, rho_box_prev;
    std::vector<int> constr_type, box_type;
    std::vector<double> K;
    LDLT ldlt;
    double rho = 0, max_Ax_z_norm = 0, max_Hx_ATy_h_norm = 0;
    int iter = 0;
    // Conditioning gate of the orders that eliminate the diagonal constraint block first (PIVOT_SWEEP's constraint-first sweep, PIVOT_CONDSWEEP,
    // PIVOT_CONDENSED): they invert / factorise S = P + A' diag(rho) A, whose condition number is rho_eq |A|^2 / lambda_min(P on the null space of A).
    // With bounded controls (every BASELINE workload) that is ~1e5 whatever rho is — rho_box scales with rho — and these orders are MORE accurate than
    // the pivoted LDL^T of the quasi-definite form; when unbounded variables (rho_box = RHO_MIN) span the null space of A it grows with rho and the
    // condensed solve loses cond(S) eps. Estimate at every factorisation: max_i S_ii * max_i |(S^-1)_ii| (the swept orders: both diagonals are at hand) or
    // max_i S_ii / min_k |d_k| (PIVOT_CONDENSED's LDL^T) — the two agree within a factor of two and lie a factor 2..10 below cond(S) on the benchmark
    // streams and their unbounded variants; beyond COND_GATE
    // PIVOT_CONDENSED (the large-instance kernel) and PIVOT_SWEEP at the QP entry point give the QP up (status QP_UNSOLVED, flag set): the drivers (SQP:
    // the whole instance, from its guesses; the QP entry point: the QP) solve it again in the full KKT form, as the product's redo launches do —
    // PIVOT_SWEEP -> PIVOT_STATIC (the LDS-resident static LDL^T), PIVOT_CONDENSED -> PIVOT_BLOCKED (the (n + m)-row blocked LDL^T).
    // The register-resident SQP kernels (PIVOT_SWEEP, PIVOT_CONDSWEEP inside an SQP solve) decide ONCE per instance from its bounds instead — an unbounded
    // control or parameter sends the instance to the redo launch before any work (SQP driver, oracle_capi.cpp): a numeric gate cost those kernels 4 .. 10 %.
    static constexpr double COND_GATE = 1e10, SCHUR_COND_GATE = 1e7;
    bool numeric_gate = true;   // PIVOT_SWEEP: the QP entry point's kernels evaluate the gate numerically; the fused SQP kernels decide from the bounds (SQP driver) and switch this off
    bool illcond = false;
    double cond_estimate = 0.0;

    BoxADMM(int n, int m) : N(n), M(m) {
        x.assign(N, 0); y.assign(N + M, 0); x_tilde.assign(N, 0); q.assign(N, 0);
        z.assign(M, 0); z_tilde.assign(M, 0); z_prev.assign(M, 0);
        rho_vec.assign(M, settings.rho); rho_inv_vec.assign(M, 1 / settings.rho);
        rho_box.assign(N, 0); rho_box_inv.assign(N, 0); rho_box_prev.assign(N, 0);
        constr_type.assign(M, 0); box_type.assign(N, 0);
        K.assign((N + M) * (N + M), 0.0);
    }

    static double inf_norm(const double* v, int n) { double r = 0; for (int i = 0; i < n; ++i) r = std::fmax(r, std::fabs(v[i])); return r; }

    static int classify(double lb, double ub) {  // qp_base.hpp:195-222
        if (lb < -LOOSE_BOUNDS_THRESH && ub > LOOSE_BOUNDS_THRESH) return LOOSE_BOUNDS;
        if (ub - lb < EQ_TOL) return EQUALITY_CONSTRAINT;
        return INEQUALITY_CONSTRAINT;
    }

    void rho_vec_update(double rho0) {  // box_admm.hpp:357-396
        for (int i = 0; i < M; ++i) {
            switch (constr_type[i]) {
                case LOOSE_BOUNDS: rho_vec[i] = RHO_MIN; break;
                case EQUALITY_CONSTRAINT: rho_vec[i] = RHO_EQ_FACTOR * rho0; break;
                default: rho_vec[i] = rho0;
            }
            rho_inv_vec[i] = 1.0 / rho_vec[i];
        }
        rho = rho0;
        for (int i = 0; i < N; ++i) {
            switch (box_type[i]) {
                case LOOSE_BOUNDS: rho_box[i] = RHO_MIN; break;
                case EQUALITY_CONSTRAINT: rho_box[i] = RHO_EQ_FACTOR * rho0; break;
                default: rho_box[i] = rho0;
            }
            rho_box_inv[i] = 1.0 / rho_box[i];
        }
        info.rho_updates += 1;
    }

    void matvec(const double* A, int rows, int cols, const double* v, double* out) const {
        for (int i = 0; i < rows; ++i) { double a = 0; for (int j = 0; j < cols; ++j) a += A[i + j * rows] * v[j]; out[i] = a; }
    }
    void matTvec(const double* A, int rows, int cols, const double* v, double* out) const {
        for (int j = 0; j < cols; ++j) { double a = 0; for (int i = 0; i < rows; ++i) a += A[i + j * rows] * v[i]; out[j] = a; }
    }

    // H x of the dual residual from the KKT identity instead of a second mat-vec with H (round 6):
    //   (H + sigma I + rho_box) x~ + A' nu = rhs_1   =>   H x~ = (rhs_1 - A' nu) - (sigma + rho_box) o x~        (alpha = 1: x = x~, quirk Q1)
    // * PIVOT_CONDSWEEP and PIVOT_CONDENSED restate the condensed register kernel (pmpc_qp_cond.hpp) and the large-instance kernel's condensed mode
    //   (pmpc_qp.hpp, qp_residuals_sparse), which do this whenever alpha == 1 and the iterate is finite: A' nu is the
    //   fma chain of kkt_solve_condsweep's first product started from 0, then one subtraction, one product, one subtraction. The reference (and every other
    //   order here) multiplies by H: qp_base.hpp:240-252. The two differ by the linear solve's own residual; on the streams of configs A / D / B / R no instance
    //   changes an iteration count (tests/test_oracle_pins.py::test_dual_residual_from_the_kkt_identity_changes_no_trajectory).
    // * hx_identity() — a process-wide TEST switch (orc_set_hx_identity) that applies the same identity, with dense products, under any order: the experiment
    //   behind the statement above. last_rhs / last_sol: the operands of the last solve.
    static int& hx_identity() { static int v = 0; return v; }
    std::vector<double> last_rhs, last_sol;
    void residuals_update(const double* H, const double* h, const double* A) {  // :398-415
        std::vector<double> Ax(M), Hx(N), ATy(N);
        matvec(A, M, N, x.data(), Ax.data());
        const double norm_Ax = inf_norm(Ax.data(), M), norm_z = inf_norm(z.data(), M);
        max_Ax_z_norm = std::fmax(norm_Ax, std::fmax(norm_z, inf_norm(x.data(), N)));
        bool finite_iterate = true;   // (the kernel's test: x and the constraint multipliers)
        for (int i = 0; i < N; ++i) finite_iterate = finite_iterate && std::isfinite(x[i]);
        for (int i = 0; i < M; ++i) finite_iterate = finite_iterate && std::isfinite(y[i]);
        if (pivot == PIVOT_CONDSWEEP && settings.alpha == 1.0 && finite_iterate && (int)last_sol.size() == N + M) {
            const int NM = N + M, nx = schur.nx, nu = schur.nu, nn = schur.nn, VARX = nx * nn, N0 = N - schur.np, ME = nx * nn, ng = (M - ME) / nn;
            const double* nuv = last_sol.data() + N;
            if (schur.np) {   // the parameter's column: the wavefront's tree (cond_wave_dot)
                double hx = last_rhs[N0] - cond_wave_dot(nuv);
                hx -= (settings.sigma + rho_box[N0]) * x[N0];
                Hx[N0] = hx;
            }
            for (int c = 0; c < N0; ++c) {
                const bool xcol = c < VARX;
                const int jn = xcol ? c / nx : (c - VARX) / nu, qx = xcol ? c - jn * nx : 0;
                double a = 0.0;
                if (c < 64 || VARX > 64)
                    for (int k = 0; k < nn; ++k) { const double coef = (xcol && k != jn) ? K[(N + k * nx + qx) + c * NM] : 0.0; a = std::fma(coef, nuv[k * nx + qx], a); }
                for (int q = 0; q < nx; ++q) a = std::fma(K[(N + jn * nx + q) + c * NM], nuv[jn * nx + q], a);
                for (int g = 0; g < ng; ++g) a = std::fma(K[(N + ME + jn * ng + g) + c * NM], nuv[ME + jn * ng + g], a);
                double hx = last_rhs[c] - a;
                hx -= (settings.sigma + rho_box[c]) * x[c];
                Hx[c] = hx;
            }
        } else if (pivot == PIVOT_CONDENSED && settings.alpha == 1.0 && finite_iterate && (int)last_sol.size() == N + M) {
            // the large-instance kernel's condensed mode (pmpc_qp.hpp qp_residuals_sparse, ident): A' nu as the fma chain of kkt_solve's first product — the entries of
            // column c that are not exactly zero, rows ascending — started from 0
            const int NM = N + M;
            const double* nuv = last_sol.data() + N;
            for (int c = 0; c < N; ++c) {
                double a = 0.0;
                for (int r = 0; r < M; ++r) { const double v = K[(N + r) + c * NM]; if (v != 0.0) a = std::fma(v, nuv[r], a); }
                double hx = last_rhs[c] - a;
                hx -= (settings.sigma + rho_box[c]) * x[c];
                Hx[c] = hx;
            }
        } else if (hx_identity() && settings.alpha == 1.0 && (int)last_sol.size() == N + M) {
            std::vector<double> ATnu(N);
            matTvec(A, M, N, last_sol.data() + N, ATnu.data());
            for (int i = 0; i < N; ++i) Hx[i] = (last_rhs[i] - (settings.sigma + rho_box[i]) * x[i]) - ATnu[i];
        } else
        matvec(H, N, N, x.data(), Hx.data());
        matTvec(A, M, N, y.data(), ATy.data());
        const double norm_Hx = inf_norm(Hx.data(), N), norm_ATy = inf_norm(ATy.data(), N);
        const double norm_h = inf_norm(h, N), norm_ybox = inf_norm(y.data() + M, N);
        max_Hx_ATy_h_norm = std::fmax(norm_Hx, std::fmax(norm_ATy, std::fmax(norm_h, norm_ybox)));
        double rp = 0, rq = 0, rd = 0;
        for (int i = 0; i < M; ++i) rp = std::fmax(rp, std::fabs(Ax[i] - z[i]));
        for (int i = 0; i < N; ++i) rq = std::fmax(rq, std::fabs(x[i] - q[i]));
        info.res_prim = rp + rq;
        for (int i = 0; i < N; ++i) rd = std::fmax(rd, std::fabs(((Hx[i] + h[i]) + ATy[i]) + y[M + i]));  // qp_base.hpp:251
        info.res_dual = rd;
    }
    double eps_prim() const { return settings.eps_abs + settings.eps_rel * max_Ax_z_norm; }
    double eps_dual() const { return settings.eps_abs + settings.eps_rel * max_Hx_ATy_h_norm; }
    bool termination_criteria() const { return info.res_prim <= eps_prim() && info.res_dual <= eps_dual(); }
    double estimate_rho(double rho0) const {  // :433-445
        double rp = info.res_prim / (max_Ax_z_norm + DIV_BY_ZERO_REGUL);
        double rd = info.res_dual / (max_Hx_ATy_h_norm + DIV_BY_ZERO_REGUL);
        return rho0 * std::sqrt(rp / (rd + DIV_BY_ZERO_REGUL));
    }

    void construct_kkt(const double* H, const double* A) {  // :209-223 (lower triangle only)
        const int NM = N + M;
        std::fill(K.begin(), K.end(), 0.0);
        for (int j = 0; j < N; ++j) for (int i = 0; i < N; ++i) K[i + j * NM] = H[i + j * N];
        for (int i = 0; i < N; ++i) K[i + i * NM] += settings.sigma;
        for (int i = 0; i < N; ++i) K[i + i * NM] += rho_box[i];
        for (int j = 0; j < N; ++j) for (int i = 0; i < M; ++i) K[(N + i) + j * NM] = A[i + j * M];
        for (int i = 0; i < M; ++i) K[(N + i) + (N + i) * NM] = -rho_inv_vec[i];
    }
    void update_kkt_rho() {  // :448-452
        const int NM = N + M;
        for (int i = 0; i < N; ++i) K[i + i * NM] += (rho_box[i] - rho_box_prev[i]);
        for (int i = 0; i < M; ++i) K[(N + i) + (N + i) * NM] = -rho_inv_vec[i];
    }
    // PIVOT_CONDENSED — the order of the large-instance SQP kernel since round 3 (pmpc_qp_big.hpp, condensed mode). The constraint block of K is
    // diagonal, -diag(1 / rho): eliminating it first leaves the n x n SPD system
    //     S x = r1 + A'(rho o r2),   S = H + sigma I + rho_box + A' diag(rho) A,   nu = rho o (A x - r2)
    // — the same solution in exact arithmetic from a matrix of n instead of n + m rows (config C: 256 instead of 464; the factor the substitutions
    // stream is 3.3 x smaller). Operation order restated here: S_ij = K_ij, then fma(rho_r A_ri, A_rj, .) for r ascending (the k-ascending chain of
    // the matrix cores, every r included); S factorised and solved in PIVOT_BLOCKED's order; t_i = r1_i, then fma(A_ri, rho_r r2_r, .) over the rows
    // r ascending with A_ri != 0; nu_r = rho_r ((sum_j fma(A_rj, x_j, .), A_rj != 0, j ascending) - r2_r). Entries of A that are exactly zero are
    // skipped in the two vector products, which is what a kernel that walks the block-sparse structure of A does — and unlike "adds an exact zero"
    // it is the same statement for non-finite operands.
    std::vector<double> Sc;
    // ---- PIVOT_SCHUR -------------------------------------------------------------------------------------------------------------------------
    // K = [P A'; A -1/rho] with P = H + sigma I + rho_box block diagonal per node (d = nx + nu entries: x_k then u_k) and A = J of a
    // Chebyshev collocation: row (ni, si) holds the differentiation-matrix entry Dt(ni, k) on column (k, si) for the nodes k != ni of the
    // segment that produces node ni, and the own-node block b (d entries, the D self entry included) on the columns of node ni.
    //   factorise:  Q_k = P_k^{-1}  (symmetric sweep on the lower triangle, one pivot at a time, pivots ascending; Q = -(swept matrix), mirrored)
    //               G row i:  g_k[c] = sum_c' fma(b_i[c'], Q_k(c', c), .)  (k = ni, c' ascending from 0);  Dt(ni, k) * Q_k(si, c)  (k coupled);  0
    //               S(i, j) = [i == j] / rho_i, then over the nodes k ascending: fma(g_k[c], b_j[c], .) c ascending for k = nj, fma(g_k[sj], Dt(nj, k), .) for k coupled to nj
    //               W = -S^{-1}  by PIVOT_SWEEP's blocked sweep on the block-lower tiles, diagonal tiles as the rows gave them
    //   solve:      t = Q r1 (per node, fma chain c' ascending);  g_i = (own block fma chain over c, then coupled nodes k ascending) - r2_i;
    //               nu = S^{-1} g (PIVOT_SWEEP's mat-vec);  w = A' nu (own block: rows q ascending; then coupled row nodes ascending);
    //               x = Q (r1 - w);  then one refinement step on the constraint rows (kkt_solve_schur)
    //   np = 1 (round 5: minimal_time_test.cpp's grid): the parameter p is the last variable; H has the arrow shape (node blocks, a border row / column,
    //   a corner) and A a dense column a_p. With K0 = [P A_z'; A_z -1/rho] the solve above (`solve0`), w = [H(p, z); a_p] and pi = H_pp + sigma + rho_p:
    //               factorise:  (q_z, q_nu) = solve0(w);  delta = pi - w'q
    //               solve:      (z0, nu0) = solve0(r1_z, r2);  p = (r1_p - w'[z0; nu0]) / delta;  z = z0 - p q_z  (fma),  nu = nu0 - p q_nu  (fma)
    //   w'v is summed the way the wavefront does: lane l = 0..63 forms fma chains over its primal slots g = l + 64 e (ascending e), then its constraint
    //   row (l < m), from 0; the 64 partial sums are added pairwise over adjacent lanes, level by level (schur_wave_dot).
    std::vector<double> Qs, gws, bq;   // bq: [q_z (n0) | q_nu (m)] of the border
    double bdelta = 0.0;
    int n0() const { return N - schur.np; }
    int sg(int k, int c) const { return c < schur.nx ? k * schur.nx + c : schur.nx * schur.nn + k * schur.nu + (c - schur.nx); }
    bool coupled(int rownode, int k) const {   // Dt(rownode, k) structurally non-zero, own node excluded (its D entry lives in the block)
        const int kb = (rownode == schur.nn - 1) ? schur.nn - 1 - schur.P : (rownode / schur.P) * schur.P;
        return k != rownode && k >= kb && k <= kb + schur.P;
    }
    double Aent(int r, int c) const { return K[(N + r) + c * (N + M)]; }
    void schur_check() const {
        const int nx = schur.nx, nu = schur.nu, nn = schur.nn, d = nx + nu, NM = N + M;
        const int N0 = n0();
        if (nx < 1 || nn < 2 || schur.P < 1 || (nn - 1) % schur.P != 0 || d * nn != N0 || nx * nn != M || schur.np < 0 || schur.np > 1)
            throw std::invalid_argument("oracle: PIVOT_SCHUR needs the collocation structure (nx, nu, nn, P[, np <= 1]) of a QP with n = (nx+nu) nn + np, m = nx nn");
        std::vector<int> node(N0), loc(N0);
        for (int k = 0; k < nn; ++k) for (int c = 0; c < d; ++c) { node[sg(k, c)] = k; loc[sg(k, c)] = c; }
        for (int j = 0; j < N0; ++j) for (int i = j; i < N0; ++i)
            if (node[i] != node[j] && K[i + j * NM] != 0.0) throw std::invalid_argument("oracle: PIVOT_SCHUR: the Hessian is not block diagonal per node");
        for (int r = 0; r < M; ++r) for (int c = 0; c < N0; ++c) {
            const int ni = r / nx, si = r % nx;
            const bool ok = node[c] == ni || (loc[c] == si && coupled(ni, node[c]));
            if (!ok && Aent(r, c) != 0.0) throw std::invalid_argument("oracle: PIVOT_SCHUR: A is not a collocation Jacobian of the given structure");
        }
    }
    void factorise_schur() {
        schur_check();
        const int nx = schur.nx, nn = schur.nn, d = nx + schur.nu, NM = N + M;
        Qs.assign((size_t)nn * d * d, 0.0);
        std::vector<double> Mk(d * d), c(d), l(d);
        for (int k = 0; k < nn; ++k) {
            for (int j = 0; j < d; ++j) for (int i = j; i < d; ++i) Mk[i + j * d] = K[sg(k, i) + sg(k, j) * NM];
            auto lo = [&](int i, int j) -> double& { return i >= j ? Mk[i + j * d] : Mk[j + i * d]; };
            for (int p = 0; p < d; ++p) {
                const double r = 1.0 / lo(p, p);
                for (int i = 0; i < d; ++i) { c[i] = lo(i, p); l[i] = c[i] * r; }
                for (int j = 0; j < d; ++j) {
                    if (j == p) continue;
                    for (int i = j; i < d; ++i) { if (i == p) continue; Mk[i + j * d] = std::fma(-l[i], c[j], Mk[i + j * d]); }
                }
                for (int i = 0; i < d; ++i) if (i != p) lo(i, p) = l[i];
                lo(p, p) = -r;
            }
            for (int j = 0; j < d; ++j) for (int i = 0; i < d; ++i) Qs[(size_t)k * d * d + i + j * d] = -lo(i, j);
        }
        auto Q = [&](int k, int i, int j) { return Qs[(size_t)k * d * d + i + j * d]; };
        std::vector<double> S((size_t)M * M, 0.0), g((size_t)nn * d);
        for (int i = 0; i < M; ++i) {
            const int ni = i / nx, si = i % nx;
            for (int k = 0; k < nn; ++k)
                for (int cc = 0; cc < d; ++cc) {
                    double a = 0.0;
                    if (k == ni) { for (int c2 = 0; c2 < d; ++c2) a = std::fma(Aent(i, sg(ni, c2)), Q(k, c2, cc), a); }
                    else a = (coupled(ni, k) ? Aent(i, k * nx + si) : 0.0) * Q(k, si, cc);   // (every lane forms the product; uncoupled nodes carry the coefficient 0)
                    g[(size_t)k * d + cc] = a;
                }
            for (int j = 0; j < M; ++j) {
                if (j / 16 > i / 16) continue;   // block-lower tile storage: tiles above the block diagonal are mirror images
                const int nj = j / nx, sj = j % nx;
                double a = (i == j) ? rho_inv_vec[i] : 0.0;
                for (int k = 0; k < nn; ++k) {   // nodes ascending: the own node of row j contributes its block (c ascending), a coupled node one product
                    if (k == nj) { for (int cc = 0; cc < d; ++cc) a = std::fma(g[(size_t)k * d + cc], Aent(j, sg(nj, cc)), a); }
                    else if (coupled(nj, k)) a = std::fma(g[(size_t)k * d + sj], Aent(j, k * nx + sj), a);
                }
                S[i + (size_t)j * M] = a;
            }
        }
        ldlt.n = M; ldlt.policy = PIVOT_SWEEP; ldlt.M = S; ldlt.tr.assign(M, 0); ldlt.temp.assign(M, 0.0);
        if (M > 64) throw std::invalid_argument("oracle: PIVOT_SCHUR restates a kernel with at most 64 constraint rows");
        double smax = 0.0;
        for (int a = 0; a < M; ++a) smax = std::fmax(smax, std::fabs(S[a + (size_t)a * M]));
        ldlt.compute_sweep(true);
        // conditioning gate (round 5): cond(S) = rho_eq lambda_max(A Q A') once 1/rho is what keeps S regular — unbounded states whose dynamics do not depend on
        // them (A_x singular: a constant state profile) — and the range-space solve then loses what the (n + m)-row orders keep; max S_ii max |(S^-1)_ii| off the swept tiles
        // (x = Q (r1 - A' nu) is a difference of quantities ~ rho_eq times larger than itself: the error of a solve grows like the estimate squared — robot,
        //  16 nodes: 2e-9 at 4e6, 2e-5 at 4e7; parking: 4e-7 at 4e8, 1e-3 at 4e9, nothing at 4e10. With the gate at 1e7 every instance this order keeps follows
        //  exact arithmetic as closely as Eigen's pivoted order does, whatever rho (tests/test_oracle_pins.py); the redo is the static LDL^T of the (n + m)-row matrix)
        if (gate_trips_inverse(smax, M, SCHUR_COND_GATE)) { illcond = true; info.flags |= QP_FLAG_ILLCOND; }
        if (schur.np == 1) {   // border: (q_z, q_nu) = K0^{-1} [H(p, z); a_p], delta = (H_pp + sigma + rho_p) - w'q
            const int N0 = n0();
            std::vector<double> w1(N0), w2(M);
            for (int g = 0; g < N0; ++g) w1[g] = K[N0 + (size_t)g * NM];
            for (int i = 0; i < M; ++i) w2[i] = K[(N + i) + (size_t)N0 * NM];
            bq.assign(N0 + M, 0.0);
            schur_solve0(w1.data(), w2.data(), bq.data(), bq.data() + N0);
            bdelta = K[N0 + (size_t)N0 * NM] - schur_wave_dot(bq.data(), bq.data() + N0);
        }
    }
    // x = Q (r1 - A' nu) for a given nu: w = A' nu (own block: rows q ascending; then the coupled row nodes ascending), u = r1 - w, x = Q u
    void schur_primal(const double* rhs, const double* nu, double* xs) const {
        const int nx = schur.nx, nn = schur.nn, d = nx + schur.nu;
        auto Q = [&](int k, int i, int j) { return Qs[(size_t)k * d * d + i + j * d]; };
        std::vector<double> u(n0());
        for (int k = 0; k < nn; ++k) for (int cc = 0; cc < d; ++cc) {
            double a = 0.0;
            for (int q2 = 0; q2 < nx; ++q2) a = std::fma(Aent(k * nx + q2, sg(k, cc)), nu[k * nx + q2], a);
            // every row node enters the chain: coefficient D~(kr, k) on a state column whose node lies in the segment that produces node kr, 0 otherwise
            // (own node, other segments, control columns — which read the state-0 entry of nu)
            const int cx = cc < nx ? cc : 0;
            for (int kr = 0; kr < nn; ++kr) a = std::fma((cc < nx && coupled(kr, k)) ? Aent(kr * nx + cc, k * nx + cc) : 0.0, nu[kr * nx + cx], a);
            u[sg(k, cc)] = rhs[sg(k, cc)] - a;
        }
        for (int k = 0; k < nn; ++k) for (int cc = 0; cc < d; ++cc) {
            double a = 0.0;
            for (int c2 = 0; c2 < d; ++c2) a = std::fma(Q(k, cc, c2), u[sg(k, c2)], a);
            xs[sg(k, cc)] = a;
        }
    }
    // (A v)_i as the kernel forms it: own block fma chain over c ascending, then the coupled nodes k ascending, from `init`
    double schur_rowdot(int i, const double* v, double init) const {
        const int nx = schur.nx, nn = schur.nn, d = nx + schur.nu, ni = i / nx, si = i % nx;
        double a = init;
        for (int cc = 0; cc < d; ++cc) a = std::fma(Aent(i, sg(ni, cc)), v[sg(ni, cc)], a);
        for (int k = 0; k < nn; ++k) a = std::fma(coupled(ni, k) ? Aent(i, k * nx + si) : 0.0, v[k * nx + si], a);   // every node enters: 0 outside the row's segment and on the own node
        return a;
    }
    // One solve = the range-space solve and ONE step of iterative refinement on the constraint rows. The explicit (swept) inverse of S has an
    // isotropic forward error ~ eps cond(S) |nu|, whereas x = Q (r1 - A' nu) tolerates errors of nu only in the near-null directions of
    // Q^(1/2) A' (measured on config B's QPs after a rho update, cond(S) = 6e5: |dx| 2e-9 .. 2e-8 without the step, 2e-13 with it — the dense
    // orders reach 8e-12). The first block row holds to working precision by construction, so the residual lives in the second one:
    //   e = (A x - nu / rho) - r2,   nu += S^{-1} e... sign: K [dx; dnu] = [0; -e]  <=>  -S dnu = -e,   then x = Q (r1 - A' nu) again.
    // schur_refine_gate() — a process-wide TEST switch (orc_set_schur_refine_gate): the refinement step only when the factorisation's conditioning estimate exceeds it
    // (0: always, the shipped order). schur_refine_counts(): solves with / without the step since the last reset (the experiment's statistics).
    static double& schur_refine_gate() { static double v = 0.0; return v; }
    static std::atomic<long long>* schur_refine_counts() { static std::atomic<long long> c[2]; return c; }
    void schur_solve0(const double* r1, const double* r2, double* xs, double* nu) {
        const int nx = schur.nx, nn = schur.nn, d = nx + schur.nu, N0 = n0();
        auto Q = [&](int k, int i, int j) { return Qs[(size_t)k * d * d + i + j * d]; };
        std::vector<double> t(N0), gv(M), dnu(M);
        for (int k = 0; k < nn; ++k) for (int cc = 0; cc < d; ++cc) {
            double a = 0.0;
            for (int c2 = 0; c2 < d; ++c2) a = std::fma(Q(k, cc, c2), r1[sg(k, c2)], a);
            t[sg(k, cc)] = a;
        }
        for (int i = 0; i < M; ++i) gv[i] = schur_rowdot(i, t.data(), 0.0) - r2[i];
        ldlt.solve(gv.data(), nu);
        schur_primal(r1, nu, xs);
        const bool refine = !(cond_estimate <= schur_refine_gate());
        schur_refine_counts()[refine ? 0 : 1]++;
        if (!refine) return;
        for (int i = 0; i < M; ++i) gv[i] = std::fma(-rho_inv_vec[i], nu[i], schur_rowdot(i, xs, 0.0)) - r2[i];
        ldlt.solve(gv.data(), dnu.data());
        for (int i = 0; i < M; ++i) nu[i] = nu[i] + dnu[i];
        schur_primal(r1, nu, xs);
    }
    // w'[z; nu] as the wavefront sums it (np = 1): w = [H(p, z) | a_p] read from K's border row / the parameter column of A
    double schur_wave_dot(const double* z, const double* nu) const {
        const int N0 = n0(), NM = N + M;
        double part[64];
        for (int l = 0; l < 64; ++l) {
            double a = 0.0;
            for (int g = l; g < N0; g += 64) a = std::fma(K[N0 + (size_t)g * NM], z[g], a);
            if (l < M) a = std::fma(K[(N + l) + (size_t)N0 * NM], nu[l], a);
            part[l] = a;
        }
        for (int w = 1; w < 64; w *= 2) for (int l = 0; l < 64; l += 2 * w) part[l] = part[l] + part[l + w];
        return part[0];
    }
    void kkt_solve_schur(const double* rhs, double* sol) {
        const int N0 = n0();
        if (schur.np == 0) { schur_solve0(rhs, rhs + N, sol, sol + N); return; }
        std::vector<double> z(N0), nu(M);
        schur_solve0(rhs, rhs + N, z.data(), nu.data());
        const double pv = (rhs[N0] - schur_wave_dot(z.data(), nu.data())) / bdelta;
        for (int g = 0; g < N0; ++g) z[g] = std::fma(-pv, bq[g], z[g]);
        for (int i = 0; i < M; ++i) nu[i] = std::fma(-pv, bq[N0 + i], nu[i]);
        for (int g = 0; g < N0; ++g) sol[g] = z[g];
        sol[N0] = pv;
        for (int i = 0; i < M; ++i) sol[N + i] = nu[i];
    }
    // PIVOT_SWEEP since round 4 (the one-row-per-lane register kernel, pmpc_qp_reg.hpp): the constraint block of K is diagonal, -1/rho, and is swept in
    // CLOSED FORM first — sweeping pivot n + j of [P A'; A -1/rho] adds rho_j A_j' A_j to the primal block, turns row / column n + j into -rho_j A_j and
    // the pivot into rho_j — so that only the n primal pivots go through the blocked sweep (config A: 35 instead of 56, nine blocks instead of fourteen):
    //   M(a, b)      = K(a, b), then fma(rho_j A(j, a), A(j, b), .) for j ascending       (a, b primal; block-lower tiles, diagonal tiles in full)
    //   M(n + j, b)  = M(b, n + j) = -(rho_j A(j, b)),   M(n + j, n + j') = [j == j'] rho_j
    // then PIVOT_SWEEP's blocked sweep over the pivots [0, n). W = -K^{-1} as before; the mat-vec of solve() is unchanged.
    bool gate_trips(double smax) {   // PIVOT_CONDENSED (an LDL^T, no inverse at hand): max S_ii / min |d_k|. (NaN operands: no trip — a non-finite solve is reported by its own flag)
        cond_estimate = smax / ldlt.piv_min_abs;
        return smax > COND_GATE * ldlt.piv_min_abs;
    }
    bool gate_trips_inverse(double smax, int nprimal, double gate = COND_GATE) {   // the swept orders: max S_ii * max |(S^-1)_ii| from the diagonal of the swept matrix (the kernels read it off their tiles)
        double wmax = 0.0;
        for (int a = 0; a < nprimal; ++a) wmax = std::fmax(wmax, std::fabs(ldlt.M[a + (size_t)a * ldlt.n]));
        cond_estimate = smax * wmax;
        return smax * wmax > gate;
    }
    void factorise_sweep_cf() {
        const int NM = N + M;
        std::vector<double> Mm((size_t)NM * NM, 0.0);
        for (int b = 0; b < N; ++b)
            for (int a = 0; a < N; ++a) {
                if (a / 16 < b / 16) continue;
                double v = a >= b ? K[a + b * NM] : K[b + a * NM];
                for (int j = 0; j < M; ++j) v = std::fma(rho_vec[j] * K[(N + j) + a * NM], K[(N + j) + b * NM], v);
                Mm[a + (size_t)b * NM] = v;
            }
        for (int j = 0; j < M; ++j) {
            for (int b = 0; b < N; ++b) {
                const double v = -(rho_vec[j] * K[(N + j) + b * NM]);
                Mm[(N + j) + (size_t)b * NM] = v;
                if (b / 16 == (N + j) / 16) Mm[b + (size_t)(N + j) * NM] = v;   // the part of the transposed block that shares a diagonal tile with it
            }
            Mm[(N + j) + (size_t)(N + j) * NM] = rho_vec[j];
        }
        double smax = 0.0;
        for (int a = 0; a < N; ++a) smax = std::fmax(smax, std::fabs(Mm[a + (size_t)a * NM]));
        ldlt.n = NM; ldlt.policy = PIVOT_SWEEP; ldlt.M.swap(Mm); ldlt.tr.assign(NM, 0); ldlt.temp.assign(NM, 0.0);
        if (NM > 64) throw std::invalid_argument("oracle: PIVOT_SWEEP restates the 64-row register kernel; use PIVOT_STATIC / PIVOT_EIGEN for larger systems");
        ldlt.compute_sweep(true, N);
        if (numeric_gate && gate_trips_inverse(smax, N)) { illcond = true; info.flags |= QP_FLAG_ILLCOND; }
    }
    // PIVOT_CONDSWEEP (the condensed register kernel, pmpc_qp_cond.hpp): the constraint block of K is eliminated in closed form as in factorise_sweep_cf,
    // but the constraint rows are not carried at all — only S = P + A' diag(rho) A (n x n; block-lower 16 x 16 tiles, diagonal tiles in full) is swept,
    //   S(a, b) = K(a, b) [lower-triangle read], then fma(rho_j A(j, a), A(j, b), .) for j ascending,
    // with PIVOT_SWEEP's blocked sweep (mat-vec order of PIVOT_SWEEP for n <= 64, of PIVOT_SWEEP2 above), and every solve is
    //   t = r1 + A'(rho o r2),   x = S^{-1} t,   nu = rho o (A x - r2)
    // with the two products formed as fma chains over the structural entries (kkt_solve_condsweep).
    void factorise_condsweep() {
        const int NM = N + M;
        if (N > 128) throw std::invalid_argument("oracle: PIVOT_CONDSWEEP restates the condensed register kernel: at most 128 primal rows");
        std::vector<double> Mm((size_t)N * N, 0.0);
        for (int b = 0; b < N; ++b)
            for (int a = 0; a < N; ++a) {
                if (a / 16 < b / 16) continue;
                double v = a >= b ? K[a + b * NM] : K[b + a * NM];
                for (int j = 0; j < M; ++j) v = std::fma(rho_vec[j] * K[(N + j) + a * NM], K[(N + j) + b * NM], v);
                Mm[a + (size_t)b * N] = v;
            }
        double smax = 0.0;
        for (int a = 0; a < N; ++a) smax = std::fmax(smax, std::fabs(Mm[a + (size_t)a * N]));
        ldlt.n = N; ldlt.policy = N <= 64 ? PIVOT_SWEEP : PIVOT_SWEEP2; ldlt.M.swap(Mm); ldlt.tr.assign(N, 0); ldlt.temp.assign(N, 0.0);
        ldlt.compute_sweep(true);
        (void)smax;   // (no numeric gate in the condensed register kernels: they decide from the bounds, once per instance — SQP driver)
    }
    // the two products as the kernel forms them (pmpc_qp_cond.hpp): fma chains — the differentiation-matrix entries of the column / row over the nodes
    // ascending (0 on the own node and outside the segments; a control column of the first 64 variables walks zeros), then the own node's block. Needs the
    // collocation structure (schur.nx, .nu, .nn; ng from m = (nx + ng) nn).
    // NP = 1 (round 6): the parameter is the last primal variable, its column of A is DENSE. Its entry of the first product is formed the way the wavefront forms it —
    // lane r multiplies A(r, p) with its own u_r, the 64 products are added pairwise over adjacent lanes, level by level (cond_wave_dot; the tree of wave_sum,
    // pmpc_qp.hpp) — and every row of the second product takes A(r, p) x_p as its last term (the parameter is the last column).
    double cond_wave_dot(const double* u) const {
        const int NM = N + M, N0 = N - schur.np;
        double part[64];
        for (int l = 0; l < 64; ++l) part[l] = (l < M) ? K[(N + l) + (size_t)N0 * NM] * u[l] : 0.0;
        for (int w = 1; w < 64; w *= 2) for (int l = 0; l < 64; l += 2 * w) part[l] = part[l] + part[l + w];
        return part[0];
    }
    void kkt_solve_condsweep(const double* rhs, double* sol) {
        const int NM = N + M, nx = schur.nx, nu = schur.nu, nn = schur.nn, VARX = nx * nn, np_ = schur.np, N0 = N - np_;
        if (nx < 1 || nn < 1 || np_ < 0 || np_ > 1 || M < nx * nn || (M - nx * nn) % nn != 0 || (nx + nu) * nn + np_ != N) throw std::invalid_argument("oracle: PIVOT_CONDSWEEP needs the collocation structure of the QP (nx, nu, nn[, np <= 1]; m = (nx + ng) nn)");
        // NG > 0 (round 6): the path-constraint rows ME + k ng + g follow the equality rows; such a row holds its own node's block only — a column takes its own node's
        // path rows behind the equality rows of its chain, a path row's product is its block (states, controls, the parameter)
        const int ME = nx * nn, ng = (M - ME) / nn;
        std::vector<double> u(M), t(N), xs(N);
        for (int r = 0; r < M; ++r) u[r] = rho_vec[r] * rhs[N + r];
        for (int c = 0; c < N0; ++c) {
            const bool xcol = c < VARX;
            const int jn = xcol ? c / nx : (c - VARX) / nu, qx = xcol ? c - jn * nx : 0;
            double a = rhs[c];
            if (c < 64 || VARX > 64)
                for (int k = 0; k < nn; ++k) { const double coef = (xcol && k != jn) ? K[(N + k * nx + qx) + c * NM] : 0.0; a = std::fma(coef, u[k * nx + qx], a); }
            for (int q = 0; q < nx; ++q) a = std::fma(K[(N + jn * nx + q) + c * NM], u[jn * nx + q], a);
            for (int g = 0; g < ng; ++g) a = std::fma(K[(N + ME + jn * ng + g) + c * NM], u[ME + jn * ng + g], a);
            t[c] = a;
        }
        if (np_) t[N0] = rhs[N0] + cond_wave_dot(u.data());
        ldlt.solve(t.data(), xs.data());
        for (int i = 0; i < N; ++i) sol[i] = xs[i];
        for (int r = 0; r < M; ++r) {
            const bool eq = r < ME;
            const int k = eq ? r / nx : (r - ME) / ng, q = eq ? r - k * nx : 0;
            double a = 0.0;
            for (int j = 0; j < nn; ++j) { const double coef = (eq && j != k) ? K[(N + r) + (j * nx + q) * NM] : 0.0; a = std::fma(coef, xs[j * nx + q], a); }
            for (int i = 0; i < nx; ++i) a = std::fma(K[(N + r) + (k * nx + i) * NM], xs[k * nx + i], a);
            for (int i = 0; i < nu; ++i) a = std::fma(K[(N + r) + (VARX + k * nu + i) * NM], xs[VARX + k * nu + i], a);
            if (np_) a = std::fma(K[(N + r) + (size_t)N0 * NM], xs[N0], a);
            sol[N + r] = rho_vec[r] * (a - rhs[N + r]);
        }
    }
    void factorise() {
        if (pivot == PIVOT_SCHUR) { factorise_schur(); return; }
        if (pivot == PIVOT_SWEEP) { factorise_sweep_cf(); return; }
        if (pivot == PIVOT_CONDSWEEP) { factorise_condsweep(); return; }
        if (pivot != PIVOT_CONDENSED) { ldlt.compute(K, N + M, pivot); return; }
        const int NM = N + M;
        Sc.assign((size_t)N * N, 0.0);
        for (int j = 0; j < N; ++j)
            for (int i = j; i < N; ++i) {
                double a = K[i + j * NM];
                for (int r = 0; r < M; ++r) a = std::fma(rho_vec[r] * K[(N + r) + i * NM], K[(N + r) + j * NM], a);
                Sc[i + j * N] = a;
            }
        ldlt.compute(Sc, N, PIVOT_BLOCKED);
        double smax = 0.0;
        for (int a = 0; a < N; ++a) smax = std::fmax(smax, std::fabs(Sc[a + (size_t)a * N]));
        if (gate_trips(smax)) { illcond = true; info.flags |= QP_FLAG_ILLCOND; }
    }
    bool gives_up() const { return illcond; }
    static pivot_policy redo_policy(pivot_policy p) { return (p == PIVOT_SWEEP || p == PIVOT_SCHUR) ? PIVOT_STATIC : (p == PIVOT_CONDSWEEP ? PIVOT_SWEEP2 : PIVOT_BLOCKED); }
    void kkt_solve(const double* rhs, double* sol) {
        if (pivot == PIVOT_SCHUR) { kkt_solve_schur(rhs, sol); return; }
        if (pivot == PIVOT_CONDSWEEP) { kkt_solve_condsweep(rhs, sol); return; }
        if (pivot != PIVOT_CONDENSED) { ldlt.solve(rhs, sol); return; }
        const int NM = N + M;
        std::vector<double> t(N), xs(N);
        for (int i = 0; i < N; ++i) {
            double a = rhs[i];
            for (int r = 0; r < M; ++r) { const double v = K[(N + r) + i * NM]; if (v != 0.0) a = std::fma(v, rho_vec[r] * rhs[N + r], a); }
            t[i] = a;
        }
        ldlt.solve(t.data(), xs.data());
        for (int i = 0; i < N; ++i) sol[i] = xs[i];
        for (int r = 0; r < M; ++r) {
            double a = 0.0;
            for (int j = 0; j < N; ++j) { const double v = K[(N + r) + j * NM]; if (v != 0.0) a = std::fma(v, xs[j], a); }
            sol[N + r] = rho_vec[r] * (a - rhs[N + r]);
        }
    }

    // 7-argument form (box_admm.hpp:81-86): zero guesses
    int solve(const double* H, const double* h, const double* A, const double* Alb, const double* Aub,
              const double* xlb, const double* xub) {
        std::vector<double> x0(N, 0.0), y0(N + M, 0.0);
        return solve(H, h, A, Alb, Aub, xlb, xub, x0.data(), y0.data());
    }

    // box_admm.hpp:88-205
    int solve(const double* H, const double* h, const double* A, const double* Alb, const double* Aub,
              const double* xlb, const double* xub, const double* x_guess, const double* y_guess) {
        const int NM = N + M;
        std::vector<double> rhs(NM), sol(NM);
        bool check_termination = false;
        for (int i = 0; i < N; ++i) x[i] = x_guess[i];
        for (int i = 0; i < NM; ++i) y[i] = y_guess[i];
        matvec(A, M, N, x_guess, z.data());
        for (int i = 0; i < N; ++i) q[i] = x_guess[i];
        for (int i = 0; i < M; ++i) constr_type[i] = classify(Alb[i], Aub[i]);
        for (int i = 0; i < N; ++i) box_type[i] = classify(xlb[i], xub[i]);
        illcond = false; info.flags = 0;
        rho_vec_update(settings.rho);
        construct_kkt(H, A);
        factorise();
        info.status = QP_UNSOLVED;
        if (gives_up()) { iter = 1; info.iter = 1; return info.status; }
        const double alpha = settings.alpha;

        for (iter = 1; iter <= settings.max_iter; iter++) {
            z_prev = z;
            // compute_kkt_rhs :351-355
            for (int i = 0; i < N; ++i) rhs[i] = ((settings.sigma * x[i] - h[i]) + rho_box[i] * q[i]) - y[M + i];
            for (int i = 0; i < M; ++i) rhs[N + i] = z[i] - rho_inv_vec[i] * y[i];
            kkt_solve(rhs.data(), sol.data());
            last_rhs = rhs; last_sol = sol;
            for (int i = 0; i < N; ++i) x_tilde[i] = sol[i];
            for (int i = 0; i < M; ++i) z_tilde[i] = z_prev[i] + rho_inv_vec[i] * (sol[N + i] - y[i]);
            // quirk Q1 (:129-130): x = alpha*x_tilde; x += (1-alpha)*x
            for (int i = 0; i < N; ++i) { x[i] = alpha * x_tilde[i]; x[i] += (1 - alpha) * x[i]; }
            for (int i = 0; i < M; ++i) {
                z[i] = alpha * z_tilde[i];
                z[i] += (1 - alpha) * z_prev[i] + rho_inv_vec[i] * y[i];
                z[i] = std::fmin(std::fmax(z[i], Alb[i]), Aub[i]);
            }
            for (int i = 0; i < N; ++i) {
                q[i] = x[i] + rho_box_inv[i] * y[M + i];
                q[i] = std::fmin(std::fmax(q[i], xlb[i]), xub[i]);
            }
            for (int i = 0; i < M; ++i) y[i] += rho_vec[i] * ((alpha * z_tilde[i] + (1 - alpha) * z_prev[i]) - z[i]);
            for (int i = 0; i < N; ++i) y[M + i] += rho_box[i] * (x[i] - q[i]);

            check_termination = (settings.check_termination != 0 && iter % settings.check_termination == 0);
            if (check_termination) {
                residuals_update(H, h, A);
                if (termination_criteria()) { info.status = QP_SOLVED; break; }
            }
            if (settings.adaptive_rho && iter % settings.adaptive_rho_interval == 0) {
                if (!check_termination) residuals_update(H, h, A);
                double new_rho = estimate_rho(rho);
                new_rho = std::fmax(RHO_MIN, std::fmin(new_rho, RHO_MAX));
                info.rho_estimate = new_rho;
                if (new_rho < rho / settings.adaptive_rho_tolerance || new_rho > rho * settings.adaptive_rho_tolerance) {
                    rho_box_prev = rho_box;
                    rho_vec_update(new_rho);
                    update_kkt_rho();
                    factorise();
                    if (gives_up()) { info.iter = iter; return info.status; }
                }
            }
        }
        if (iter > settings.max_iter) info.status = QP_MAX_ITER_EXCEEDED;
        info.iter = iter;
        return info.status;
    }
};

}  // namespace oracle
