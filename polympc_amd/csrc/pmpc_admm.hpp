// polympc_amd — the OSQP-style ADMM QP solver on the device, one wavefront per QP instance.
//
// Replaces polympc::ADMM<N, M, Scalar, DENSE> (/root/reference/src/solvers/admm.hpp): the box constraints are stacked under the
// general ones, A_e = [A ; I] (construct_A :215-222), so z, y and rho have M+N entries and there is ONE quasi-definite KKT
// matrix [H + sigma I, A_e' ; A_e, -diag(1/rho)] of dimension 2N+M (construct_kkt_matrix :249-263). solve_impl :112-212,
// compute_kkt_rhs :390-394, box_projection :397-403, rho_vec_update :405-440, residuals_update :442-462, termination and
// estimate_rho :464-488, update_kkt_rho :490-494. Differences from boxADMM that are kept: x = alpha*x_tilde + (1-alpha)*x
// (no quirk Q1), res_prim = max(|Ax - z_A|, |x - z_box|) instead of a sum, norm_Ax includes |x|.
//
// The factorisation, substitution and LDS layout are those of the LDS-resident boxADMM path (pmpc_qp.hpp: packed lower
// triangle, static right-looking LDL^T, fma substitutions) applied to the (2N+M)-row system, so the test suite checks this
// kernel against the CPU restatement with the same static order.
#pragma once
#include <hip/hip_runtime.h>
#include "pmpc_qp.hpp"

namespace pmpc {

// K (lower triangle) <- [H + diag(kdiag[0:n]) ; A ; I | diag(kdiag[n:])]
__device__ __forceinline__ void admm_kkt_build(const QpLds& w, int n, int m, const double* __restrict__ H, int ldh, const double* __restrict__ A, int lda) {
    const int ln = lane_id();
    const int me = m + n;
    for (int j = 0; j < n; ++j) {
        const int o = w.off(j);
        for (int i = j + ln; i < n; i += WAVE) w.K[o + i] = (i == j) ? w.kdiag[i] : H[(size_t)j * ldh + i];
        for (int r = ln; r < m; r += WAVE) w.K[o + n + r] = A[(size_t)j * lda + r];
        for (int i = ln; i < n; i += WAVE) w.K[o + n + m + i] = (i == j) ? 1.0 : 0.0;
    }
    for (int j = 0; j < me; ++j) {
        const int o = w.off(n + j);
        for (int i = j + ln; i < me; i += WAVE) w.K[o + n + i] = (i == j) ? w.kdiag[n + i] : 0.0;
    }
    wsync();
}

struct AdmmResidualState { double max_Ax_z_norm, max_Hx_ATy_h_norm, res_prim, res_dual; };

__device__ __forceinline__ void admm_residuals(const QpLds& w, int n, int m, const double* __restrict__ H, int ldh, const double* __restrict__ h,
                                      const double* __restrict__ A, int lda, AdmmResidualState& r) {
    const int ln = lane_id();
    double nAx = 0, nz = 0, rp = 0;
    for (int i = ln; i < m; i += WAVE) {
        const double a = seq_dot_strided(A, (size_t)lda, 1, i, n, w.x);
        nAx = fmax(nAx, fabs(a)); rp = fmax(rp, fabs(a - w.z[i]));
    }
    for (int i = ln; i < m + n; i += WAVE) nz = fmax(nz, fabs(w.z[i]));
    double nx = 0, nHx = 0, nATy = 0, nh = 0, nyb = 0, rb = 0, rd = 0;
    for (int i = ln; i < n; i += WAVE) {
        const double a = seq_dot_strided(H, (size_t)ldh, 1, i, n, w.x);
        const double b = (m > 0) ? seq_dot_strided(A, 1, (size_t)lda, i, m, w.y) : 0.0;
        nx = fmax(nx, fabs(w.x[i])); nHx = fmax(nHx, fabs(a)); nATy = fmax(nATy, fabs(b));
        nh = fmax(nh, fabs(h[i])); nyb = fmax(nyb, fabs(w.y[m + i]));
        rb = fmax(rb, fabs(w.x[i] - w.z[m + i]));
        rd = fmax(rd, fabs(((a + h[i]) + b) + w.y[m + i]));
    }
    r.max_Ax_z_norm = wave_max(fmax(fmax(nAx, nx), nz));
    r.max_Hx_ATy_h_norm = wave_max(fmax(fmax(nHx, nATy), fmax(nh, nyb)));
    r.res_prim = wave_max(fmax(rp, rb));
    r.res_dual = wave_max(rd);
}

__device__ __forceinline__ void admm_rho_vec_update(const QpLds& w, int n, int m, const double* Alb, const double* Aub, const double* xl, const double* xu, double rho0) {
    const int ln = lane_id();
    for (int i = ln; i < m + n; i += WAVE) {
        const int t = (i < m) ? classify_bounds(Alb[i], Aub[i]) : classify_bounds(xl[i - m], xu[i - m]);
        const double r = rho_of(t, rho0);
        w.rho[i] = r; w.rhoinv[i] = 1.0 / r;
    }
}

// ADMM::solve_impl. w carved with QpLds::carve(base, n, m + n). Result in w.x (n) and w.y (m+n).
__device__ __forceinline__ void admm_solve(QpLds& w, int n, int m, const double* __restrict__ H, int ldh, const double* h, const double* __restrict__ A, int lda,
                                  const double* Alb, const double* Aub, const double* xl, const double* xu, const double* x0, const double* y0,
                                  const pmpc_qp_settings& s, pmpc_qp_info& info) {
    const int ln = lane_id();
    const int me = m + n, N2 = n + me;
    for (int i = ln; i < n; i += WAVE) w.x[i] = x0 ? x0[i] : 0.0;
    for (int i = ln; i < me; i += WAVE) w.y[i] = y0 ? y0[i] : 0.0;
    wsync();
    for (int i = ln; i < m; i += WAVE) {
        double a = 0.0;
        if (x0) for (int j = 0; j < n; ++j) a += A[(size_t)j * lda + i] * w.x[j];
        w.z[i] = a;
    }
    for (int i = ln; i < n; i += WAVE) w.z[m + i] = w.x[i];
    double rho = s.rho;
    int rho_updates = 1;
    admm_rho_vec_update(w, n, m, Alb, Aub, xl, xu, rho);
    wsync();
    for (int i = ln; i < n; i += WAVE) { double dgl = H[(size_t)i * ldh + i]; dgl += s.sigma; w.kdiag[i] = dgl; }
    for (int i = ln; i < me; i += WAVE) w.kdiag[n + i] = -1.0 * w.rhoinv[i];
    wsync();
    admm_kkt_build(w, n, m, H, ldh, A, lda);
    kkt_factor(w, N2);

    int status = PMPC_QP_UNSOLVED;
    const double alpha = s.alpha;
    AdmmResidualState rs{0, 0, 1, 1};
    double rho_estimate = 0.0;
    int iter;
    for (iter = 1; iter <= s.max_iter; ++iter) {
        for (int i = ln; i < n; i += WAVE) w.rhs[i] = s.sigma * w.x[i] - h[i];
        for (int i = ln; i < me; i += WAVE) { w.zprev[i] = w.z[i]; w.rhs[n + i] = w.z[i] - w.rhoinv[i] * w.y[i]; }
        wsync();
        kkt_solve(w, N2, w.rhs);
        for (int i = ln; i < n; i += WAVE) w.x[i] = alpha * w.rhs[i] + (1 - alpha) * w.x[i];
        for (int i = ln; i < me; i += WAVE) {
            const double zt = w.zprev[i] + w.rhoinv[i] * (w.rhs[n + i] - w.y[i]);
            double zz = alpha * zt;
            zz += (1 - alpha) * w.zprev[i] + w.rhoinv[i] * w.y[i];
            const double lo = (i < m) ? Alb[i] : xl[i - m], hi = (i < m) ? Aub[i] : xu[i - m];
            zz = fmin(fmax(zz, lo), hi);
            w.z[i] = zz;
            w.y[i] += w.rho[i] * ((alpha * zt + (1 - alpha) * w.zprev[i]) - zz);
        }
        wsync();
        const bool check = (s.check_termination != 0 && iter % s.check_termination == 0);
        if (check) {
            admm_residuals(w, n, m, H, ldh, h, A, lda, rs);
            const double ep = s.eps_abs + s.eps_rel * rs.max_Ax_z_norm, ed = s.eps_abs + s.eps_rel * rs.max_Hx_ATy_h_norm;
            if (rs.res_prim <= ep && rs.res_dual <= ed) { status = PMPC_QP_SOLVED; break; }
        }
        if (s.adaptive_rho && iter % s.adaptive_rho_interval == 0) {
            if (!check) admm_residuals(w, n, m, H, ldh, h, A, lda, rs);
            const double rpn = rs.res_prim / (rs.max_Ax_z_norm + DIV_BY_ZERO_REGUL);
            const double rdn = rs.res_dual / (rs.max_Hx_ATy_h_norm + DIV_BY_ZERO_REGUL);
            double new_rho = rho * sqrt(rpn / (rdn + DIV_BY_ZERO_REGUL));
            new_rho = fmax(RHO_MIN, fmin(new_rho, RHO_MAX));
            rho_estimate = new_rho;
            if (new_rho < rho / s.adaptive_rho_tolerance || new_rho > rho * s.adaptive_rho_tolerance) {
                rho = new_rho;
                admm_rho_vec_update(w, n, m, Alb, Aub, xl, xu, rho);
                ++rho_updates;
                wsync();
                for (int i = ln; i < me; i += WAVE) w.kdiag[n + i] = -w.rhoinv[i];   // update_kkt_rho
                wsync();
                admm_kkt_build(w, n, m, H, ldh, A, lda);
                kkt_factor(w, N2);
            }
        }
    }
    if (iter > s.max_iter) status = PMPC_QP_MAX_ITER_EXCEEDED;
    info.status = status; info.iter = iter; info.rho_updates = rho_updates; info.flags = 0;
    info.rho_estimate = rho_estimate; info.res_prim = rs.res_prim; info.res_dual = rs.res_dual;
}

static __global__ __launch_bounds__(64) void qp_admm_kernel(int B, int n, int m, const double* __restrict__ H, const double* __restrict__ h,
                                                            const double* __restrict__ A, const double* __restrict__ Alb, const double* __restrict__ Aub,
                                                            const double* __restrict__ xlb, const double* __restrict__ xub, const double* __restrict__ x0,
                                                            const double* __restrict__ y0, pmpc_qp_settings s, double* __restrict__ x, double* __restrict__ y,
                                                            pmpc_qp_info* __restrict__ info) {
    extern __shared__ double smem[];
    const int b = blockIdx.x;
    if (b >= B) return;
    QpLds w;
    w.carve(smem, n, m + n);
    pmpc_qp_info qi;
    admm_solve(w, n, m, H + (size_t)b * n * n, n, h + (size_t)b * n, A + (size_t)b * m * n, m, Alb + (size_t)b * m, Aub + (size_t)b * m,
               xlb + (size_t)b * n, xub + (size_t)b * n, x0 ? x0 + (size_t)b * n : nullptr, y0 ? y0 + (size_t)b * (m + n) : nullptr, s, qi);
    const int ln = lane_id();
    for (int i = ln; i < n; i += WAVE) x[(size_t)b * n + i] = w.x[i];
    for (int i = ln; i < m + n; i += WAVE) y[(size_t)b * (m + n) + i] = w.y[i];
    if (ln == 0) info[b] = qi;
}

}  // namespace pmpc
