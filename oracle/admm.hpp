// ORACLE — TEST INFRASTRUCTURE ONLY (CPU restatement of the reference algorithm).
//
// OSQP-style ADMM QP solver of the reference: /root/reference/src/solvers/admm.hpp. Same problem as boxADMM
// (min 1/2 x'Hx + h'x, Alb <= Ax <= Aub, xl <= x <= xu) but the box constraints are stacked under the general ones:
// A_e = [A ; I] (construct_A :215-222), z, y, rho of size M+N, one (2N+M) x (2N+M) quasi-definite KKT matrix
// [H + sigma I, A_e' ; A_e, -diag(1/rho)] (construct_kkt_matrix :249-263), no q / y_box split and NOT boxADMM's quirk Q1
// (x = alpha*x_tilde + (1-alpha)*x, :152). solve_impl :112-212, compute_kkt_rhs :390-394, box_projection :397-403,
// rho_vec_update :405-440, residuals_update :442-462, eps / termination / estimate_rho :464-488, update_kkt_rho :490-494.
// Linear solves go through the same LDLT restatement and pivot policies as qp.hpp.
// Pinned by tests/solvers/qp/admm_solver_test.cpp (tests/test_oracle_pins.py).
#pragma once
#include "qp.hpp"

namespace oracle {

struct ADMM {
    int N, M, ME;   // ME = M + N constraint rows of A_e
    qp_settings settings;
    qp_info info;
    pivot_policy pivot = PIVOT_EIGEN;
    std::vector<double> x, y, z, z_tilde, z_prev, x_tilde, rho_vec, rho_inv_vec, K;
    std::vector<int> ctype;
    LDLT ldlt;
    double rho = 0, max_Ax_z_norm = 0, max_Hx_ATy_h_norm = 0;
    int iter = 0;

    ADMM(int n, int m) : N(n), M(m), ME(n + m) {
        x.assign(N, 0); x_tilde.assign(N, 0); y.assign(ME, 0); z.assign(ME, 0); z_tilde.assign(ME, 0); z_prev.assign(ME, 0);
        rho_vec.assign(ME, 0); rho_inv_vec.assign(ME, 0); ctype.assign(ME, 0);
        K.assign((size_t)(N + ME) * (N + ME), 0.0);
    }
    static double inf_norm(const double* v, int n) { double r = 0; for (int i = 0; i < n; ++i) r = std::fmax(r, std::fabs(v[i])); return r; }

    void rho_vec_update(double rho0) {   // :405-440
        for (int i = 0; i < ME; ++i) {
            switch (ctype[i]) {
                case BoxADMM::LOOSE_BOUNDS: rho_vec[i] = BoxADMM::RHO_MIN; break;
                case BoxADMM::EQUALITY_CONSTRAINT: rho_vec[i] = BoxADMM::RHO_EQ_FACTOR * rho0; break;
                default: rho_vec[i] = rho0;
            }
            rho_inv_vec[i] = 1.0 / rho_vec[i];
        }
        rho = rho0;
        info.rho_updates += 1;
    }
    void construct_kkt(const double* H, const double* A) {   // :249-263 (lower triangle)
        const int NM = N + ME;
        std::fill(K.begin(), K.end(), 0.0);
        for (int j = 0; j < N; ++j) for (int i = 0; i < N; ++i) K[i + (size_t)j * NM] = H[i + j * N];
        for (int i = 0; i < N; ++i) K[i + (size_t)i * NM] += settings.sigma;
        for (int j = 0; j < N; ++j) for (int i = 0; i < M; ++i) K[(N + i) + (size_t)j * NM] = A[i + j * M];
        for (int i = 0; i < N; ++i) K[(N + M + i) + (size_t)i * NM] = 1.0;
        for (int i = 0; i < ME; ++i) K[(N + i) + (size_t)(N + i) * NM] = -1.0 * rho_inv_vec[i];
    }
    void update_kkt_rho() { const int NM = N + ME; for (int i = 0; i < ME; ++i) K[(N + i) + (size_t)(N + i) * NM] = -rho_inv_vec[i]; }   // :490-494
    void factorise() { ldlt.compute(K, N + ME, pivot); }

    void residuals_update(const double* H, const double* h, const double* A) {   // :442-462
        std::vector<double> Ax(M), Hx(N), ATy(N);
        for (int i = 0; i < M; ++i) { double a = 0; for (int j = 0; j < N; ++j) a += A[i + j * M] * x[j]; Ax[i] = a; }
        double norm_Ax = inf_norm(Ax.data(), M);
        norm_Ax = std::fmax(norm_Ax, inf_norm(x.data(), N));
        const double norm_z = inf_norm(z.data(), ME);
        max_Ax_z_norm = std::fmax(norm_Ax, norm_z);
        for (int i = 0; i < N; ++i) { double a = 0; for (int j = 0; j < N; ++j) a += H[i + j * N] * x[j]; Hx[i] = a; }
        for (int j = 0; j < N; ++j) { double a = 0; for (int i = 0; i < M; ++i) a += A[i + j * M] * y[i]; ATy[j] = a; }
        const double norm_Hx = inf_norm(Hx.data(), N), norm_ATy = inf_norm(ATy.data(), N), norm_h = inf_norm(h, N), norm_y_box = inf_norm(y.data() + M, N);
        max_Hx_ATy_h_norm = std::fmax(norm_Hx, std::fmax(norm_ATy, std::fmax(norm_h, norm_y_box)));
        double rp = 0, rb = 0, rd = 0;
        for (int i = 0; i < M; ++i) rp = std::fmax(rp, std::fabs(Ax[i] - z[i]));              // primal_residual, qp_base.hpp:224-230
        for (int i = 0; i < N; ++i) rb = std::fmax(rb, std::fabs(x[i] - z[M + i]));
        info.res_prim = std::fmax(rp, rb);
        for (int i = 0; i < N; ++i) rd = std::fmax(rd, std::fabs(((Hx[i] + h[i]) + ATy[i]) + y[M + i]));   // dual_residual, qp_base.hpp:240-252
        info.res_dual = rd;
    }
    bool termination_criteria() const {
        return info.res_prim <= settings.eps_abs + settings.eps_rel * max_Ax_z_norm && info.res_dual <= settings.eps_abs + settings.eps_rel * max_Hx_ATy_h_norm;
    }
    double estimate_rho(double rho0) const {
        const double rp = info.res_prim / (max_Ax_z_norm + BoxADMM::DIV_BY_ZERO_REGUL), rd = info.res_dual / (max_Hx_ATy_h_norm + BoxADMM::DIV_BY_ZERO_REGUL);
        return rho0 * std::sqrt(rp / (rd + BoxADMM::DIV_BY_ZERO_REGUL));
    }

    int solve(const double* H, const double* h, const double* A, const double* Alb, const double* Aub, const double* xl, const double* xu,
              const double* x_guess, const double* y_guess) {   // :112-212 (guesses NULL: the 7-argument form, zeros)
        const int NM = N + ME;
        std::vector<double> rhs(NM), sol(NM);
        for (int i = 0; i < N; ++i) x[i] = x_guess ? x_guess[i] : 0.0;
        for (int i = 0; i < ME; ++i) y[i] = y_guess ? y_guess[i] : 0.0;
        for (int i = 0; i < M; ++i) { double a = 0; for (int j = 0; j < N; ++j) a += A[i + j * M] * x[j]; z[i] = a; }   // z = A_e x_guess
        for (int i = 0; i < N; ++i) z[M + i] = x[i];
        for (int i = 0; i < M; ++i) ctype[i] = BoxADMM::classify(Alb[i], Aub[i]);
        for (int i = 0; i < N; ++i) ctype[M + i] = BoxADMM::classify(xl[i], xu[i]);
        rho_vec_update(settings.rho);
        construct_kkt(H, A);
        factorise();
        info.status = QP_UNSOLVED;
        const double alpha = settings.alpha;
        for (iter = 1; iter <= settings.max_iter; iter++) {
            z_prev = z;
            for (int i = 0; i < N; ++i) rhs[i] = settings.sigma * x[i] - h[i];
            for (int i = 0; i < ME; ++i) rhs[N + i] = z[i] - rho_inv_vec[i] * y[i];
            ldlt.solve(rhs.data(), sol.data());
            for (int i = 0; i < N; ++i) x_tilde[i] = sol[i];
            for (int i = 0; i < ME; ++i) z_tilde[i] = z_prev[i] + rho_inv_vec[i] * (sol[N + i] - y[i]);
            for (int i = 0; i < N; ++i) x[i] = alpha * x_tilde[i] + (1 - alpha) * x[i];
            for (int i = 0; i < ME; ++i) {
                z[i] = alpha * z_tilde[i];
                z[i] += (1 - alpha) * z_prev[i] + rho_inv_vec[i] * y[i];
                const double lo = i < M ? Alb[i] : xl[i - M], hi = i < M ? Aub[i] : xu[i - M];
                z[i] = std::fmin(std::fmax(z[i], lo), hi);
            }
            for (int i = 0; i < ME; ++i) y[i] += rho_vec[i] * ((alpha * z_tilde[i] + (1 - alpha) * z_prev[i]) - z[i]);
            const bool check = (settings.check_termination != 0 && iter % settings.check_termination == 0);
            if (check) {
                residuals_update(H, h, A);
                if (termination_criteria()) { info.status = QP_SOLVED; break; }
            }
            if (settings.adaptive_rho && iter % settings.adaptive_rho_interval == 0) {
                if (!check) residuals_update(H, h, A);
                double new_rho = estimate_rho(rho);
                new_rho = std::fmax(BoxADMM::RHO_MIN, std::fmin(new_rho, BoxADMM::RHO_MAX));
                info.rho_estimate = new_rho;
                if (new_rho < rho / settings.adaptive_rho_tolerance || new_rho > rho * settings.adaptive_rho_tolerance) {
                    rho_vec_update(new_rho);
                    update_kkt_rho();
                    factorise();
                }
            }
        }
        if (iter > settings.max_iter) info.status = QP_MAX_ITER_EXCEEDED;
        info.iter = iter;
        return info.status;
    }
};

}  // namespace oracle
