// ORACLE — TEST INFRASTRUCTURE ONLY (CPU restatement of the reference algorithm).
//
// boxADMM<N, M, float>: the single-precision instantiation the reference tests in tests/solvers/qp/box_admm_test.cpp:85-115
// (box_admmSinglePrecisionFloat). Same algorithm as oracle/qp.hpp (box_admm.hpp:88-205 and the helpers it cites there), every
// quantity a `float` as in the reference's templates: the settings are qp_solver_settings_t<float> (qp_base.hpp:17-53), the constants
// `static constexpr scalar_t` (qp_base.hpp:124-126, box_admm.hpp:56-59) and DIV_BY_ZERO_REGUL = regulariser<float>::value = 10e-5
// (qp_base.hpp:84-86). Two factorisation orders, as in qp.hpp:
//   PIVOT_EIGEN  : Eigen::LDLT<Matrix<float,...>, Lower> restated (max-|diag| pivoting, left-looking update, D^+ solve) — ties the
//                  restatement to the reference's fixture (tests/test_oracle_pins.py)
//   PIVOT_STATIC : no permutation, right-looking fma updates, column-oriented substitutions — the order of the HIP kernel
//                  (polympc_amd/csrc/pmpc_qp_f32.hip), so that the GPU can be checked bit for bit.
// Parity pin: the reference's own float fixture (solution within 1e-2 of (0.3, 0.7), SOLVED, iter < 150) under both orders, and agreement with
// the double restatement to single-precision accuracy on random QPs.
#pragma once
#include <algorithm>
#include <cmath>
#include <limits>
#include <vector>
#include "qp.hpp"

namespace oracle {

struct LDLTf {
    int n = 0;
    pivot_policy policy = PIVOT_EIGEN;
    std::vector<float> M;
    std::vector<int> tr;
    std::vector<float> temp;

    void compute(const std::vector<float>& K, int n_, pivot_policy pol) {
        n = n_; policy = pol; M = K; tr.assign(n, 0); temp.assign(n, 0.0f);
        auto at = [&](int i, int j) -> float& { return M[i + j * n]; };
        if (policy == PIVOT_STATIC) {   // qp.hpp LDLT::compute_static in float
            for (int k = 0; k < n; ++k) tr[k] = k;
            std::vector<float> col(n);
            for (int k = 0; k < n; ++k) {
                const float dk = at(k, k);
                for (int i = k + 1; i < n; ++i) { col[i] = at(i, k); at(i, k) = col[i] / dk; }
                for (int j = k + 1; j < n; ++j) {
                    const float ljk = at(j, k);
                    for (int i = j; i < n; ++i) at(i, j) = std::fma(-col[i], ljk, at(i, j));
                }
            }
            return;
        }
        for (int k = 0; k < n; ++k) {   // qp.hpp LDLT::compute (Eigen semantics) in float
            int big = k; float bv = std::fabs(at(k, k));
            for (int i = k + 1; i < n; ++i) { float v = std::fabs(at(i, i)); if (v > bv) { bv = v; big = i; } }
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
                float acc = 0.0f;
                for (int j = 0; j < k; ++j) acc += at(k, j) * temp[j];
                at(k, k) -= acc;
                for (int i = k + 1; i < n; ++i) {
                    float a = 0.0f;
                    for (int j = 0; j < k; ++j) a += at(i, j) * temp[j];
                    at(i, k) -= a;
                }
            }
            const float akk = at(k, k);
            const bool valid = std::fabs(akk) > 0.0f;
            if (k == 0 && !valid) { for (int j = 0; j < n; ++j) tr[j] = j; return; }
            if (rs > 0 && valid) for (int i = k + 1; i < n; ++i) at(i, k) /= akk;
        }
    }

    void solve(const float* b, float* x) const {
        auto at = [&](int i, int j) -> float { return M[i + j * n]; };
        for (int i = 0; i < n; ++i) x[i] = b[i];
        if (policy == PIVOT_STATIC) {
            for (int j = 0; j < n; ++j) for (int i = j + 1; i < n; ++i) x[i] = std::fma(-at(i, j), x[j], x[i]);
            for (int i = 0; i < n; ++i) x[i] = x[i] / at(i, i);
            for (int j = n - 1; j >= 0; --j) for (int i = j - 1; i >= 0; --i) x[i] = std::fma(-at(j, i), x[j], x[i]);
            return;
        }
        for (int k = 0; k < n; ++k) if (tr[k] != k) std::swap(x[k], x[tr[k]]);
        for (int i = 0; i < n; ++i) { float a = x[i]; for (int j = 0; j < i; ++j) a -= at(i, j) * x[j]; x[i] = a; }
        const float tol = 1.0f / std::numeric_limits<float>::max();
        for (int i = 0; i < n; ++i) { if (std::fabs(at(i, i)) > tol) x[i] /= at(i, i); else x[i] = 0.0f; }
        for (int i = n - 1; i >= 0; --i) { float a = x[i]; for (int j = i + 1; j < n; ++j) a -= at(j, i) * x[j]; x[i] = a; }
        for (int k = n - 1; k >= 0; --k) if (tr[k] != k) std::swap(x[k], x[tr[k]]);
    }
};

struct qp_settings_f {   // qp_solver_settings_t<float>, ADMM-related subset
    float eps_rel = 1e-3f, eps_abs = 1e-3f;
    int max_iter = 1000;
    float rho = 1e-1f, sigma = 1e-6f, alpha = 1.0f;
    int check_termination = 25;
    bool adaptive_rho = false;
    float adaptive_rho_tolerance = 5;
    int adaptive_rho_interval = 25;
};
struct qp_info_f {
    int status = QP_UNINITIALIZED, iter = 0, rho_updates = 0;
    float rho_estimate = 0, res_prim = 1, res_dual = 1;
};

struct BoxADMMf {
    static constexpr float RHO_MIN = 1e-6f, RHO_MAX = 1e+6f, RHO_EQ_FACTOR = 1e+3f;   // box_admm.hpp:56-59
    static constexpr float LOOSE_BOUNDS_THRESH = 1e+10f, EQ_TOL = 1e-4f;              // qp_base.hpp:124-125
    static constexpr float DIV_BY_ZERO_REGUL = (float)10e-5;                          // regulariser<float>, qp_base.hpp:84-86
    enum ctype { INEQUALITY_CONSTRAINT = 0, EQUALITY_CONSTRAINT = 1, LOOSE_BOUNDS = 2 };

    int N, M;
    qp_settings_f settings;
    qp_info_f info;
    pivot_policy pivot = PIVOT_EIGEN;
    std::vector<float> x, y, x_tilde, q, z, z_tilde, z_prev, rho_vec, rho_inv_vec, rho_box, rho_box_inv, rho_box_prev, K;
    std::vector<int> constr_type, box_type;
    LDLTf ldlt;
    float rho = 0, max_Ax_z_norm = 0, max_Hx_ATy_h_norm = 0;
    int iter = 0;

    BoxADMMf(int n, int m) : N(n), M(m) {
        x.assign(N, 0); y.assign(N + M, 0); x_tilde.assign(N, 0); q.assign(N, 0);
        z.assign(M, 0); z_tilde.assign(M, 0); z_prev.assign(M, 0);
        rho_vec.assign(M, 0); rho_inv_vec.assign(M, 0);
        rho_box.assign(N, 0); rho_box_inv.assign(N, 0); rho_box_prev.assign(N, 0);
        constr_type.assign(M, 0); box_type.assign(N, 0);
        K.assign((size_t)(N + M) * (N + M), 0.0f);
    }
    static float inf_norm(const float* v, int n) { float r = 0; for (int i = 0; i < n; ++i) r = std::fmax(r, std::fabs(v[i])); return r; }
    static int classify(float lb, float ub) {
        if (lb < -LOOSE_BOUNDS_THRESH && ub > LOOSE_BOUNDS_THRESH) return LOOSE_BOUNDS;
        if (ub - lb < EQ_TOL) return EQUALITY_CONSTRAINT;
        return INEQUALITY_CONSTRAINT;
    }
    static float rho_of(int type, float rho0) { return type == LOOSE_BOUNDS ? RHO_MIN : (type == EQUALITY_CONSTRAINT ? RHO_EQ_FACTOR * rho0 : rho0); }
    void rho_vec_update(float rho0) {
        for (int i = 0; i < M; ++i) { rho_vec[i] = rho_of(constr_type[i], rho0); rho_inv_vec[i] = 1.0f / rho_vec[i]; }
        rho = rho0;
        for (int i = 0; i < N; ++i) { rho_box[i] = rho_of(box_type[i], rho0); rho_box_inv[i] = 1.0f / rho_box[i]; }
        info.rho_updates += 1;
    }
    void matvec(const float* A, int rows, int cols, const float* v, float* out) const {
        for (int i = 0; i < rows; ++i) { float a = 0; for (int j = 0; j < cols; ++j) a += A[i + j * rows] * v[j]; out[i] = a; }
    }
    void matTvec(const float* A, int rows, int cols, const float* v, float* out) const {
        for (int j = 0; j < cols; ++j) { float a = 0; for (int i = 0; i < rows; ++i) a += A[i + j * rows] * v[i]; out[j] = a; }
    }
    void residuals_update(const float* H, const float* h, const float* A) {
        std::vector<float> Ax(M), Hx(N), ATy(N);
        matvec(A, M, N, x.data(), Ax.data());
        max_Ax_z_norm = std::fmax(inf_norm(Ax.data(), M), std::fmax(inf_norm(z.data(), M), inf_norm(x.data(), N)));
        matvec(H, N, N, x.data(), Hx.data());
        matTvec(A, M, N, y.data(), ATy.data());
        max_Hx_ATy_h_norm = std::fmax(inf_norm(Hx.data(), N), std::fmax(inf_norm(ATy.data(), N), std::fmax(inf_norm(h, N), inf_norm(y.data() + M, N))));
        float rp = 0, rq = 0, rd = 0;
        for (int i = 0; i < M; ++i) rp = std::fmax(rp, std::fabs(Ax[i] - z[i]));
        for (int i = 0; i < N; ++i) rq = std::fmax(rq, std::fabs(x[i] - q[i]));
        info.res_prim = rp + rq;
        for (int i = 0; i < N; ++i) rd = std::fmax(rd, std::fabs(((Hx[i] + h[i]) + ATy[i]) + y[M + i]));
        info.res_dual = rd;
    }
    bool termination_criteria() const {
        return info.res_prim <= settings.eps_abs + settings.eps_rel * max_Ax_z_norm && info.res_dual <= settings.eps_abs + settings.eps_rel * max_Hx_ATy_h_norm;
    }
    float estimate_rho(float rho0) const {
        const float rp = info.res_prim / (max_Ax_z_norm + DIV_BY_ZERO_REGUL);
        const float rd = info.res_dual / (max_Hx_ATy_h_norm + DIV_BY_ZERO_REGUL);
        return rho0 * std::sqrt(rp / (rd + DIV_BY_ZERO_REGUL));
    }
    void construct_kkt(const float* H, const float* A) {
        const int NM = N + M;
        std::fill(K.begin(), K.end(), 0.0f);
        for (int j = 0; j < N; ++j) for (int i = 0; i < N; ++i) K[i + j * NM] = H[i + j * N];
        for (int i = 0; i < N; ++i) K[i + i * NM] += settings.sigma;
        for (int i = 0; i < N; ++i) K[i + i * NM] += rho_box[i];
        for (int j = 0; j < N; ++j) for (int i = 0; i < M; ++i) K[(N + i) + j * NM] = A[i + j * M];
        for (int i = 0; i < M; ++i) K[(N + i) + (N + i) * NM] = -rho_inv_vec[i];
    }
    void update_kkt_rho() {
        const int NM = N + M;
        for (int i = 0; i < N; ++i) K[i + i * NM] += (rho_box[i] - rho_box_prev[i]);
        for (int i = 0; i < M; ++i) K[(N + i) + (N + i) * NM] = -rho_inv_vec[i];
    }

    int solve(const float* H, const float* h, const float* A, const float* Alb, const float* Aub, const float* xlb, const float* xub,
              const float* x_guess, const float* y_guess) {   // null guesses: the 7-argument form (zeros)
        const int NM = N + M;
        std::vector<float> rhs(NM), sol(NM);
        for (int i = 0; i < N; ++i) x[i] = x_guess ? x_guess[i] : 0.0f;
        for (int i = 0; i < NM; ++i) y[i] = y_guess ? y_guess[i] : 0.0f;
        matvec(A, M, N, x.data(), z.data());
        for (int i = 0; i < N; ++i) q[i] = x[i];
        for (int i = 0; i < M; ++i) constr_type[i] = classify(Alb[i], Aub[i]);
        for (int i = 0; i < N; ++i) box_type[i] = classify(xlb[i], xub[i]);
        rho_vec_update(settings.rho);
        construct_kkt(H, A);
        ldlt.compute(K, NM, pivot);
        info.status = QP_UNSOLVED;
        const float alpha = settings.alpha;
        bool check_termination = false;
        for (iter = 1; iter <= settings.max_iter; iter++) {
            z_prev = z;
            for (int i = 0; i < N; ++i) rhs[i] = ((settings.sigma * x[i] - h[i]) + rho_box[i] * q[i]) - y[M + i];
            for (int i = 0; i < M; ++i) rhs[N + i] = z[i] - rho_inv_vec[i] * y[i];
            ldlt.solve(rhs.data(), sol.data());
            for (int i = 0; i < N; ++i) x_tilde[i] = sol[i];
            for (int i = 0; i < M; ++i) z_tilde[i] = z_prev[i] + rho_inv_vec[i] * (sol[N + i] - y[i]);
            for (int i = 0; i < N; ++i) { x[i] = alpha * x_tilde[i]; x[i] += (1 - alpha) * x[i]; }   // quirk Q1 (:129-130)
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
                float new_rho = estimate_rho(rho);
                new_rho = std::fmax(RHO_MIN, std::fmin(new_rho, RHO_MAX));
                info.rho_estimate = new_rho;
                if (new_rho < rho / settings.adaptive_rho_tolerance || new_rho > rho * settings.adaptive_rho_tolerance) {
                    rho_box_prev = rho_box;
                    rho_vec_update(new_rho);
                    update_kkt_rho();
                    ldlt.compute(K, NM, pivot);
                }
            }
        }
        if (iter > settings.max_iter) info.status = QP_MAX_ITER_EXCEEDED;
        info.iter = iter;
        return info.status;
    }
};

// ADMM<N, M, float> (admm.hpp, OSQP form: box constraints stacked under the general ones; tests/solvers/qp/admm_solver_test.cpp:84-113): oracle/admm.hpp in float
struct ADMMf {
    int N, M, ME;
    qp_settings_f settings;
    qp_info_f info;
    pivot_policy pivot = PIVOT_EIGEN;
    std::vector<float> x, y, z, z_tilde, z_prev, x_tilde, rho_vec, rho_inv_vec, K;
    std::vector<int> ctype;
    LDLTf ldlt;
    float rho = 0, max_Ax_z_norm = 0, max_Hx_ATy_h_norm = 0;
    int iter = 0;

    ADMMf(int n, int m) : N(n), M(m), ME(n + m) {
        x.assign(N, 0); x_tilde.assign(N, 0); y.assign(ME, 0); z.assign(ME, 0); z_tilde.assign(ME, 0); z_prev.assign(ME, 0);
        rho_vec.assign(ME, 0); rho_inv_vec.assign(ME, 0); ctype.assign(ME, 0);
        K.assign((size_t)(N + ME) * (N + ME), 0.0f);
    }
    void rho_vec_update(float rho0) {   // admm.hpp:405-440
        for (int i = 0; i < ME; ++i) { rho_vec[i] = BoxADMMf::rho_of(ctype[i], rho0); rho_inv_vec[i] = 1.0f / rho_vec[i]; }
        rho = rho0;
        info.rho_updates += 1;
    }
    void construct_kkt(const float* H, const float* A) {   // :249-263
        const int NM = N + ME;
        std::fill(K.begin(), K.end(), 0.0f);
        for (int j = 0; j < N; ++j) for (int i = 0; i < N; ++i) K[i + (size_t)j * NM] = H[i + j * N];
        for (int i = 0; i < N; ++i) K[i + (size_t)i * NM] += settings.sigma;
        for (int j = 0; j < N; ++j) for (int i = 0; i < M; ++i) K[(N + i) + (size_t)j * NM] = A[i + j * M];
        for (int i = 0; i < N; ++i) K[(N + M + i) + (size_t)i * NM] = 1.0f;
        for (int i = 0; i < ME; ++i) K[(N + i) + (size_t)(N + i) * NM] = -1.0f * rho_inv_vec[i];
    }
    void update_kkt_rho() { const int NM = N + ME; for (int i = 0; i < ME; ++i) K[(N + i) + (size_t)(N + i) * NM] = -rho_inv_vec[i]; }   // :490-494
    void residuals_update(const float* H, const float* h, const float* A) {   // :442-462
        std::vector<float> Ax(M), Hx(N), ATy(N);
        for (int i = 0; i < M; ++i) { float a = 0; for (int j = 0; j < N; ++j) a += A[i + j * M] * x[j]; Ax[i] = a; }
        float norm_Ax = BoxADMMf::inf_norm(Ax.data(), M);
        norm_Ax = std::fmax(norm_Ax, BoxADMMf::inf_norm(x.data(), N));
        max_Ax_z_norm = std::fmax(norm_Ax, BoxADMMf::inf_norm(z.data(), ME));
        for (int i = 0; i < N; ++i) { float a = 0; for (int j = 0; j < N; ++j) a += H[i + j * N] * x[j]; Hx[i] = a; }
        for (int j = 0; j < N; ++j) { float a = 0; for (int i = 0; i < M; ++i) a += A[i + j * M] * y[i]; ATy[j] = a; }
        max_Hx_ATy_h_norm = std::fmax(BoxADMMf::inf_norm(Hx.data(), N), std::fmax(BoxADMMf::inf_norm(ATy.data(), N),
                                      std::fmax(BoxADMMf::inf_norm(h, N), BoxADMMf::inf_norm(y.data() + M, N))));
        float rp = 0, rb = 0, rd = 0;
        for (int i = 0; i < M; ++i) rp = std::fmax(rp, std::fabs(Ax[i] - z[i]));
        for (int i = 0; i < N; ++i) rb = std::fmax(rb, std::fabs(x[i] - z[M + i]));
        info.res_prim = std::fmax(rp, rb);
        for (int i = 0; i < N; ++i) rd = std::fmax(rd, std::fabs(((Hx[i] + h[i]) + ATy[i]) + y[M + i]));
        info.res_dual = rd;
    }
    int solve(const float* H, const float* h, const float* A, const float* Alb, const float* Aub, const float* xl, const float* xu,
              const float* x_guess, const float* y_guess) {   // :112-212
        const int NM = N + ME;
        std::vector<float> rhs(NM), sol(NM);
        for (int i = 0; i < N; ++i) x[i] = x_guess ? x_guess[i] : 0.0f;
        for (int i = 0; i < ME; ++i) y[i] = y_guess ? y_guess[i] : 0.0f;
        for (int i = 0; i < M; ++i) { float a = 0; for (int j = 0; j < N; ++j) a += A[i + j * M] * x[j]; z[i] = a; }
        for (int i = 0; i < N; ++i) z[M + i] = x[i];
        for (int i = 0; i < M; ++i) ctype[i] = BoxADMMf::classify(Alb[i], Aub[i]);
        for (int i = 0; i < N; ++i) ctype[M + i] = BoxADMMf::classify(xl[i], xu[i]);
        rho_vec_update(settings.rho);
        construct_kkt(H, A);
        ldlt.compute(K, NM, pivot);
        info.status = QP_UNSOLVED;
        const float alpha = settings.alpha;
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
                const float lo = i < M ? Alb[i] : xl[i - M], hi = i < M ? Aub[i] : xu[i - M];
                z[i] = std::fmin(std::fmax(z[i], lo), hi);
            }
            for (int i = 0; i < ME; ++i) y[i] += rho_vec[i] * ((alpha * z_tilde[i] + (1 - alpha) * z_prev[i]) - z[i]);
            const bool check = (settings.check_termination != 0 && iter % settings.check_termination == 0);
            if (check) {
                residuals_update(H, h, A);
                if (info.res_prim <= settings.eps_abs + settings.eps_rel * max_Ax_z_norm && info.res_dual <= settings.eps_abs + settings.eps_rel * max_Hx_ATy_h_norm) { info.status = QP_SOLVED; break; }
            }
            if (settings.adaptive_rho && iter % settings.adaptive_rho_interval == 0) {
                if (!check) residuals_update(H, h, A);
                const float rp = info.res_prim / (max_Ax_z_norm + BoxADMMf::DIV_BY_ZERO_REGUL), rd = info.res_dual / (max_Hx_ATy_h_norm + BoxADMMf::DIV_BY_ZERO_REGUL);
                float new_rho = rho * std::sqrt(rp / (rd + BoxADMMf::DIV_BY_ZERO_REGUL));
                new_rho = std::fmax(BoxADMMf::RHO_MIN, std::fmin(new_rho, BoxADMMf::RHO_MAX));
                info.rho_estimate = new_rho;
                if (new_rho < rho / settings.adaptive_rho_tolerance || new_rho > rho * settings.adaptive_rho_tolerance) {
                    rho_vec_update(new_rho);
                    update_kkt_rho();
                    ldlt.compute(K, NM, pivot);
                }
            }
        }
        if (iter > settings.max_iter) info.status = QP_MAX_ITER_EXCEEDED;
        info.iter = iter;
        return info.status;
    }
};

}  // namespace oracle
