// ORACLE — TEST INFRASTRUCTURE ONLY (CPU restatement of the reference algorithm).
//
// Ruiz equilibration of a dense QP (H n x n, A m x n, column-major), in place.
// Follows /root/reference/src/solvers/qp_preconditioners.hpp: RuizEquilibration::compute<DENSE> :160-233 (at most 4
// sweeps; loop condition "(1 - scaling_norm) >= 1e-3" with scaling_norm = largest row/column infinity norm BEFORE the
// sweep's scaling, so a badly scaled problem leaves the loop after ONE sweep — restated as written), scale :352-357,
// unscale(x, y) :359-364, unscale(H, h, A, ...) :367-383. Call site: SQPBase::solve, sqp_base.hpp:605-611 / :661-665.
// Eigen expressions are restated coefficient by coefficient in their association order:
//   D.asDiagonal() * H * D.asDiagonal()                ->  (D_i * H_ij) * D_j
//   (1/c) * Dinv.asDiagonal() * H * Dinv.asDiagonal()  ->  (((1/c) * Dinv_i) * H_ij) * Dinv_j
//   m_D.mean()                                         ->  sequential sum / n  (Eigen's packet reduction order may differ in
//                                                          the last bit; no known-answer test resolves that)
// Pinned by tests/solvers/qp/box_admm_test.cpp:47-83 (tests/test_oracle_pins.py).
#pragma once
#include <algorithm>
#include <cmath>
#include <limits>
#include <vector>

namespace oracle {

struct Ruiz {
    int n = 0, m = 0;
    std::vector<double> D, E;   // accumulated scalings (qp_preconditioners.hpp:153-154)
    double c = 1.0;             // cost scaling m_c

    Ruiz(int n_, int m_) : n(n_), m(m_), D(n_, 1.0), E(m_, 1.0) {}

    void compute(double* H, double* h, double* A, double* Al, double* Au, double* l, double* u) {   // :160-233
        const int max_iter = 4;
        c = 1.0;
        std::fill(D.begin(), D.end(), 1.0); std::fill(E.begin(), E.end(), 1.0);
        std::vector<double> mD(n), mE(m);
        const double approx_zero = std::numeric_limits<double>::epsilon();
        const double tolerance = 1e-3;
        double scaling_norm = 10 * tolerance;
        int iter = 0;
        while (iter < max_iter && (1.0 - scaling_norm) >= tolerance) {
            for (int i = 0; i < m; ++i) { double r = 0; for (int j = 0; j < n; ++j) r = std::fmax(r, std::fabs(A[i + j * m])); mE[i] = r; }
            for (int j = 0; j < n; ++j) {
                double r = 0; for (int i = 0; i < n; ++i) r = std::fmax(r, std::fabs(H[i + j * n]));
                double x = 0; for (int i = 0; i < m; ++i) x = std::fmax(x, std::fabs(A[i + j * m]));
                mD[j] = std::fmax(r, x);
            }
            double mx = mD[0];   // maxCoeff
            for (int j = 1; j < n; ++j) mx = std::max(mx, mD[j]);
            if (m > 0) { double me = mE[0]; for (int i = 1; i < m; ++i) me = std::max(me, mE[i]); mx = std::max(mx, me); }
            scaling_norm = mx;
            for (int k = 0; k < n; ++k) if (mD[k] < approx_zero) mD[k] = 1.0;
            for (int k = 0; k < m; ++k) if (mE[k] < approx_zero) mE[k] = 1.0;
            for (int k = 0; k < n; ++k) mD[k] = 1.0 / std::sqrt(mD[k]);
            for (int k = 0; k < m; ++k) mE[k] = 1.0 / std::sqrt(mE[k]);
            for (int j = 0; j < n; ++j) for (int i = 0; i < n; ++i) H[i + j * n] = (mD[i] * H[i + j * n]) * mD[j];
            for (int j = 0; j < n; ++j) for (int i = 0; i < m; ++i) A[i + j * m] = (mE[i] * A[i + j * m]) * mD[j];
            for (int k = 0; k < n; ++k) h[k] = h[k] * mD[k];
            for (int k = 0; k < n; ++k) D[k] = D[k] * mD[k];
            for (int k = 0; k < m; ++k) E[k] = E[k] * mE[k];
            double sum = 0.0;
            for (int j = 0; j < n; ++j) { double r = 0; for (int i = 0; i < n; ++i) r = std::fmax(r, std::fabs(H[i + j * n])); mD[j] = r; sum += r; }
            double h_inf = 0; for (int k = 0; k < n; ++k) h_inf = std::fmax(h_inf, std::fabs(h[k]));
            h_inf = h_inf > approx_zero ? h_inf : 1.0;
            const double gamma = 1.0 / std::max(sum / n, h_inf);
            for (int e = 0; e < n * n; ++e) H[e] *= gamma;
            for (int k = 0; k < n; ++k) h[k] *= gamma;
            c *= gamma;
            ++iter;
        }
        for (int k = 0; k < m; ++k) { Au[k] = Au[k] * E[k]; Al[k] = Al[k] * E[k]; }
        for (int k = 0; k < n; ++k) { l[k] = l[k] * (1.0 / D[k]); u[k] = u[k] * (1.0 / D[k]); }
    }
    void scale(double* x, double* y) const {   // :352-357
        for (int k = 0; k < n; ++k) x[k] = x[k] * (1.0 / D[k]);
        for (int k = 0; k < m; ++k) y[k] = c * (y[k] * (1.0 / E[k]));
        for (int k = 0; k < n; ++k) y[m + k] = c * (y[m + k] * D[k]);
    }
    void unscale(double* x, double* y) const {   // :359-364
        for (int k = 0; k < n; ++k) x[k] = x[k] * D[k];
        for (int k = 0; k < m; ++k) y[k] = (1 / c) * (y[k] * E[k]);
        for (int k = 0; k < n; ++k) y[m + k] = (1 / c) * (y[m + k] * (1.0 / D[k]));
    }
    void unscale(double* H, double* h, double* A, double* Al, double* Au, double* l, double* u) const {   // :367-383
        const double ic = 1 / c;
        for (int j = 0; j < n; ++j) for (int i = 0; i < n; ++i) H[i + j * n] = ((ic * (1.0 / D[i])) * H[i + j * n]) * (1.0 / D[j]);
        for (int j = 0; j < n; ++j) for (int i = 0; i < m; ++i) A[i + j * m] = ((1.0 / E[i]) * A[i + j * m]) * (1.0 / D[j]);
        for (int k = 0; k < n; ++k) h[k] = ic * (h[k] * (1.0 / D[k]));
        for (int k = 0; k < m; ++k) { Au[k] = Au[k] * (1.0 / E[k]); Al[k] = Al[k] * (1.0 / E[k]); }
        for (int k = 0; k < n; ++k) { l[k] = l[k] * D[k]; u[k] = u[k] * D[k]; }
    }
};

}  // namespace oracle
