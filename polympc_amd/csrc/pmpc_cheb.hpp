// polympc_amd — host-side Chebyshev–Gauss–Lobatto constants (computed once per solver, staged to LDS by the kernels).
// Replaces Chebyshev<P>::compute_nodes / compute_int_weights / compute_diff_matrix
// (/root/reference/src/polynomials/ebyshev.hpp:111-117, :120-159, :198-214), Spline<Poly,S> (splines.hpp:22-46) and
// the time grid of ContinuousOCP (continuous_ocp.hpp:45-66, :147-159).
#pragma once
#include <algorithm>
#include <cmath>
#include <vector>
#include "pmpc_ocp.hpp"

namespace pmpc {

inline void cheb_nodes(int P, double* nodes) {
    for (int j = 0; j <= P; ++j) nodes[j] = std::cos(double(j) * (M_PI / P));
}

inline void cheb_weights(int P, double* w) {
    std::vector<double> theta(P + 1), v(P > 1 ? P - 1 : 0, 1.0);
    for (int j = 0; j <= P; ++j) theta[j] = double(j) * (M_PI / P);
    for (int j = 0; j <= P; ++j) w[j] = 0.0;
    const double P2 = std::pow(double(P), 2);
    if (P % 2 == 0) {
        w[0] = 1.0 / (P2 - 1); w[P] = w[0];
        for (int k = 1; k <= P / 2 - 1; ++k) {
            const double f = 2.0 / (4 * std::pow(double(k), 2) - 1);
            for (int i = 0; i < P - 1; ++i) v[i] -= f * std::cos(2 * k * theta[i + 1]);
        }
        for (int i = 0; i < P - 1; ++i) v[i] -= std::cos(P * theta[i + 1]) / (P2 - 1);
    } else {
        w[0] = 1.0 / P2; w[P] = w[0];
        for (int k = 1; k <= (P - 1) / 2; ++k) {
            const double f = 2.0 / (4 * std::pow(double(k), 2) - 1);
            for (int i = 0; i < P - 1; ++i) v[i] -= f * std::cos(2 * k * theta[i + 1]);
        }
    }
    for (int i = 0; i < P - 1; ++i) w[i + 1] = (2.0 / P) * v[i];
}

// Trefethen: c = [2,1..1,2].*(-1)^j ; Dn = (c c^-T) ./ (dX + I) ; D = Dn - diag(rowsum Dn). Column-major (P+1)x(P+1).
inline void cheb_diff_matrix(int P, double* D) {
    const int n = P + 1;
    std::vector<double> x(n), c(n, 1.0), Dn(n * n);
    cheb_nodes(P, x.data());
    c[0] = 2.0; c[P] = 2.0;
    for (int j = 0; j < n; ++j) c[j] = std::pow(-1.0, double(j)) * c[j];
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) Dn[i + j * n] = (c[i] * (1.0 / c[j])) * (1.0 / ((x[i] - x[j]) + (i == j ? 1.0 : 0.0)));
    for (int i = 0; i < n * n; ++i) D[i] = Dn[i];
    for (int i = 0; i < n; ++i) {
        double rs = 0.0;
        for (int j = 0; j < n; ++j) rs += Dn[i + j * n];
        D[i + i * n] = Dn[i + i * n] - rs;
    }
}

inline bool make_cheb_data(int P, int S, double t0, double tf, ChebData& cd) {
    if (P < 1 || P > MAX_P || S < 1 || P * S + 1 > MAX_NODES) return false;
    cd.P = P; cd.S = S; cd.NN = P * S + 1; cd._pad = 0;
    cd.t_start = t0; cd.t_stop = tf; cd.t_scale = (tf - t0) / (2 * S);
    cheb_diff_matrix(P, cd.D);
    cheb_weights(P, cd.w);
    std::vector<double> nodes(P + 1);
    cheb_nodes(P, nodes.data());
    const double t_length = (tf - t0) / S, t_shift = t_length / 2;
    for (int i = 0; i < S; ++i)
        for (int j = 0; j <= P; ++j) cd.tn[i * P + j] = (t_length / 2) * nodes[P - j] + (t0 + t_shift + i * t_length) * 1.0;
    std::reverse(cd.tn, cd.tn + cd.NN);
    return true;
}

}  // namespace pmpc
