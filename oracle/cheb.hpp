// ORACLE — TEST INFRASTRUCTURE ONLY (CPU restatement of the reference algorithm).
//
// Chebyshev–Gauss–Lobatto collocation constants.
// Follows /root/reference/src/polynomials/ebyshev.hpp:
//   compute_nodes        :111-117   nodes_j = cos(pi*j/P)
//   compute_int_weights  :120-159   Clenshaw–Curtis weights (even / odd P branches)
//   compute_diff_matrix  :198-214   Trefethen's differentiation matrix
// and the Spline<Poly,S> tag of src/polynomials/splines.hpp:22-46 (NUM_NODES = P*S+1).
#pragma once
#include <cmath>
#include <vector>

namespace oracle {

struct Chebyshev {
    int P;
    std::vector<double> nodes;    // P+1
    std::vector<double> weights;  // P+1 (Clenshaw–Curtis)
    std::vector<double> D;        // (P+1)x(P+1), column-major: D(i,j) = D[i + j*(P+1)]

    explicit Chebyshev(int P_) : P(P_) {
        compute_nodes();
        compute_int_weights();
        compute_diff_matrix();
    }
    double Dij(int i, int j) const { return D[i + j * (P + 1)]; }

    // ebyshev.hpp:111-117
    void compute_nodes() {
        nodes.resize(P + 1);
        for (int j = 0; j <= P; ++j) nodes[j] = std::cos(double(j) * (M_PI / P));
    }

    // ebyshev.hpp:120-159
    void compute_int_weights() {
        std::vector<double> theta(P + 1);
        for (int j = 0; j <= P; ++j) theta[j] = double(j) * (M_PI / P);
        weights.assign(P + 1, 0.0);
        std::vector<double> v(P > 1 ? P - 1 : 0, 1.0);
        if (P % 2 == 0) {
            weights[0] = 1.0 / (std::pow(double(P), 2) - 1);
            weights[P] = weights[0];
            for (int k = 1; k <= P / 2 - 1; ++k)
                for (int i = 0; i < P - 1; ++i)
                    v[i] -= (2.0 / (4 * std::pow(double(k), 2) - 1)) * std::cos(2 * k * theta[i + 1]);
            for (int i = 0; i < P - 1; ++i)
                v[i] -= std::cos(P * theta[i + 1]) / (std::pow(double(P), 2) - 1);
        } else {
            weights[0] = 1.0 / std::pow(double(P), 2);
            weights[P] = weights[0];
            for (int k = 1; k <= (P - 1) / 2; ++k)
                for (int i = 0; i < P - 1; ++i)
                    v[i] -= (2.0 / (4 * std::pow(double(k), 2) - 1)) * std::cos(2 * k * theta[i + 1]);
        }
        for (int i = 0; i < P - 1; ++i) weights[i + 1] = (2.0 / P) * v[i];
    }

    // ebyshev.hpp:198-214
    void compute_diff_matrix() {
        const int n = P + 1;
        std::vector<double> c(n, 1.0);
        c[0] = 2.0; c[P] = 2.0;
        for (int j = 0; j < n; ++j) c[j] = std::pow(-1.0, double(j)) * c[j];
        std::vector<double> Dn(n * n);
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) {
                double dX = nodes[i] - nodes[j];
                double eye = (i == j) ? 1.0 : 0.0;
                Dn[i + j * n] = (c[i] * (1.0 / c[j])) * (1.0 / (dX + eye));
            }
        D = Dn;
        for (int i = 0; i < n; ++i) {
            double rs = 0.0;
            for (int j = 0; j < n; ++j) rs += Dn[i + j * n];
            D[i + i * n] = Dn[i + i * n] - rs;
        }
    }
};

}  // namespace oracle
