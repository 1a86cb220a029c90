// ORACLE — TEST INFRASTRUCTURE ONLY (CPU restatement of the reference algorithm).
//
// Chebyshev pseudospectral transcription OCP -> NLP, dense members only.
// Follows /root/reference/src/control/continuous_ocp.hpp:
//   time grid                      :45-66, :147-159
//   equalities                     :739-766
//   inequalities                   :770-782
//   equalities_linearised<DENSE>   :797-878
//   _inequalities_linearised_dense :546-575
//   cost                           :1182-1207
//   cost_gradient                  :1210-1249
//   cost_gradient_hessian<DENSE>   :1256-1367   (incl. quirk Q4: Mayer pp-block into bottomLeftCorner)
//   lagrangian_gradient<DENSE>     :1960-1975
//   lagrangian_gradient_hessian<DENSE> :2100-2174
// Variable layout (:757-765): var = [x_0..x_{nn-1} | u_0..u_{nn-1} | p], node k <-> tau_k = cos(pi k/P)
// inside its segment, i.e. time DESCENDING (node 0 = t_stop).
// Dual layout: lam = [lam_eq (NX*nn) | lam_ineq (NG*nn) | lam_box (n)].
// All matrices column-major (Eigen default): M(i,j) = M[i + j*rows].
#pragma once
#include <algorithm>
#include <type_traits>
#include <vector>
#include "ad.hpp"
#include "cheb.hpp"

namespace oracle {

template <class Model>
struct ContinuousOCP {
    enum { NX = Model::NX, NU = Model::NU, NP = Model::NP, ND = Model::ND, NG = Model::NG, NDER = NX + NU + NP };
    using ad1 = Dual<double, NDER>;
    using ad2 = Dual<ad1, NDER>;

    Model model;
    Chebyshev cheb;
    int P, S, NN;  // poly order, segments, nodes
    int VARX, VARU, VAR_SIZE, NUM_EQ, NUM_INEQ, NUM_BOX, DUAL_SIZE;
    double t_start{0}, t_stop{1};
    std::vector<double> time_nodes;

    ContinuousOCP(int P_, int S_, const Model& m = Model()) : model(m), cheb(P_), P(P_), S(S_) {
        NN = P * S + 1;
        VARX = NX * NN; VARU = NU * NN;
        VAR_SIZE = VARX + VARU + NP;
        NUM_EQ = VARX; NUM_INEQ = NG * NN; NUM_BOX = VAR_SIZE;
        DUAL_SIZE = NUM_EQ + NUM_INEQ + NUM_BOX;
        time_nodes.assign(NN, 0.0);
        set_time_limits(0.0, 1.0);
    }

    // Sparsity-preserving block BFGS, continuous_ocp.hpp:2304-2431 (hessian_update_impl<SPARSE>; the update the reference's MPC
    // tests select with "this->problem.hessian_update_impl", mpc_wrapper_test.cpp:100-105), restated on dense column-major
    // storage: per node k only the (x_k, u_k) diagonal block receives the damped rank-2 update, with the GLOBAL scalars
    // s'Bs, s'y, s'r; with NP > 0 also the parameter border and corner. Coefficients in the reference's association order:
    //   (-scaling_inv * v_i) * v_j  then  += (c_inv * w_i) * w_j  (w = y, c = s'y, or the damped r, c = s'r); hes_xu = hes_ux'.
    static constexpr bool HAS_BLOCK_BFGS = true;
    void hessian_update_block(double* Hm, const double* s, const double* y) const {
        const int n = VAR_SIZE;
        std::vector<double> v(n), r(n);
        for (int i = 0; i < n; ++i) { double a = 0; for (int j = 0; j < n; ++j) a += Hm[i + j * n] * s[j]; v[i] = a; }
        double scaling = 0, sy = 0;
        for (int i = 0; i < n; ++i) scaling += s[i] * v[i];
        const double scaling_inv = 1.0 / scaling;
        for (int i = 0; i < n; ++i) sy += s[i] * y[i];
        const double sy_inv = 1.0 / sy;
        const bool plain = sy >= 0.2 * scaling;
        const double* w = y; double c_inv = sy_inv;
        if (!plain) {
            const double theta = 0.8 * scaling / (scaling - sy);
            for (int i = 0; i < n; ++i) r[i] = theta * y[i] + (1 - theta) * v[i];
            double sr = 0; for (int i = 0; i < n; ++i) sr += s[i] * r[i];
            c_inv = 1.0 / sr; w = r.data();
        }
        auto term = [&](int i, int j) { double t = (-scaling_inv * v[i]) * v[j]; t += (c_inv * w[i]) * w[j]; return t; };
        for (int k = 0; k < NN; ++k) {
            for (int j = 0; j < NX; ++j) {
                const int cj = k * NX + j;
                for (int i = 0; i < NX; ++i) Hm[(k * NX + i) + cj * n] += term(k * NX + i, cj);            // hes_xx
                for (int i = 0; i < NU; ++i) Hm[(VARX + k * NU + i) + cj * n] += term(VARX + k * NU + i, cj);   // hes_ux
            }
            for (int j = 0; j < NU; ++j) {
                const int cj = VARX + k * NU + j;
                for (int i = 0; i < NX; ++i) Hm[(k * NX + i) + cj * n] += term(cj, k * NX + i);            // hes_xu = hes_ux'
                for (int i = 0; i < NU; ++i) Hm[(VARX + k * NU + i) + cj * n] += term(VARX + k * NU + i, cj);   // hes_uu
            }
        }
        if (NP > 0) {
            const int a = VARX + VARU;
            for (int j = 0; j < NP; ++j) {
                for (int i = 0; i < a; ++i) Hm[i + (a + j) * n] += term(i, a + j);          // hes_ap into the parameter columns
                for (int i = 0; i < NP; ++i) Hm[(a + i) + (a + j) * n] += term(a + i, a + j);   // hes_pp
            }
            for (int j = 0; j < a; ++j) for (int i = 0; i < NP; ++i) Hm[(a + i) + j * n] += term(j, a + i);   // rows: hes_ap(j, :)
        }
    }

    // continuous_ocp.hpp:147-159
    void set_time_limits(double t0, double tf) {
        t_start = t0; t_stop = tf;
        const double t_length = (t_stop - t_start) / S;
        const double t_shift = t_length / 2;
        for (int i = 0; i < S; ++i)
            for (int j = 0; j <= P; ++j)  // m_nodes.reverse()
                time_nodes[i * P + j] = (t_length / 2) * cheb.nodes[P - j] + (t_start + t_shift + i * t_length) * 1.0;
        std::reverse(time_nodes.begin(), time_nodes.end());
    }
    double t_scale() const { return (t_stop - t_start) / (2 * S); }

    // DX = D * X per segment, later segments overwrite the junction row (:750-751, :811-812)
    void diff_states(const double* var, std::vector<double>& DX) const {
        DX.assign(NN * NX, 0.0);  // DX(node, state), node-major: DX[k*NX + s]
        for (int seg = 0; seg < S; ++seg)
            for (int i = 0; i <= P; ++i)
                for (int s = 0; s < NX; ++s) {
                    double acc = 0.0;
                    for (int j = 0; j <= P; ++j) acc += cheb.Dij(i, j) * var[(seg * P + j) * NX + s];
                    DX[(seg * P + i) * NX + s] = acc;
                }
    }

    // :739-766
    void equalities(const double* var, const double* d, double* c) const {
        std::vector<double> DX; diff_states(var, DX);
        const double ts = t_scale();
        double f[NX > 0 ? NX : 1];
        const double* p = var + VARX + VARU;
        for (int k = 0; k < NN; ++k) {
            for (int s = 0; s < NX; ++s) f[s] = 0.0;
            model.template dynamics<double>(var + k * NX, var + VARX + k * NU, p, d, time_nodes[k], f);
            for (int s = 0; s < NX; ++s) { c[k * NX + s] = DX[k * NX + s]; c[k * NX + s] -= ts * f[s]; }
        }
    }

    // :770-782
    void inequalities(const double* var, const double* d, double* g) const {
        if (NG == 0) return;
        double gr[NG > 0 ? NG : 1];
        const double* p = var + VARX + VARU;
        for (int k = 0; k < NN; ++k) {
            for (int s = 0; s < NG; ++s) gr[s] = 0.0;
            model.template inequality<double>(var + k * NX, var + VARX + k * NU, p, d, time_nodes[k], gr);
            for (int s = 0; s < NG; ++s) g[k * NG + s] = gr[s];
        }
    }

    template <class T> void seed1(const double* var, int k, T* x, T* u, T* p) const {
        int idx = 0;
        for (int i = 0; i < NX; ++i, ++idx) { x[i] = T(var[k * NX + i]); x[i].d[idx] = 1.0; }
        for (int i = 0; i < NU; ++i, ++idx) { u[i] = T(var[VARX + k * NU + i]); u[i].d[idx] = 1.0; }
        for (int i = 0; i < NP; ++i, ++idx) { p[i] = T(var[VARX + VARU + i]); p[i].d[idx] = 1.0; }
    }
    // second-order seeding, :691-735: outer derivative = unit, inner value's derivative = unit
    void seed2(const double* var, int k, ad2* x, ad2* u, ad2* p) const {
        int idx = 0;
        auto mk = [&](double val, int id) {
            ad2 r; r.v = ad1(val); r.v.d[id] = 1.0; r.d[id] = ad1(1.0); return r;
        };
        for (int i = 0; i < NX; ++i, ++idx) x[i] = mk(var[k * NX + i], idx);
        for (int i = 0; i < NU; ++i, ++idx) u[i] = mk(var[VARX + k * NU + i], idx);
        for (int i = 0; i < NP; ++i, ++idx) p[i] = mk(var[VARX + VARU + i], idx);
    }

    // :797-878 ; J is NUM_EQ x VAR_SIZE written into a matrix with leading dimension ldj at row offset 0
    void equalities_linearised(const double* var, const double* d, double* c, double* J, int ldj) const {
        const int n = VAR_SIZE;
        for (int j = 0; j < n; ++j) for (int i = 0; i < NUM_EQ; ++i) J[i + j * ldj] = 0.0;
        const double ts = t_scale();
        std::vector<double> DX; diff_states(var, DX);
        // D (x) I blocks, rows i<P of each segment (:817-827)
        for (int s = 0; s < S; ++s)
            for (int i = 0; i < P; ++i)
                for (int j = 0; j <= P; ++j) {
                    int shift = s * P * NX;
                    for (int q = 0; q < NX; ++q)
                        for (int r = 0; r < NX; ++r)
                            J[(shift + i * NX + r) + (shift + j * NX + q) * ldj] = (r == q) ? cheb.Dij(i, j) * 1.0 : 0.0;
                }
        // last node row = -reverse(first block row) (:845-846)
        {
            const int W = NX * (P + 1);
            std::vector<double> first(NX * W);
            for (int r = 0; r < NX; ++r) for (int cidx = 0; cidx < W; ++cidx) first[r + cidx * NX] = J[r + cidx * ldj];
            for (int r = 0; r < NX; ++r)
                for (int cidx = 0; cidx < W; ++cidx)
                    J[(VARX - NX + r) + (VARX - W + cidx) * ldj] = -first[(NX - 1 - r) + (W - 1 - cidx) * NX];
        }
        ad1 x[NX > 0 ? NX : 1], u[NU > 0 ? NU : 1], p[NP > 0 ? NP : 1], y[NX > 0 ? NX : 1];
        for (int k = 0; k < NN; ++k) {
            seed1<ad1>(var, k, x, u, p);
            for (int s = 0; s < NX; ++s) y[s] = ad1(0.0);
            model.template dynamics<ad1>(x, u, p, d, ad1(time_nodes[k]), y);
            for (int i = 0; i < NX; ++i) { c[k * NX + i] = -ts * y[i].v; c[k * NX + i] += DX[k * NX + i]; }
            for (int i = 0; i < NX; ++i) {
                for (int j = 0; j < NX; ++j) J[(k * NX + i) + (k * NX + j) * ldj] -= ts * y[i].d[j];
                for (int j = 0; j < NU; ++j) J[(k * NX + i) + (VARX + k * NU + j) * ldj] -= ts * y[i].d[NX + j];
                for (int j = 0; j < NP; ++j) J[(k * NX + i) + (VARX + VARU + j) * ldj] -= ts * y[i].d[NX + NU + j];
            }
        }
    }

    // :546-575 ; Jg is NUM_INEQ x VAR_SIZE inside a matrix with leading dimension ldj
    void inequalities_linearised(const double* var, const double* d, double* g, double* Jg, int ldj) const {
        if (NG == 0) return;
        for (int j = 0; j < VAR_SIZE; ++j) for (int i = 0; i < NUM_INEQ; ++i) Jg[i + j * ldj] = 0.0;
        ad1 x[NX > 0 ? NX : 1], u[NU > 0 ? NU : 1], p[NP > 0 ? NP : 1], gg[NG > 0 ? NG : 1];
        for (int k = 0; k < NN; ++k) {
            seed1<ad1>(var, k, x, u, p);
            for (int s = 0; s < NG; ++s) gg[s] = ad1(0.0);
            model.template inequality<ad1>(x, u, p, d, time_nodes[k], gg);
            for (int i = 0; i < NG; ++i) {
                g[k * NG + i] = gg[i].v;
                for (int j = 0; j < NX; ++j) Jg[(k * NG + i) + (k * NX + j) * ldj] = gg[i].d[j];
                for (int j = 0; j < NU; ++j) Jg[(k * NG + i) + (VARX + k * NU + j) * ldj] = gg[i].d[NX + j];
                for (int j = 0; j < NP; ++j) Jg[(k * NG + i) + (VARX + VARU + j) * ldj] = gg[i].d[NX + NU + j];
            }
        }
    }

    // :1182-1207
    void cost(const double* var, const double* d, double& c) const {
        c = 0.0; double ci = 0.0;
        const double ts = t_scale();
        const double* p = var + VARX + VARU;
        for (int s = 0; s < S; ++s) {
            int shift = s * P;
            for (int k = 0; k <= P; ++k) {
                model.template lagrange<double>(var + (k + shift) * NX, var + VARX + (k + shift) * NU, p, d,
                                                time_nodes[k + shift], ci);
                c += ts * cheb.weights[k] * ci;
            }
        }
        ci = 0.0;
        model.template mayer<double>(var, var + VARX, p, d, time_nodes[0], ci);
        c += ci;
    }

    // :1210-1249
    void cost_gradient(const double* var, const double* d, double& c, double* grad) const {
        c = 0.0;
        for (int i = 0; i < VAR_SIZE; ++i) grad[i] = 0.0;
        const double ts = t_scale();
        ad1 x[NX > 0 ? NX : 1], u[NU > 0 ? NU : 1], p[NP > 0 ? NP : 1], L;
        L = ad1(0.0);
        for (int s = 0; s < S; ++s) {
            int shift = s * P;
            for (int k = 0; k <= P; ++k) {
                seed1<ad1>(var, k + shift, x, u, p);
                model.template lagrange<ad1>(x, u, p, d, time_nodes[k + shift], L);
                const double wk = ts * cheb.weights[k];
                c += wk * L.v;
                for (int i = 0; i < NX; ++i) grad[(k + shift) * NX + i] += wk * L.d[i];
                for (int i = 0; i < NU; ++i) grad[VARX + (k + shift) * NU + i] += wk * L.d[NX + i];
                for (int i = 0; i < NP; ++i) grad[VARX + VARU + i] += wk * L.d[NX + NU + i];
            }
        }
        ad1 M(0.0);
        seed1<ad1>(var, 0, x, u, p);
        model.template mayer<ad1>(x, u, p, d, time_nodes[0], M);
        c += M.v;
        for (int i = 0; i < NX; ++i) grad[i] += M.d[i];
        for (int i = 0; i < NU; ++i) grad[VARX + i] += M.d[NX + i];
        for (int i = 0; i < NP; ++i) grad[VARX + VARU + i] += M.d[NX + NU + i];
    }

    // map local derivative index -> global variable index for node k
    int gidx(int k, int i) const {
        if (i < NX) return k * NX + i;
        if (i < NX + NU) return VARX + k * NU + (i - NX);
        return VARX + VARU + (i - NX - NU);
    }

    // :1256-1367 ; H is VAR_SIZE x VAR_SIZE column-major
    void cost_gradient_hessian(const double* var, const double* d, double& c, double* grad, double* H) const {
        const int n = VAR_SIZE;
        c = 0.0;
        for (int i = 0; i < n; ++i) grad[i] = 0.0;
        for (int i = 0; i < n * n; ++i) H[i] = 0.0;
        const double ts = t_scale();
        ad2 x[NX > 0 ? NX : 1], u[NU > 0 ? NU : 1], p[NP > 0 ? NP : 1], L;
        L = ad2(0.0);
        for (int s = 0; s < S; ++s) {
            int shift = s * P;
            for (int k = 0; k <= P; ++k) {
                const int kk = k + shift;
                seed2(var, kk, x, u, p);
                model.template lagrange<ad2>(x, u, p, d, time_nodes[kk], L);
                const double coeff = ts * cheb.weights[k];
                c += coeff * L.v.v;
                for (int i = 0; i < NDER; ++i) grad[gidx(kk, i)] += coeff * L.v.d[i];
                // hes.col(i) = L.d[i].d  => hes(r, i) = L.d[i].d[r]
                for (int i = 0; i < NDER; ++i)
                    for (int r = 0; r < NDER; ++r)
                        H[gidx(kk, r) + gidx(kk, i) * n] += coeff * L.d[i].d[r];
            }
        }
        // Mayer term at node 0
        ad2 M(0.0);
        seed2(var, 0, x, u, p);
        model.template mayer<ad2>(x, u, p, d, time_nodes[0], M);
        c += M.v.v;
        for (int i = 0; i < NDER; ++i) grad[gidx(0, i)] += M.v.d[i];
        for (int i = 0; i < NDER; ++i)
            for (int r = 0; r < NDER; ++r) {
                const bool pp = (r >= NX + NU) && (i >= NX + NU);
                if (!pp) { H[gidx(0, r) + gidx(0, i) * n] += M.d[i].d[r]; continue; }
                // quirk Q4 (:1354): pp block goes to bottomLeftCorner<NP,NP> and is read from hes.bottomLeftCorner
                // i.e. dst(n-NP+a, b) += hes(NDER-NP+a, b) for a,b<NP  — handled below
            }
        for (int a = 0; a < NP; ++a)
            for (int b = 0; b < NP; ++b)
                H[(n - NP + a) + b * n] += M.d[b].d[NDER - NP + a];
    }

    // :1960-1975 ; g has NUM_EQ+NUM_INEQ entries, jac is (NUM_EQ+NUM_INEQ) x VAR_SIZE column-major
    void lagrangian_gradient(const double* var, const double* d, const double* lam, double& lag, double* lag_grad,
                             double* cost_grad, double* g, double* jac) const {
        const int n = VAR_SIZE, m = NUM_EQ + NUM_INEQ;
        cost_gradient(var, d, lag, cost_grad);
        equalities_linearised(var, d, g, jac, m);
        inequalities_linearised(var, d, g + NUM_EQ, jac + NUM_EQ, m);
        for (int j = 0; j < n; ++j) {
            double acc = 0.0;
            for (int i = 0; i < m; ++i) acc += jac[i + j * m] * lam[i];
            lag_grad[j] = acc;
        }
        for (int j = 0; j < n; ++j) lag_grad[j] += cost_grad[j];
        for (int j = 0; j < n; ++j) lag_grad[j] += lam[m + j];
    }

    // :2100-2174
    void lagrangian_gradient_hessian(const double* var, const double* d, const double* lam, double& lag,
                                     double* lag_grad, double* H, double* cost_grad, double* g, double* jac,
                                     double cost_scale = 1.0) const {
        const int n = VAR_SIZE, m = NUM_EQ + NUM_INEQ;
        cost_gradient_hessian(var, d, lag, cost_grad, H);
        equalities_linearised(var, d, g, jac, m);
        inequalities_linearised(var, d, g + NUM_EQ, jac + NUM_EQ, m);
        for (int j = 0; j < n; ++j) {
            double acc = 0.0;
            for (int i = 0; i < m; ++i) acc += jac[i + j * m] * lam[i];
            lag_grad[j] = acc;
        }
        for (int j = 0; j < n; ++j) lag_grad[j] += cost_grad[j];
        for (int j = 0; j < n; ++j) lag_grad[j] += lam[m + j];

        if (cost_scale != 1.0) for (int i = 0; i < n * n; ++i) H[i] = cost_scale * H[i];

        const double ts = t_scale();
        ad2 x[NX > 0 ? NX : 1], u[NU > 0 ? NU : 1], p[NP > 0 ? NP : 1], xdot[NX > 0 ? NX : 1], gg[NG > 0 ? NG : 1];
        double hes[NDER * NDER];
        for (int k = 0; k < NN; ++k) {
            seed2(var, k, x, u, p);
            for (int i = 0; i < NDER * NDER; ++i) hes[i] = 0.0;
            for (int s = 0; s < NX; ++s) xdot[s] = ad2(0.0);
            model.template dynamics<ad2>(x, u, p, d, ad2(time_nodes[k]), xdot);
            for (int q = 0; q < NX; ++q) {
                const double coeff = -lam[q + k * NX] * ts;
                for (int i = 0; i < NDER; ++i)
                    for (int r = 0; r < NDER; ++r) hes[r + i * NDER] += coeff * xdot[q].d[i].d[r];
            }
            if (NG > 0) {
                for (int s = 0; s < NG; ++s) gg[s] = ad2(0.0);
                model.template inequality<ad2>(x, u, p, d, time_nodes[k], gg);
                for (int q = 0; q < NG; ++q) {
                    const double coeff = lam[q + k * NG + NUM_EQ];
                    for (int i = 0; i < NDER; ++i)
                        for (int r = 0; r < NDER; ++r) hes[r + i * NDER] += coeff * gg[q].d[i].d[r];
                }
            }
            for (int i = 0; i < NDER; ++i)
                for (int r = 0; r < NDER; ++r) H[gidx(k, r) + gidx(k, i) * n] += hes[r + i * NDER];
        }
    }
};

}  // namespace oracle
