// ORACLE — TEST INFRASTRUCTURE ONLY.
//
// Generic (non-OCP) NLP with whole-vector forward AD. Follows /root/reference/src/solvers/nlproblem.hpp
// (cost_gradient_impl, cost_gradient_hessian_impl, lagrangian_gradient_impl, lagrangian_gradient_hessian_impl,
// lines ~480-690). Used only to transcribe the SQP known-answer tests of
// /root/reference/tests/solvers/sqp/sqp_test_autodiff.cpp (they pin SQP + boxADMM + BFGS end to end).
// Def concept: enum {NX, NE, NI}; cost<T>(x,c); eq<T>(x,ce); ineq<T>(x,ci).
#pragma once
#include <vector>
#include "ad.hpp"

namespace oracle {

template <class Def>
struct GenericNLP {
    enum { NXV = Def::NX, NE = Def::NE, NI = Def::NI };
    using ad1 = Dual<double, NXV>;
    using ad2 = Dual<ad1, NXV>;
    Def def;
    int VAR_SIZE = NXV, NUM_EQ = NE, NUM_INEQ = NI;
    static constexpr bool HAS_BLOCK_BFGS = false;

    void seed1(const double* x, ad1* v) const { for (int i = 0; i < NXV; ++i) { v[i] = ad1(x[i]); v[i].d[i] = 1.0; } }
    void seed2(const double* x, ad2* v) const {
        for (int i = 0; i < NXV; ++i) { ad2 r; r.v = ad1(x[i]); r.v.d[i] = 1.0; r.d[i] = ad1(1.0); v[i] = r; }
    }
    void cost(const double* x, const double*, double& c) const { def.template cost<double>(x, c); }
    void equalities(const double* x, const double*, double* c) const { if (NE > 0) def.template eq<double>(x, c); }
    void inequalities(const double* x, const double*, double* g) const { if (NI > 0) def.template ineq<double>(x, g); }

    void linearise_constraints(const double* x, double* g, double* jac) const {
        const int m = NE + NI;
        ad1 v[NXV]; seed1(x, v);
        ad1 ce[NE > 0 ? NE : 1], ci[NI > 0 ? NI : 1];
        if (NE > 0) def.template eq<ad1>(v, ce);
        if (NI > 0) def.template ineq<ad1>(v, ci);
        for (int i = 0; i < NE; ++i) { g[i] = ce[i].v; for (int j = 0; j < NXV; ++j) jac[i + j * m] = ce[i].d[j]; }
        for (int i = 0; i < NI; ++i) { g[NE + i] = ci[i].v; for (int j = 0; j < NXV; ++j) jac[(NE + i) + j * m] = ci[i].d[j]; }
    }
    void finish_lag_grad(const double* lam, double* lag_grad, const double* cost_grad, const double* jac) const {
        const int m = NE + NI;
        for (int j = 0; j < NXV; ++j) { double a = 0; for (int i = 0; i < m; ++i) a += jac[i + j * m] * lam[i]; lag_grad[j] = a; }
        for (int j = 0; j < NXV; ++j) lag_grad[j] += cost_grad[j];
        for (int j = 0; j < NXV; ++j) lag_grad[j] += lam[m + j];
    }
    void lagrangian_gradient(const double* x, const double*, const double* lam, double& lag, double* lag_grad,
                             double* cost_grad, double* g, double* jac) const {
        ad1 v[NXV]; seed1(x, v); ad1 c(0.0);
        def.template cost<ad1>(v, c);
        lag = c.v; for (int j = 0; j < NXV; ++j) cost_grad[j] = c.d[j];
        linearise_constraints(x, g, jac);
        finish_lag_grad(lam, lag_grad, cost_grad, jac);
    }
    void lagrangian_gradient_hessian(const double* x, const double*, const double* lam, double& lag, double* lag_grad,
                                     double* H, double* cost_grad, double* g, double* jac) const {
        ad2 v[NXV]; seed2(x, v); ad2 c(0.0);
        def.template cost<ad2>(v, c);
        lag = c.v.v;
        for (int j = 0; j < NXV; ++j) cost_grad[j] = c.v.d[j];
        for (int i = 0; i < NXV; ++i) for (int r = 0; r < NXV; ++r) H[r + i * NXV] = c.d[i].d[r];
        linearise_constraints(x, g, jac);
        finish_lag_grad(lam, lag_grad, cost_grad, jac);
        ad2 ce[NE > 0 ? NE : 1], ci[NI > 0 ? NI : 1];
        if (NE > 0) def.template eq<ad2>(v, ce);
        for (int q = 0; q < NE; ++q)  // hes.col(i) = d[i].d ; transposeInPlace ; H += lam(q)*hes
            for (int i = 0; i < NXV; ++i) for (int r = 0; r < NXV; ++r) H[i + r * NXV] += lam[q] * ce[q].d[i].d[r];
        if (NI > 0) def.template ineq<ad2>(v, ci);
        for (int q = 0; q < NI; ++q)
            for (int i = 0; i < NXV; ++i) for (int r = 0; r < NXV; ++r) H[i + r * NXV] += lam[q + NE] * ci[q].d[i].d[r];
    }
};

// sqp_test_autodiff.cpp:50-76
struct ConstrainedRosenbrockDef {
    enum { NX = 2, NE = 1, NI = 0 };
    template <class T> void cost(const T* x, T& c) const {
        T a(1.0), b(100.0);
        c = (a - x[0]) * (a - x[0]) + b * (x[1] - x[0] * x[0]) * (x[1] - x[0] * x[0]);
    }
    template <class T> void eq(const T* x, T* ce) const { ce[0] = (x[0] * x[0] + x[1] * x[1]) - T(1.0); }
    template <class T> void ineq(const T*, T*) const {}
};
// :100-117
struct RosenbrockDef {
    enum { NX = 2, NE = 0, NI = 0 };
    template <class T> void cost(const T* x, T& c) const {
        T a(1.0), b(100.0);
        c = (a - x[0]) * (a - x[0]) + b * (x[1] - x[0] * x[0]) * (x[1] - x[0] * x[0]);
    }
    template <class T> void eq(const T*, T*) const {}
    template <class T> void ineq(const T*, T*) const {}
};
// :140-163
struct SimpleNLPDef {
    enum { NX = 2, NE = 0, NI = 1 };
    template <class T> void cost(const T* x, T& c) const { c = -x[0] - x[1]; }
    template <class T> void eq(const T*, T*) const {}
    template <class T> void ineq(const T* x, T* ci) const { ci[0] = x[0] * x[0] + x[1] * x[1]; }
};
// :191-221
struct HS071Def {
    enum { NX = 4, NE = 1, NI = 1 };
    template <class T> void cost(const T* x, T& c) const { c = x[0] * x[3] * (x[0] + x[1] + x[2]) + x[2]; }
    template <class T> void eq(const T* x, T* ce) const {
        ce[0] = (x[0] * x[0] + x[1] * x[1] + x[2] * x[2] + x[3] * x[3]) - T(40.0);
    }
    template <class T> void ineq(const T* x, T* ci) const { ci[0] = x[0] * x[1] * x[2] * x[3]; }
};

}  // namespace oracle
