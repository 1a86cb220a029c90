// ORACLE — TEST INFRASTRUCTURE ONLY (CPU restatement of the reference algorithm).
// Nothing under polympc_amd/ or include/ may include or link this file.
//
// Forward-mode automatic differentiation scalar, nestable for second derivatives.
// Restates the derivative rules of the reference's forked Eigen AutoDiffScalar
// (/root/reference/src/autodiff/AutoDiffScalar.h:545-690 unary rules, and the
// binary operators of the same file): value + N partials, with
//   (a*b)' = a'*b.v + b'*a.v
//   (a/b)' = (a'*b.v - b'*a.v) * (1/(b.v*b.v))
//   sin' = a'*cos(a.v), cos' = a'*(-sin(a.v)), exp' = a'*exp(a.v), sqrt' = a'*(1/(2 sqrt))
// Nested once (Dual<Dual<double,N>,N>) it gives the same value / gradient / Hessian
// triple that ContinuousOCP seeds at continuous_ocp.hpp:691-735.
#pragma once
#include <cmath>
#include <type_traits>
#include "../polympc_amd/csrc/pmpc_math.hpp"   // a maths library (sin / cos / exp restated with IEEE operations only), shared with the
                                               // device code so that both sides produce the same bits; pinned to glibc within 1 ulp by the tests

namespace oracle {

// Which sin / cos / exp the restatement evaluates the models with:
//   false (default) — pmpc::detmath, the implementation the HIP kernels use: GPU-vs-oracle comparisons are then bit for bit;
//   true            — glibc, what the reference binary itself calls: used by the tests that tie the two choices together
//                     (identical SQP / ADMM iteration counts, solutions within 1e-8) and by the known-answer pins.
inline bool& use_libm() { static bool flag = false; return flag; }

template <class S, int N>
struct Dual {
    S v;
    S d[N > 0 ? N : 1];

    Dual() : v(0.0) { for (int i = 0; i < N; ++i) d[i] = S(0.0); }
    Dual(double c) : v(c) { for (int i = 0; i < N; ++i) d[i] = S(0.0); }
    // promote an inner scalar (only meaningful when S is itself a Dual)
    template <class Q = S, class = typename std::enable_if<!std::is_same<Q, double>::value>::type>
    Dual(const S& s) : v(s) { for (int i = 0; i < N; ++i) d[i] = S(0.0); }

    friend Dual operator+(const Dual& a, const Dual& b) {
        Dual r; r.v = a.v + b.v;
        for (int i = 0; i < N; ++i) r.d[i] = a.d[i] + b.d[i];
        return r;
    }
    friend Dual operator-(const Dual& a, const Dual& b) {
        Dual r; r.v = a.v - b.v;
        for (int i = 0; i < N; ++i) r.d[i] = a.d[i] - b.d[i];
        return r;
    }
    friend Dual operator-(const Dual& a) {
        Dual r; r.v = -a.v;
        for (int i = 0; i < N; ++i) r.d[i] = -a.d[i];
        return r;
    }
    friend Dual operator*(const Dual& a, const Dual& b) {
        Dual r; r.v = a.v * b.v;
        for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * b.v + b.d[i] * a.v;
        return r;
    }
    friend Dual operator/(const Dual& a, const Dual& b) {
        Dual r; r.v = a.v / b.v;
        S inv = S(1.0) / (b.v * b.v);
        for (int i = 0; i < N; ++i) r.d[i] = (a.d[i] * b.v - b.d[i] * a.v) * inv;
        return r;
    }
    Dual& operator+=(const Dual& o) { *this = *this + o; return *this; }
    Dual& operator-=(const Dual& o) { *this = *this - o; return *this; }
    Dual& operator*=(const Dual& o) { *this = *this * o; return *this; }
    Dual& operator/=(const Dual& o) { *this = *this / o; return *this; }
};

// scalar helpers so the same rules recurse through the nesting
inline double ad_sin(double x) { return use_libm() ? std::sin(x) : pmpc::detmath::sin(x); }
inline double ad_cos(double x) { return use_libm() ? std::cos(x) : pmpc::detmath::cos(x); }
inline double ad_exp(double x) { return use_libm() ? std::exp(x) : pmpc::detmath::exp(x); }
inline double ad_sqrt(double x) { return std::sqrt(x); }
inline double ad_tanh(double x) { return std::tanh(x); }

template <class S, int N> Dual<S, N> ad_sin(const Dual<S, N>& a);
template <class S, int N> Dual<S, N> ad_cos(const Dual<S, N>& a);
template <class S, int N> Dual<S, N> ad_exp(const Dual<S, N>& a);
template <class S, int N> Dual<S, N> ad_sqrt(const Dual<S, N>& a);
template <class S, int N> Dual<S, N> ad_tanh(const Dual<S, N>& a);

template <class S, int N> Dual<S, N> ad_sin(const Dual<S, N>& a) {
    Dual<S, N> r; r.v = ad_sin(a.v); S c = ad_cos(a.v);
    for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * c;
    return r;
}
template <class S, int N> Dual<S, N> ad_cos(const Dual<S, N>& a) {
    Dual<S, N> r; r.v = ad_cos(a.v); S s = -ad_sin(a.v);
    for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * s;
    return r;
}
template <class S, int N> Dual<S, N> ad_exp(const Dual<S, N>& a) {
    Dual<S, N> r; r.v = ad_exp(a.v);
    for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * r.v;
    return r;
}
template <class S, int N> Dual<S, N> ad_sqrt(const Dual<S, N>& a) {
    Dual<S, N> r; r.v = ad_sqrt(a.v); S h = S(1.0) / (S(2.0) * r.v);
    for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * h;
    return r;
}
template <class S, int N> Dual<S, N> ad_tanh(const Dual<S, N>& a) {
    Dual<S, N> r; r.v = ad_tanh(a.v); S h = S(1.0) - r.v * r.v;
    for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * h;
    return r;
}

// names the model code calls (ADL finds these for Dual; the double overloads route through the switch above)
inline double sin(double x) { return ad_sin(x); }
inline double cos(double x) { return ad_cos(x); }
inline double exp(double x) { return ad_exp(x); }
template <class S, int N> Dual<S, N> sin(const Dual<S, N>& a) { return ad_sin(a); }
template <class S, int N> Dual<S, N> cos(const Dual<S, N>& a) { return ad_cos(a); }
template <class S, int N> Dual<S, N> exp(const Dual<S, N>& a) { return ad_exp(a); }
template <class S, int N> Dual<S, N> sqrt(const Dual<S, N>& a) { return ad_sqrt(a); }
template <class S, int N> Dual<S, N> tanh(const Dual<S, N>& a) { return ad_tanh(a); }

// value extraction through any nesting depth
inline double value_of(double x) { return x; }
template <class S, int N> double value_of(const Dual<S, N>& a) { return value_of(a.v); }

}  // namespace oracle
