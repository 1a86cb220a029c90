// polympc_amd — device forward-mode AD scalar (gfx950).
// Capability replaced: the reference's forked Eigen::AutoDiffScalar (src/autodiff/AutoDiffScalar.h), nested twice for
// Hessians (continuous_ocp.hpp:124-142, seeding :691-735). Same derivative rules, plain registers instead of Eigen
// expression templates, usable from __device__ code so a user's templated dynamics_impl<T> compiles for the GPU.
#pragma once
#include <hip/hip_runtime.h>
#include "pmpc_math.hpp"

namespace pmpc {

template <class S, int N>
struct Dual {
    S v;
    S d[N > 0 ? N : 1];

    __host__ __device__ Dual() : v(0.0) {
#pragma unroll
        for (int i = 0; i < N; ++i) d[i] = S(0.0);
    }
    __host__ __device__ Dual(double c) : v(c) {
#pragma unroll
        for (int i = 0; i < N; ++i) d[i] = S(0.0);
    }
    template <class Q = S, class = typename std::enable_if<!std::is_same<Q, double>::value>::type>
    __host__ __device__ Dual(const S& s) : v(s) {
#pragma unroll
        for (int i = 0; i < N; ++i) d[i] = S(0.0);
    }

    __host__ __device__ friend Dual operator+(const Dual& a, const Dual& b) {
        Dual r; r.v = a.v + b.v;
#pragma unroll
        for (int i = 0; i < N; ++i) r.d[i] = a.d[i] + b.d[i];
        return r;
    }
    __host__ __device__ friend Dual operator-(const Dual& a, const Dual& b) {
        Dual r; r.v = a.v - b.v;
#pragma unroll
        for (int i = 0; i < N; ++i) r.d[i] = a.d[i] - b.d[i];
        return r;
    }
    __host__ __device__ friend Dual operator-(const Dual& a) {
        Dual r; r.v = -a.v;
#pragma unroll
        for (int i = 0; i < N; ++i) r.d[i] = -a.d[i];
        return r;
    }
    __host__ __device__ friend Dual operator*(const Dual& a, const Dual& b) {
        Dual r; r.v = a.v * b.v;
#pragma unroll
        for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * b.v + b.d[i] * a.v;
        return r;
    }
    __host__ __device__ friend Dual operator/(const Dual& a, const Dual& b) {
        Dual r; r.v = a.v / b.v;
        S inv = S(1.0) / (b.v * b.v);
#pragma unroll
        for (int i = 0; i < N; ++i) r.d[i] = (a.d[i] * b.v - b.d[i] * a.v) * inv;
        return r;
    }
};

// sin / cos / exp of a double: pmpc::detmath (pmpc_math.hpp) — IEEE operations only, so the CPU checker that includes the same header
// reproduces every model evaluation bit for bit, and within 1 ulp of the glibc functions the reference calls. Building with
// -DPMPC_LIBM_TRANSCENDENTALS selects the device maths library instead (last-bit differences against any CPU run).
#ifdef PMPC_LIBM_TRANSCENDENTALS
__host__ __device__ inline double m_sin(double x) { return ::sin(x); }
__host__ __device__ inline double m_cos(double x) { return ::cos(x); }
__host__ __device__ inline double m_exp(double x) { return ::exp(x); }
__host__ __device__ inline void m_sincos(double x, double& s, double& c) { ::sincos(x, &s, &c); }
#else
__host__ __device__ inline double m_sin(double x) { return detmath::sin(x); }
__host__ __device__ inline double m_cos(double x) { return detmath::cos(x); }
__host__ __device__ inline double m_exp(double x) { return detmath::exp(x); }
// sine and cosine of one argument with ONE argument reduction
__host__ __device__ inline void m_sincos(double x, double& s, double& c) { const detmath::SinCos r = detmath::sincos(x); s = r.s; c = r.c; }
#endif
__host__ __device__ inline double m_sqrt(double x) { return ::sqrt(x); }

template <class S, int N> __host__ __device__ Dual<S, N> m_sin(const Dual<S, N>& a);
template <class S, int N> __host__ __device__ Dual<S, N> m_cos(const Dual<S, N>& a);
template <class S, int N> __host__ __device__ Dual<S, N> m_exp(const Dual<S, N>& a);
template <class S, int N> __host__ __device__ Dual<S, N> m_sqrt(const Dual<S, N>& a);
template <class S, int N> __host__ __device__ void m_sincos(const Dual<S, N>& a, Dual<S, N>& s, Dual<S, N>& c);

// same derivative rules and operation order as m_sin / m_cos taken separately: d sin = a' * cos, d cos = a' * (-sin)
template <class S, int N> __host__ __device__ void m_sincos(const Dual<S, N>& a, Dual<S, N>& s, Dual<S, N>& c) {
    S sv, cv;
    m_sincos(a.v, sv, cv);
    const S ns = -sv;
    s.v = sv; c.v = cv;
#pragma unroll
    for (int i = 0; i < N; ++i) { s.d[i] = a.d[i] * cv; c.d[i] = a.d[i] * ns; }
}
template <class S, int N> __host__ __device__ Dual<S, N> m_sin(const Dual<S, N>& a) {
    Dual<S, N> r; S sv, cv;
    m_sincos(a.v, sv, cv);
    r.v = sv;
#pragma unroll
    for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * cv;
    return r;
}
template <class S, int N> __host__ __device__ Dual<S, N> m_cos(const Dual<S, N>& a) {
    Dual<S, N> r; S sv, cv;
    m_sincos(a.v, sv, cv);
    r.v = cv; const S ns = -sv;
#pragma unroll
    for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * ns;
    return r;
}
template <class S, int N> __host__ __device__ Dual<S, N> m_exp(const Dual<S, N>& a) {
    Dual<S, N> r; r.v = m_exp(a.v);
#pragma unroll
    for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * r.v;
    return r;
}
template <class S, int N> __host__ __device__ Dual<S, N> m_sqrt(const Dual<S, N>& a) {
    Dual<S, N> r; r.v = m_sqrt(a.v); S h = S(1.0) / (S(2.0) * r.v);
#pragma unroll
    for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * h;
    return r;
}

// the names user model code writes (found by ADL for Dual; ::sin etc. for double)
__host__ __device__ inline double sin(double x) { return m_sin(x); }
__host__ __device__ inline double cos(double x) { return m_cos(x); }
__host__ __device__ inline double exp(double x) { return m_exp(x); }
__host__ __device__ inline double sqrt(double x) { return ::sqrt(x); }
template <class S, int N> __host__ __device__ Dual<S, N> sin(const Dual<S, N>& a) { return m_sin(a); }
template <class S, int N> __host__ __device__ Dual<S, N> cos(const Dual<S, N>& a) { return m_cos(a); }
template <class S, int N> __host__ __device__ Dual<S, N> exp(const Dual<S, N>& a) { return m_exp(a); }
template <class S, int N> __host__ __device__ Dual<S, N> sqrt(const Dual<S, N>& a) { return m_sqrt(a); }
// extension for model code: both values of one angle at the price of one (see m_sincos)
__host__ __device__ inline void sincos(double x, double& s, double& c) { m_sincos(x, s, c); }
template <class S, int N> __host__ __device__ void sincos(const Dual<S, N>& a, Dual<S, N>& s, Dual<S, N>& c) { m_sincos(a, s, c); }

// Value: the scalar of the VALUE-ONLY model evaluations (cost and constraints in the line search and the termination test). It is a
// double inside a struct for one reason: name lookup. A user's model calls sin(x) / cos(x) / exp(x) unqualified, as in the reference;
// with T = double those names would resolve to the device maths library, with T = Value (as with T = Dual) argument-dependent lookup
// finds the functions of this namespace — the value passes and the derivative passes then evaluate every transcendental with the
// same implementation (pmpc_math.hpp), which is also the one the CPU checker uses. Same size and layout as a double: LDS arrays of
// doubles are viewed as arrays of Value (as_values in pmpc_models.hpp).
struct Value {
    double v;
    __host__ __device__ Value() : v(0.0) {}
    __host__ __device__ Value(double c) : v(c) {}
    __host__ __device__ friend Value operator+(const Value& a, const Value& b) { return Value(a.v + b.v); }
    __host__ __device__ friend Value operator-(const Value& a, const Value& b) { return Value(a.v - b.v); }
    __host__ __device__ friend Value operator-(const Value& a) { return Value(-a.v); }
    __host__ __device__ friend Value operator*(const Value& a, const Value& b) { return Value(a.v * b.v); }
    __host__ __device__ friend Value operator/(const Value& a, const Value& b) { return Value(a.v / b.v); }
};
static_assert(sizeof(Value) == sizeof(double) && alignof(Value) == alignof(double), "Value must alias a double");
__host__ __device__ inline Value sin(const Value& a) { return Value(m_sin(a.v)); }
__host__ __device__ inline Value cos(const Value& a) { return Value(m_cos(a.v)); }
__host__ __device__ inline Value exp(const Value& a) { return Value(m_exp(a.v)); }
__host__ __device__ inline Value sqrt(const Value& a) { return Value(m_sqrt(a.v)); }
__host__ __device__ inline void sincos(const Value& a, Value& s, Value& c) { m_sincos(a.v, s.v, c.v); }

}  // namespace pmpc
