// polympc_amd — device forward-mode AD scalar (gfx950).
// Capability replaced: the reference's forked Eigen::AutoDiffScalar (src/autodiff/AutoDiffScalar.h), nested twice for
// Hessians (continuous_ocp.hpp:124-142, seeding :691-735). Same derivative rules, plain registers instead of Eigen
// expression templates, usable from __device__ code so a user's templated dynamics_impl<T> compiles for the GPU.
#pragma once
#include <hip/hip_runtime.h>

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

__host__ __device__ inline double m_sin(double x) { return ::sin(x); }
__host__ __device__ inline double m_cos(double x) { return ::cos(x); }
__host__ __device__ inline double m_exp(double x) { return ::exp(x); }
__host__ __device__ inline double m_sqrt(double x) { return ::sqrt(x); }
// sine and cosine of one argument with ONE argument reduction (the device library's sincos shares it between the two
// polynomial kernels: ~200 instructions instead of ~350 for separate calls, which the compiler does not merge)
__host__ __device__ inline void m_sincos(double x, double& s, double& c) { ::sincos(x, &s, &c); }

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
__host__ __device__ inline double sin(double x) { return ::sin(x); }
__host__ __device__ inline double cos(double x) { return ::cos(x); }
__host__ __device__ inline double exp(double x) { return ::exp(x); }
__host__ __device__ inline double sqrt(double x) { return ::sqrt(x); }
template <class S, int N> __host__ __device__ Dual<S, N> sin(const Dual<S, N>& a) { return m_sin(a); }
template <class S, int N> __host__ __device__ Dual<S, N> cos(const Dual<S, N>& a) { return m_cos(a); }
template <class S, int N> __host__ __device__ Dual<S, N> exp(const Dual<S, N>& a) { return m_exp(a); }
template <class S, int N> __host__ __device__ Dual<S, N> sqrt(const Dual<S, N>& a) { return m_sqrt(a); }
// extension for model code: both values of one angle at the price of one (see m_sincos)
__host__ __device__ inline void sincos(double x, double& s, double& c) { m_sincos(x, s, c); }
template <class S, int N> __host__ __device__ void sincos(const Dual<S, N>& a, Dual<S, N>& s, Dual<S, N>& c) { m_sincos(a, s, c); }

}  // namespace pmpc
