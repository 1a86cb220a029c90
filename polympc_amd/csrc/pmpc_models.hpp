// polympc_amd — built-in OCP definitions, written against the same override points a PolyMPC user implements
// (dynamics_impl / lagrange_term_impl / mayer_term_impl / inequality_constraints_impl, continuous_ocp.hpp:191-288).
// x, u, p, d, xdot, g are pmpc::vref views: element access with (i) exactly as with Eigen::Ref in the reference.
#pragma once
#include <hip/hip_runtime.h>
#include "pmpc_ad.hpp"

namespace pmpc {

template <class T>
struct vref {
    T* p;
    __host__ __device__ explicit vref(T* q) : p(q) {}
    __host__ __device__ T& operator()(int i) const { return p[i]; }
    __host__ __device__ T& operator[](int i) const { return p[i]; }
};
template <class T> using cref = vref<const T>;
// Read-only view of the SEEDED second-order variables of one Hessian entry (Ocp::stage_second_order_entry): element i is generated where the
// model reads it — value from LDS, inner partial 1 on variable `r`, outer partial 1 on variable `dir` — instead of living in a local array of
// NX + NU + NP nested duals (4 doubles each) for the whole model evaluation. Same values, so every entry goes through the same operations.
template <>
struct vref<const Dual<Dual<double, 1>, 1>> {
    using T = Dual<Dual<double, 1>, 1>;
    const double* val; int id0, r, dir;
    __host__ __device__ vref(const double* v_, int id0_, int r_, int dir_) : val(v_), id0(id0_), r(r_), dir(dir_) {}
    __host__ __device__ T operator()(int i) const {
        T t; t.v = Dual<double, 1>(val[i]); t.v.d[0] = (id0 + i == r) ? 1.0 : 0.0; t.d[0] = Dual<double, 1>(id0 + i == dir ? 1.0 : 0.0); return t;
    }
    __host__ __device__ T operator[](int i) const { return (*this)(i); }
};
// doubles viewed as value-only scalars (see Value in pmpc_ad.hpp)
__host__ __device__ inline cref<Value> as_cvalues(const double* p) { return cref<Value>(reinterpret_cast<const Value*>(p)); }
__host__ __device__ inline vref<Value> as_values(double* p) { return vref<Value>(reinterpret_cast<Value*>(p)); }

// Mobile robot (unicycle with steering): tests/control/mpc_wrapper_test.cpp:33-80, docs/source/ocp.rst:229-281.
struct RobotOCP {
    enum { NX = 3, NU = 2, NP = 0, ND = 1, NG = 0 };
    double Q[3] = {1, 1, 1}, R[2] = {1, 1}, QN[3] = {1, 1, 1};
    __host__ void set_params(const double* mp, int n) {
        if (n >= 1) for (int i = 0; i < 3; ++i) Q[i] = mp[0];
        if (n >= 2) for (int i = 0; i < 2; ++i) R[i] = mp[1];
        if (n >= 3) for (int i = 0; i < 3; ++i) QN[i] = mp[2];
    }
    template <class T>
    __device__ void dynamics_impl(cref<T> x, cref<T> u, cref<T>, cref<double> d, const T&, vref<T> xdot) const {
        // the reference's expressions with each angle's sine and cosine evaluated once (same values, same products)
        T s2, c2, s1, c1;
        sincos(x(2), s2, c2); sincos(u(1), s1, c1);
        xdot(0) = u(0) * c2 * c1;
        xdot(1) = u(0) * s2 * c1;
        xdot(2) = u(0) * s1 / T(d(0));
    }
    template <class T>
    __device__ void lagrange_term_impl(cref<T> x, cref<T> u, cref<T>, cref<double>, double, T& lagrange) const {
        T a = x(0) * (T(Q[0]) * x(0)); a = a + x(1) * (T(Q[1]) * x(1)); a = a + x(2) * (T(Q[2]) * x(2));
        T b = u(0) * (T(R[0]) * u(0)); b = b + u(1) * (T(R[1]) * u(1));
        lagrange = a + b;
    }
    template <class T>
    __device__ void mayer_term_impl(cref<T> x, cref<T>, cref<T>, cref<double>, double, T& mayer) const {
        T a = x(0) * (T(QN[0]) * x(0)); a = a + x(1) * (T(QN[1]) * x(1)); a = a + x(2) * (T(QN[2]) * x(2));
        mayer = a;
    }
    template <class T>
    __device__ void inequality_constraints_impl(cref<T>, cref<T>, cref<T>, cref<double>, double, vref<T>) const {}
};

// CSTR: tests/control/cstr_control_test.cpp:30-113
struct CstrOCP {
    enum { NX = 4, NU = 2, NP = 0, ND = 0, NG = 0 };
    double Q[4] = {0.2, 1.0, 0.5, 0.2};
    double R[2] = {0.5, 5.0 * 1.0e-7};
    double Pm[16] = {1.4646778374584373, 0.6676889516721198, 0.35446715117028615, 0.10324422005086348,
                     0.6676889516721198, 1.407812935783267,  0.17788030743777067, 0.050059833257226405,
                     0.3544671511702861, 0.1778803074377706, 0.6336052592712396,  0.01110329497282364,
                     0.1032442200508634, 0.05005983325722643, 0.011103294972823655, 0.229412393739723};
    double xs[4] = {2.1402105301746182e00, 1.0903043613077321e00, 1.1419108442079495e02, 1.1290659291045561e02};
    double us[2] = {14.19, -1113.50};
    __host__ void set_params(const double*, int) {}
    template <class T>
    __device__ void dynamics_impl(cref<T> x, cref<T> u, cref<T>, cref<double>, const T&, vref<T> xdot) const {
        T c_AO = T(5.1), v_0 = T(104.9), k_w = T(4032.0), A_R = T(0.215), rho = T(0.9342), C_P = T(3.01), V_R = T(10.0);
        T H_1 = T(4.2), H_2 = T(-11.0), H_3 = T(-41.85), m_K = T(5.0), C_PK = T(2.0);
        T k10 = T(1.287e12), k20 = T(1.287e12), k30 = T(9.043e09), E1 = T(-9758.3), E2 = T(-9758.3), E3 = T(-8560.0);
        T k_1 = k10 * exp(E1 / (T(273.15) + x(2)));
        T k_2 = k20 * exp(E2 / (T(273.15) + x(2)));
        T k_3 = k30 * exp(E3 / (T(273.15) + x(2)));
        T TPH = T(3600.0);
        xdot(0) = (T(1) / TPH) * (u(0) * (c_AO - x(0)) - k_1 * x(0) - k_3 * x(0) * x(0));
        xdot(1) = (T(1) / TPH) * (-u(0) * x(1) + k_1 * x(0) - k_2 * x(1));
        xdot(2) = (T(1) / TPH) * (u(0) * (v_0 - x(2)) + (k_w * A_R / (rho * C_P * V_R)) * (x(3) - x(2)) -
                                  (T(1) / (rho * C_P)) * (k_1 * x(0) * H_1 + k_2 * x(1) * H_2 + k_3 * x(0) * x(1) * H_3));
        xdot(3) = (T(1) / TPH) * ((T(1) / (m_K * C_PK)) * (u(1) + k_w * A_R * (x(2) - x(3))));
    }
    template <class T>
    __device__ void lagrange_term_impl(cref<T> x, cref<T> u, cref<T>, cref<double>, double, T& lagrange) const {
        T a(0.0);
        for (int i = 0; i < 4; ++i) { T e = x(i) - T(xs[i]); a = a + e * (T(Q[i]) * e); }
        T b(0.0);
        for (int i = 0; i < 2; ++i) { T e = u(i) - T(us[i]); b = b + e * (T(R[i]) * e); }
        lagrange = a + b;
    }
    template <class T>
    __device__ void mayer_term_impl(cref<T> x, cref<T>, cref<T>, cref<double>, double, T& mayer) const {
        T e[4];
        for (int i = 0; i < 4; ++i) e[i] = x(i) - T(xs[i]);
        T a(0.0);
        for (int i = 0; i < 4; ++i) {
            T Pe(0.0);
            for (int j = 0; j < 4; ++j) Pe = Pe + T(Pm[i * 4 + j]) * e[j];
            a = a + e[i] * Pe;
        }
        mayer = a;
    }
    template <class T>
    __device__ void inequality_constraints_impl(cref<T>, cref<T>, cref<T>, cref<double>, double, vref<T>) const {}
};

// Parking OCP with free time-scaling parameter: tests/control/dense_sparse_compare.cpp:22-55
struct ParkingOCP {
    enum { NX = 3, NU = 2, NP = 1, ND = 1, NG = 0 };
    __host__ void set_params(const double*, int) {}
    template <class T>
    __device__ void dynamics_impl(cref<T> x, cref<T> u, cref<T> p, cref<double> d, const T&, vref<T> xdot) const {
        T s2, c2, s1, c1;
        sincos(x(2), s2, c2); sincos(u(1), s1, c1);
        xdot(0) = p(0) * u(0) * c2 * c1;
        xdot(1) = p(0) * u(0) * s2 * c1;
        xdot(2) = p(0) * u(0) * s1 / T(d(0));
    }
    template <class T>
    __device__ void lagrange_term_impl(cref<T>, cref<T>, cref<T>, cref<double>, double, T&) const {}
    template <class T>
    __device__ void mayer_term_impl(cref<T>, cref<T>, cref<T> p, cref<double>, double, T& mayer) const { mayer = p(0); }
    template <class T>
    __device__ void inequality_constraints_impl(cref<T>, cref<T>, cref<T>, cref<double>, double, vref<T>) const {}
};

// Robot + nonlinear path constraint g = x0^2 + x1^2 (NG = 1)
struct RobotNGOCP : RobotOCP {
    enum { NX = 3, NU = 2, NP = 0, ND = 1, NG = 1 };
    template <class T>
    __device__ void inequality_constraints_impl(cref<T> x, cref<T>, cref<T>, cref<double>, double, vref<T> g) const {
        g(0) = x(0) * x(0) + x(1) * x(1);
    }
};

// Parking with a free time-scaling parameter and the nonlinear path constraint g = u0^2 cos(u1):
// tests/control/nonlinear_constraints_test.cpp:31-75 (NP = 1 and NG = 1 together)
struct ParkingNGOCP : ParkingOCP {
    enum { NX = 3, NU = 2, NP = 1, ND = 1, NG = 1 };
    template <class T>
    __device__ void inequality_constraints_impl(cref<T>, cref<T> u, cref<T>, cref<double>, double, vref<T> g) const {
        g(0) = u(0) * u(0) * cos(u(1));
    }
};

// SYNTHETIC 13-state / 3-input smooth dynamics: dimension stand-in for the kite NMPC config (the reference's
// KiteDynamics / kiteNMPF.h is not in the reference tree). Not a model of anything.
struct KiteStandInOCP {
    enum { NX = 13, NU = 3, NP = 0, ND = 0, NG = 0 };
    __host__ void set_params(const double*, int) {}
    template <class T>
    __device__ void dynamics_impl(cref<T> x, cref<T> u, cref<T>, cref<double>, const T&, vref<T> xdot) const {
        for (int i = 0; i < 13; ++i) {
            const int j = (i + 1) % 13, k = (i + 5) % 13;
            T a = T(-0.1 - 0.01 * i) * x(i);
            T b = T(0.5) * sin(x(j)) * cos(x(k));
            T c = T(0.3 + 0.02 * i) * u(i % 3) * cos(x(i));
            xdot(i) = a + b + c;
        }
    }
    template <class T>
    __device__ void lagrange_term_impl(cref<T> x, cref<T> u, cref<T>, cref<double>, double, T& lagrange) const {
        T a(0.0);
        for (int i = 0; i < 13; ++i) a = a + x(i) * (T(1.0 + 0.1 * i) * x(i));
        for (int i = 0; i < 3; ++i) a = a + u(i) * (T(0.5) * u(i));
        lagrange = a;
    }
    template <class T>
    __device__ void mayer_term_impl(cref<T> x, cref<T>, cref<T>, cref<double>, double, T& mayer) const {
        T a(0.0);
        for (int i = 0; i < 13; ++i) a = a + x(i) * (T(2.0) * x(i));
        mayer = a;
    }
    template <class T>
    __device__ void inequality_constraints_impl(cref<T>, cref<T>, cref<T>, cref<double>, double, vref<T>) const {}
};

}  // namespace pmpc
