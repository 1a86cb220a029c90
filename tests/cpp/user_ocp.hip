// A USER-defined OCP compiled for the GPU in the user's own translation unit (what a PolyMPC user does with their
// ContinuousOCP subclass). UserRobot restates tests/control/mpc_wrapper_test.cpp:46-74 from the "user" side so the
// test can check that the registration path gives bit-identical results to the built-in model; Pendulum is a second,
// unrelated model (NX=2, NU=1) with a path constraint (NG=1).
#include <polympc/register_ocp.hpp>

struct UserRobot {
    enum { NX = 3, NU = 2, NP = 0, ND = 1, NG = 0 };
    double q = 1.0;
    template <class T>
    __device__ void dynamics_impl(pmpc::cref<T> x, pmpc::cref<T> u, pmpc::cref<T> p, pmpc::cref<double> d, const T& t, pmpc::vref<T> xdot) const {
        xdot(0) = u(0) * cos(x(2)) * cos(u(1));
        xdot(1) = u(0) * sin(x(2)) * cos(u(1));
        xdot(2) = u(0) * sin(u(1)) / T(d(0));
    }
    template <class T>
    __device__ void lagrange_term_impl(pmpc::cref<T> x, pmpc::cref<T> u, pmpc::cref<T> p, pmpc::cref<double> d, double t, T& lagrange) const {
        T a = x(0) * (T(q) * x(0)); a = a + x(1) * (T(q) * x(1)); a = a + x(2) * (T(q) * x(2));
        T b = u(0) * (T(1.0) * u(0)); b = b + u(1) * (T(1.0) * u(1));
        lagrange = a + b;
    }
    template <class T>
    __device__ void mayer_term_impl(pmpc::cref<T> x, pmpc::cref<T> u, pmpc::cref<T> p, pmpc::cref<double> d, double t, T& mayer) const {
        T a = x(0) * (T(1.0) * x(0)); a = a + x(1) * (T(1.0) * x(1)); a = a + x(2) * (T(1.0) * x(2));
        mayer = a;
    }
    template <class T>
    __device__ void inequality_constraints_impl(pmpc::cref<T>, pmpc::cref<T>, pmpc::cref<T>, pmpc::cref<double>, double, pmpc::vref<T>) const {}
};
PMPC_REGISTER_OCP(UserRobot)

struct Pendulum {
    enum { NX = 2, NU = 1, NP = 0, ND = 1, NG = 1 };
    template <class T>
    __device__ void dynamics_impl(pmpc::cref<T> x, pmpc::cref<T> u, pmpc::cref<T>, pmpc::cref<double> d, const T&, pmpc::vref<T> xdot) const {
        xdot(0) = x(1);
        xdot(1) = T(-9.81 / d(0)) * sin(x(0)) - T(0.1) * x(1) + u(0);
    }
    template <class T>
    __device__ void lagrange_term_impl(pmpc::cref<T> x, pmpc::cref<T> u, pmpc::cref<T>, pmpc::cref<double>, double, T& lagrange) const {
        lagrange = x(0) * x(0) + T(0.1) * x(1) * x(1) + T(0.01) * u(0) * u(0);
    }
    template <class T>
    __device__ void mayer_term_impl(pmpc::cref<T> x, pmpc::cref<T>, pmpc::cref<T>, pmpc::cref<double>, double, T& mayer) const {
        mayer = T(10.0) * x(0) * x(0) + x(1) * x(1);
    }
    template <class T>
    __device__ void inequality_constraints_impl(pmpc::cref<T> x, pmpc::cref<T> u, pmpc::cref<T>, pmpc::cref<double>, double, pmpc::vref<T> g) const {
        g(0) = x(1) + T(0.5) * u(0);   // a mixed state-input path constraint
    }
};
PMPC_REGISTER_OCP(Pendulum)
