"""The IEEE-only sin / cos / exp shared by the HIP kernels and the CPU restatement (polympc_amd/csrc/pmpc_math.hpp), pinned to glibc —
the functions the reference binary calls — and to exact values. CPU only (the device side of the same header is covered by the
bit-for-bit GPU parity tests)."""
import numpy as np
import pytest


def _ulps(a, b):
    ia = a.view(np.int64).copy(); ib = b.view(np.int64).copy()
    ia[ia < 0] = np.iinfo(np.int64).min - ia[ia < 0]
    ib[ib < 0] = np.iinfo(np.int64).min - ib[ib < 0]
    return np.abs(ia - ib)


@pytest.mark.parametrize("scale", [1e-3, 1.0, 10.0, 1e3, 8e5])
def test_sincos_within_one_ulp_of_glibc(oracle, scale):
    rng = np.random.default_rng(7)
    x = rng.uniform(-scale, scale, 400_000)
    for kind in ("sin", "cos"):
        d = oracle.math_eval(kind, x, impl=0); g = oracle.math_eval(kind, x, impl=1)
        assert np.array_equal(g, getattr(np, kind)(x)) or np.max(_ulps(g, getattr(np, kind)(x))) <= 1   # glibc through the C entry point
        assert _ulps(d, g).max() <= 1, (kind, scale)
        assert (_ulps(d, g) > 0).mean() < 0.1


def test_sincos_special_values_and_identities(oracle):
    x = np.array([0.0, -0.0, 1e-300, 5e-324, np.pi / 4, np.pi / 2, np.pi, 3 * np.pi / 2, 2 * np.pi, 1e6, -1e6])
    s = oracle.math_eval("sin", x); c = oracle.math_eval("cos", x)
    assert s[0] == 0.0 and c[0] == 1.0 and np.signbit(s[1]) and s[2] == 1e-300 and s[3] == 5e-324
    assert np.abs(s * s + c * c - 1).max() < 4e-16
    assert np.abs(s - np.sin(x)).max() < 1e-15 and np.abs(c - np.cos(x)).max() < 1e-15
    assert np.array_equal(oracle.math_eval("sin", -x), -s) and np.array_equal(oracle.math_eval("cos", -x), c)     # odd / even, bit for bit
    bad = oracle.math_eval("sin", np.array([np.inf, -np.inf, np.nan]))
    assert np.isnan(bad).all()
    # beyond 2^50 the spacing of doubles leaves no phase: defined as (0, 1); large arguments below that keep ~2^-60 |x| absolute accuracy
    assert oracle.math_eval("sin", np.array([2.0 ** 50]))[0] == 0.0 and oracle.math_eval("cos", np.array([2.0 ** 60]))[0] == 1.0
    big = np.array([1e7, 3e9, 1e12, 1e15])   # beyond 2^19 pi/2 the reduction loses accuracy gradually (~1e-26 x^2 absolute)
    assert np.abs(oracle.math_eval("sin", big[:3]) - np.sin(big[:3])).max() < 1e-13 and abs(oracle.math_eval("sin", big[3:])[0] - np.sin(1e15)) < 1e-9


@pytest.mark.parametrize("lo,hi", [(-1e-3, 1e-3), (-1, 1), (-40, 40), (-700, 700), (-745, -700)])
def test_exp_within_one_ulp_of_glibc(oracle, lo, hi):
    rng = np.random.default_rng(11)
    x = rng.uniform(lo, hi, 400_000)
    d = oracle.math_eval("exp", x, impl=0); g = oracle.math_eval("exp", x, impl=1)
    assert _ulps(d, g).max() <= 1


def test_exp_special_values(oracle):
    x = np.array([0.0, -0.0, 709.78, 709.79, -745.1, -745.2, np.inf, -np.inf, 1e-10])
    e = oracle.math_eval("exp", x)
    assert e[0] == 1.0 and e[1] == 1.0 and np.isfinite(e[2]) and np.isinf(e[3]) and e[4] == 5e-324 and e[5] == 0.0 and np.isinf(e[6]) and e[7] == 0.0
    assert e[8] == 1.0 + 1e-10
    assert np.isnan(oracle.math_eval("exp", np.array([np.nan]))[0])
    # the CSTR's Arrhenius range (cstr_control_test.cpp:86-88): E / (273.15 + T), T in [50, 200]
    T = np.linspace(50, 200, 10001); a = -9758.3 / (273.15 + T)
    assert _ulps(oracle.math_eval("exp", a, impl=0), oracle.math_eval("exp", a, impl=1)).max() <= 1


def test_sqp_trajectories_agree_between_the_two_function_sets(oracle):
    """The benchmark stream solved with glibc's functions and with the shared restatement: identical SQP / ADMM iteration counts and statuses,
    solutions within 1e-8 (north_star's tolerance) — last-bit differences in sin / cos do not change any discrete decision here."""
    from polympc_amd import workloads
    B = 256
    wl = workloads.robot_batch(B)
    ss = oracle.sqp_default_settings(); ss.max_iter = 10; ss.line_search_max_iter = 10
    run = lambda: oracle.sqp_solve_batch(oracle.MODEL_ROBOT, 6, 1, 0.0, 2.0, B, wl["d"], wl["lbx"], wl["ubx"], sqp_settings=ss, pivot=oracle.PIVOT_EIGEN, threads=8)
    old = oracle.set_libm(True)
    try:
        xg, lg, ig = run()
        oracle.set_libm(False)
        xd, ld, id_ = run()
    finally:
        oracle.set_libm(old)
    assert [i.iter for i in ig] == [i.iter for i in id_] and [i.qp_solver_iter for i in ig] == [i.qp_solver_iter for i in id_]
    assert [i.status for i in ig] == [i.status for i in id_]
    assert np.abs(xg - xd).max() <= 1e-8
