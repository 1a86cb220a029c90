"""Pins the ORACLE (oracle/, the CPU restatement of the reference algorithm) against every golden vector and
known-answer test the reference holds for the hot path. CPU only."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))   # oracle.cross_order (the shared cross-order record)

GOLD = os.path.join(os.path.dirname(__file__), "golden", "casadi_robot_P5S2.npz")
inf = np.inf


@pytest.fixture(autouse=True, params=["glibc", "detmath"])
def transcendental_functions(request, oracle):
    """Every pin runs twice: with glibc's sin / cos / exp (what the reference binary calls) and with the IEEE-only restatement the HIP
    kernels share (polympc_amd/csrc/pmpc_math.hpp, pinned to glibc within 1 ulp in tests/test_math_pins.py) — the known answers hold
    for both, so the GPU-vs-oracle comparisons (which use the shared functions to be bit for bit) are pinned to the reference too."""
    old = oracle.set_libm(request.param == "glibc")
    yield request.param
    oracle.set_libm(old)


# ---------------------------------------------------------------- A1: Chebyshev constants
@pytest.mark.parametrize("P", [2, 3, 4, 5, 6, 7, 8, 15])
def test_cheb_closed_forms(oracle, P):
    nodes, w, D = oracle.cheb(P)
    assert np.allclose(nodes, np.cos(np.pi * np.arange(P + 1) / P), atol=1e-15)
    assert abs(D[0, 0] - (2 * P * P + 1) / 6.0) < 1e-12
    assert abs(D[P, P] + D[0, 0]) < 1e-12
    assert np.allclose(D, -D[::-1, ::-1], atol=1e-11)          # centro-antisymmetry (continuous_ocp.hpp:845-846)
    assert np.allclose(D.sum(axis=1), 0, atol=1e-12)
    assert abs(w.sum() - 2.0) < 1e-13
    w0 = 1.0 / (P * P) if P % 2 else 1.0 / (P * P - 1)
    assert abs(w[0] - w0) < 1e-15 and abs(w[P] - w0) < 1e-15
    # D differentiates polynomials of degree <= P exactly
    for deg in range(P + 1):
        assert np.allclose(D @ nodes ** deg, deg * nodes ** max(deg - 1, 0) if deg else 0 * nodes, atol=1e-9)
    # Clenshaw–Curtis integrates even monomials of degree <= P exactly
    for deg in range(0, P + 1, 2):
        assert abs(w @ nodes ** deg - 2.0 / (deg + 1)) < 1e-12


def test_cheb_table_values(oracle):
    n5, w5, D5 = oracle.cheb(5)
    assert np.allclose(w5[:3], [0.04, 0.360743041200011, 0.599256958799989], atol=1e-14)
    assert np.allclose(D5[0], [8.5, -10.472135954999581, 2.894427190999916, -1.527864045000421, 1.105572809000084, -0.5], atol=1e-12)
    n6, w6, D6 = oracle.cheb(6)
    assert np.allclose(w6[:4], [0.028571428571429, 0.253968253968254, 0.457142857142857, 0.520634920634921], atol=1e-14)
    assert np.allclose(D6[0], [12.166666666666671, -14.928203230275516, 4, -2, 1.333333333333333, -1.071796769724491, 0.5], atol=1e-12)


# ---------------------------------------------------------------- A2/A4/A6/A7/A8/A9/A10 vs the reference's fixtures
def test_collocation_against_reference_golden(oracle):
    g = np.load(GOLD)
    for k in range(len(g["x"])):
        x, lam = g["x"][k], g["lam"][k]
        ev = oracle.ocp_eval(oracle.MODEL_ROBOT, 5, 2, 0.0, 1.0, x, [1.0], lam=np.concatenate([lam, np.zeros(55)]))
        assert abs(ev["cost"] - g["cost"][k]) <= 1e-14 * max(1, abs(g["cost"][k]))
        assert np.abs(ev["c"] - g["c"][k]).max() <= 1e-13
        assert np.abs(ev["jac"] - g["jac"][k]).max() <= 1e-13
        assert np.abs(ev["cost_grad"] - g["cost_grad"][k]).max() <= 1e-13
        assert np.abs(ev["cost_hess"] - g["cost_hess"][k]).max() <= 1e-13
        assert np.abs(ev["lag_grad"] - g["lag_grad"][k]).max() <= 1e-13      # box multipliers are zero here
        assert np.abs(ev["lag_hess"] - g["lag_hess"][k]).max() <= 1e-13
        assert abs((ev["cost"] + lam @ ev["c"]) - g["lag"][k]) <= 1e-13


def test_collocation_against_live_reference_build(oracle):
    if not os.path.exists(oracle.REF_PATH):
        pytest.skip("oracle/_ref not built (reference tree absent)")
    r = oracle.RefCasadiRobot()
    x = np.random.default_rng(3).uniform(-2, 2, 55)
    ev = oracle.ocp_eval(oracle.MODEL_ROBOT, 5, 2, 0.0, 1.0, x, [1.0])
    assert abs(ev["cost"] - r.cost(x)) < 1e-13
    assert np.abs(ev["c"] - r.constraint(x)).max() < 1e-13


def test_dense_layout_parking_np1(oracle):
    """dense_sparse_compare.cpp:151-172 test point: NP=1 border blocks; finite-difference cross-check of J and H."""
    var = np.array([1.5, 0.5, 0.5] * 11 + [0.0] * 22 + [0.5])
    var[33:55] = 0.1 * np.arange(22) - 0.7
    dm = oracle.ocp_dims(oracle.MODEL_PARKING, 5, 2)
    assert (dm["n"], dm["m"]) == (56, 33)
    lam = np.concatenate([np.linspace(-1, 1, 33), np.zeros(56)])
    ev = oracle.ocp_eval(oracle.MODEL_PARKING, 5, 2, 0.0, 1.0, var, [2.0], lam=lam)
    eps = 1e-6
    Jfd = np.zeros((33, 56)); Hfd = np.zeros((56, 56))
    for j in range(56):
        e = np.zeros(56); e[j] = eps
        ep = oracle.ocp_eval(oracle.MODEL_PARKING, 5, 2, 0.0, 1.0, var + e, [2.0], lam=lam)
        em = oracle.ocp_eval(oracle.MODEL_PARKING, 5, 2, 0.0, 1.0, var - e, [2.0], lam=lam)
        Jfd[:, j] = (ep["c"] - em["c"]) / (2 * eps)
        Hfd[:, j] = (ep["lag_grad"] - em["lag_grad"]) / (2 * eps)
    assert np.abs(ev["jac"] - Jfd).max() < 1e-7
    assert np.abs(ev["lag_hess"] - Hfd).max() < 1e-6
    assert abs(ev["cost"] - 0.5) < 1e-15                      # Mayer = p


# ---------------------------------------------------------------- A17: bound classification (admm_solver_test.cpp:259-301)
def test_constraint_classification(oracle):
    INEQ, EQ, LOOSE = 0, 1, 2
    assert oracle.classify(-1e17, 1e17) == LOOSE
    assert oracle.classify(-101, 1e17) == INEQ
    assert oracle.classify(-1e17, 123) == INEQ
    assert oracle.classify(-1, 1) == INEQ
    assert oracle.classify(42, 42) == EQ
    assert oracle.classify(-inf, inf) == LOOSE


# ---------------------------------------------------------------- A16: boxADMM known answers (box_admm_test.cpp)
def _simple_qp():
    H = np.array([[4.0, 1.0], [1.0, 2.0]]).T.ravel()[None]
    return H, np.array([[1.0, 1.0]]), np.array([[1.0, 1.0]]), [[1.0]], [[1.0]], [[0.0, 0.0]], [[0.7, 0.7]]


def _is_approx(a, b, prec):
    return np.linalg.norm(a - b) <= prec * min(np.linalg.norm(a), np.linalg.norm(b))


@pytest.mark.parametrize("pivot", [0, 1, 2])
def test_boxadmm_simple_qp(oracle, pivot):  # :15-45
    s = oracle.qp_default_settings(); s.max_iter = 150
    x, y, info = oracle.qp_solve_batch(*_simple_qp(), settings=s, pivot=pivot)
    assert _is_approx(x[0], np.array([0.3, 0.7]), 1e-2)
    assert info[0].iter < 150 and info[0].status == oracle.QP_SOLVED


@pytest.mark.parametrize("pivot", [0, 1, 2])
def test_boxadmm_constraint_violation(oracle, pivot):  # :117-155
    s = oracle.qp_default_settings(); s.eps_rel = 1e-4; s.eps_abs = 1e-4
    x, y, info = oracle.qp_solve_batch(*_simple_qp(), settings=s, pivot=pivot)
    sol = x[0]
    lower = np.array([sol.sum() - 1, sol[0], sol[1]]); upper = np.array([sol.sum() - 1, sol[0] - 0.7, sol[1] - 0.7])
    assert lower.min() >= -1e-3 and upper.max() <= 1e-3


def test_boxadmm_adaptive_rho_helps(oracle):  # :190-231
    s = oracle.qp_default_settings(); s.max_iter = 1000; s.rho = 0.1; s.adaptive_rho = 0
    _, _, i0 = oracle.qp_solve_batch(*_simple_qp(), settings=s)
    s.adaptive_rho = 1; s.adaptive_rho_interval = 10
    _, _, i1 = oracle.qp_solve_batch(*_simple_qp(), settings=s)
    assert i1[0].iter < 1000 and i1[0].iter < i0[0].iter and i1[0].status == oracle.QP_SOLVED


@pytest.mark.parametrize("pivot", [0, 1, 2])
def test_boxadmm_ruiz_equilibration(oracle, pivot):  # box_admm_test.cpp:47-83 — the reference's known-answer test of RuizEquilibration
    s = oracle.qp_default_settings(); s.max_iter = 150
    H, h, A, al, au, xl, xu, D, E, c = oracle.ruiz_compute_batch(*_simple_qp())
    x, y, info = oracle.qp_solve_batch(H, h, A, al, au, xl, xu, settings=s, pivot=pivot)
    sol, dual = oracle.ruiz_unscale_solution_batch(D, E, c, x, y)
    assert _is_approx(sol[0], np.array([0.3, 0.7]), 1e-2)
    assert info[0].iter < 150 and info[0].status == oracle.QP_SOLVED


def test_ruiz_scaling_is_a_congruence(oracle):
    """compute() returns D, E, c with  H_s = c D H D,  A_s = E A D,  h_s = c D h,  bounds scaled by E and 1/D
    (qp_preconditioners.hpp:197-232), and a badly scaled problem leaves the loop after one sweep (the loop condition
    tests the norm measured BEFORE the sweep, :178,:187)."""
    from polympc_amd import workloads
    n, m, B = 7, 4, 6
    qp = workloads.random_qp_batch(B, n, m, seed=3)
    H0 = qp["H"].reshape(B, n, n).transpose(0, 2, 1) * 50.0; A0 = qp["A"].reshape(B, n, m).transpose(0, 2, 1)
    Hc = np.ascontiguousarray(H0.transpose(0, 2, 1)).reshape(B, n * n)
    H, h, A, al, au, xl, xu, D, E, c = oracle.ruiz_compute_batch(Hc, qp["h"], qp["A"], qp["Alb"], qp["Aub"], qp["xlb"], qp["xub"])
    Hs = H.reshape(B, n, n).transpose(0, 2, 1); As = A.reshape(B, n, m).transpose(0, 2, 1)
    for b in range(B):
        assert np.allclose(Hs[b], c[b] * (D[b][:, None] * H0[b] * D[b][None, :]), rtol=1e-13)
        assert np.allclose(As[b], E[b][:, None] * A0[b] * D[b][None, :], rtol=1e-13)
        assert np.allclose(h[b], c[b] * D[b] * qp["h"][b], rtol=1e-13)
        fin = np.isfinite(qp["Alb"][b]); assert np.allclose(al[b][fin], (qp["Alb"][b] * E[b])[fin], rtol=1e-14)
        fin = np.isfinite(qp["xub"][b]); assert np.allclose(xu[b][fin], (qp["xub"][b] / D[b])[fin], rtol=1e-14)
        # one sweep only: D = 1/sqrt(column norms of the ORIGINAL matrices)
        d1 = 1.0 / np.sqrt(np.maximum(np.abs(H0[b]).max(axis=0), np.abs(A0[b]).max(axis=0)))
        assert np.allclose(D[b], d1, rtol=1e-14)


def test_sqp_with_ruiz_preconditioner(oracle):
    """SQPBase<..., RuizEquilibration> (sqp_base.hpp:605-611): scaling the QP must not change what the SQP converges to."""
    ss = oracle.sqp_default_settings(); ss.max_iter = 10; ss.line_search_max_iter = 10
    lbx, ubx = _robot_bounds(7, [0.5, 0.5, 0.5])
    x0, _, i0 = oracle.sqp_solve_batch(oracle.MODEL_ROBOT, 6, 1, 0.0, 2.0, 1, [[2.0]], lbx, ubx, sqp_settings=ss)
    ss.preconditioner = 1
    x1, _, i1 = oracle.sqp_solve_batch(oracle.MODEL_ROBOT, 6, 1, 0.0, 2.0, 1, [[2.0]], lbx, ubx, sqp_settings=ss)
    assert i0[0].status == oracle.SQP_SOLVED and i1[0].status == oracle.SQP_SOLVED
    assert np.abs(x0 - x1).max() < 5e-3


def test_boxadmm_simple_lp(oracle):  # :266-297
    s = oracle.qp_default_settings(); s.max_iter = 200; s.alpha = 1.0; s.adaptive_rho = 1; s.check_termination = 10
    z = np.zeros((1, 0))
    x, y, info = oracle.qp_solve_batch(np.zeros((1, 1)), np.ones((1, 1)), z, z, z, [[-1e6]], [[1e6]], settings=s)
    assert _is_approx(x[0], np.array([-1e6]), 1e-2) and info[0].iter < 200 and info[0].status == oracle.QP_SOLVED


def test_boxadmm_nonconvex(oracle):  # :299-334
    s = oracle.qp_default_settings(); s.max_iter = 200; s.alpha = 1.0; s.adaptive_rho = 1; s.rho = 2; s.check_termination = 10
    z = np.zeros((1, 0))
    x, y, info = oracle.qp_solve_batch(-np.ones((1, 1)), np.zeros((1, 1)), z, z, z, [[-1.0]], [[2.0]], settings=s,
                                       x0=[[0.1]], y0=[[0.1]])
    assert _is_approx(x[0], np.array([2.0]), 1e-2) and info[0].iter < 200 and info[0].status == oracle.QP_SOLVED


# ---------------------------------------------------------------- §8f-4: the OSQP-style ADMM solver (admm_solver_test.cpp)
@pytest.mark.parametrize("pivot", [0, 1, 3])
def test_admm_simple_qp(oracle, pivot):  # admm_solver_test.cpp:16-45
    s = oracle.qp_default_settings(); s.max_iter = 1000
    x, y, info = oracle.qp_admm_solve_batch(*_simple_qp(), settings=s, pivot=pivot)
    assert _is_approx(x[0], np.array([0.3, 0.7]), 1e-2) and info[0].iter < 1000 and info[0].status == oracle.QP_SOLVED


def test_admm_ruiz_equilibration(oracle):  # :47-82
    s = oracle.qp_default_settings(); s.max_iter = 1000
    H, h, A, al, au, xl, xu, D, E, c = oracle.ruiz_compute_batch(*_simple_qp())
    x, y, info = oracle.qp_admm_solve_batch(H, h, A, al, au, xl, xu, settings=s)
    sol, dual = oracle.ruiz_unscale_solution_batch(D, E, c, x, y)
    assert _is_approx(sol[0], np.array([0.3, 0.7]), 1e-2) and info[0].iter < 1000 and info[0].status == oracle.QP_SOLVED


def test_admm_constraint_violation(oracle):  # :114-151
    s = oracle.qp_default_settings(); s.eps_rel = 1e-4; s.eps_abs = 1e-4
    x, y, info = oracle.qp_admm_solve_batch(*_simple_qp(), settings=s)
    sol = x[0]
    lower = np.array([sol.sum() - 1, sol[0], sol[1]]); upper = np.array([sol.sum() - 1, sol[0] - 0.7, sol[1] - 0.7])
    assert lower.min() >= -1e-3 and upper.max() <= 1e-3


def test_admm_adaptive_rho(oracle):  # :153-183 (default settings solve it) and :185-225 (adaptive rho needs fewer iterations)
    s = oracle.qp_default_settings(); s.adaptive_rho = 0; s.adaptive_rho_interval = 10
    _, _, i = oracle.qp_admm_solve_batch(*_simple_qp(), settings=s)
    assert i[0].status == oracle.QP_SOLVED
    s = oracle.qp_default_settings(); s.max_iter = 1000; s.rho = 0.1; s.adaptive_rho = 0
    _, _, i0 = oracle.qp_admm_solve_batch(*_simple_qp(), settings=s)
    s.adaptive_rho = 1; s.adaptive_rho_interval = 10
    _, _, i1 = oracle.qp_admm_solve_batch(*_simple_qp(), settings=s)
    assert i1[0].iter < 1000 and i1[0].iter < i0[0].iter and i1[0].status == oracle.QP_SOLVED


def test_admm_lp_and_nonconvex(oracle):  # :303-334, :336-371
    z = np.zeros((1, 0))
    s = oracle.qp_default_settings(); s.max_iter = 200; s.alpha = 1.0; s.adaptive_rho = 1; s.check_termination = 10
    x, y, info = oracle.qp_admm_solve_batch(np.zeros((1, 1)), np.ones((1, 1)), z, z, z, [[-1e6]], [[1e6]], settings=s)
    assert _is_approx(x[0], np.array([-1e6]), 1e-2) and info[0].iter < 200 and info[0].status == oracle.QP_SOLVED
    s.rho = 2
    x, y, info = oracle.qp_admm_solve_batch(-np.ones((1, 1)), np.zeros((1, 1)), z, z, z, [[-1.0]], [[2.0]], settings=s, x0=[[0.1]], y0=[[0.1]])
    assert _is_approx(x[0], np.array([2.0]), 1e-2) and info[0].iter < 200 and info[0].status == oracle.QP_SOLVED


def test_ldlt_policies_agree(oracle):
    rng = np.random.default_rng(5)
    n, m = 12, 7
    G = rng.normal(size=(n, n)); H = G @ G.T + 0.1 * np.eye(n); A = rng.normal(size=(m, n))
    K = np.block([[H, A.T], [A, -np.diag(rng.uniform(0.01, 10, m))]])
    b = rng.normal(size=n + m)
    x_ref = np.linalg.solve(K, b)
    for piv in (0, 1, 2):
        x = oracle.ldlt_solve(np.tril(K), b, piv)          # only the lower triangle is read
        assert np.abs(x - x_ref).max() < 1e-9


@pytest.mark.parametrize("n,m", [(35, 21), (25, 15), (5, 3), (40, 24)])
def test_sweep_inverse_policy_matches_factorisations(oracle, n, m):
    """The swept-inverse restatement (block-lower 16x16 tile storage, blocks of 8 pivots) against numpy and against both LDL^T
    policies on quasi-definite KKT matrices with the conditioning of the ADMM (sigma = 1e-6, rho between 1e-1 and 1e2);
    sizes cover n+m = 56 (config A), 40 (5-node grids), one partial block (8) and the 64-row limit."""
    rng = np.random.default_rng(n * 100 + m)
    G = rng.normal(size=(n, n)); H = G @ G.T / n + (1e-6 + 0.1) * np.eye(n); A = rng.normal(size=(m, n))
    K = np.block([[H, A.T], [A, -np.diag(1.0 / rng.choice([0.1, 100.0], m))]])
    b = rng.normal(size=n + m)
    x_ref = np.linalg.solve(K, b)
    scale = np.abs(x_ref).max()
    xs = oracle.ldlt_solve(np.tril(K), b, oracle.PIVOT_SWEEP)
    assert np.abs(xs - x_ref).max() < 1e-9 * scale
    for piv in (oracle.PIVOT_EIGEN, oracle.PIVOT_STATIC):
        assert np.abs(xs - oracle.ldlt_solve(np.tril(K), b, piv)).max() < 1e-9 * scale


# ---------------------------------------------------------------- A12: BFGS (bfgs_test.cpp:21-66)
@pytest.mark.parametrize("Htrue,check_converged", [(np.diag([2.0, 1.0]), True), (np.diag([2.0, -1.0]), False)])
def test_bfgs(oracle, Htrue, check_converged):
    B = np.eye(2)
    for i in range(10):
        step = np.array([np.sin(i), np.cos(i)])
        B = oracle.bfgs(B, step, Htrue @ step)
        assert np.all(np.linalg.eigvalsh((B + B.T) / 2) > 0)
    if check_converged:
        assert _is_approx(B, Htrue, 1e-3)


def test_regularisers(oracle):
    H = np.array([[1.0, 2.0, 0.0], [2.0, -3.0, 0.5], [0.0, 0.5, 0.2]])
    Hm = oracle.regularise(1, H)
    w, V = np.linalg.eigh(H)
    w2 = np.where(w <= 0, -w + 0.1, w)
    assert np.allclose(Hm, V @ np.diag(w2) @ V.T, atol=1e-10)
    Hg = oracle.regularise(2, H)
    assert np.all(np.linalg.eigvalsh(Hg) > 0)
    exp = H.copy()
    for i in range(3):
        ri = np.abs(H[:, i]).sum() - abs(H[i, i])
        if H[i, i] - ri <= 0:
            exp[i, i] += (ri - H[i, i]) + 0.01
    assert np.allclose(Hg, exp)


# ---------------------------------------------------------------- A14/A15: SQP end to end (sqp_test_autodiff.cpp)
def _nlp_settings(oracle):
    ss = oracle.sqp_default_settings(); ss.max_iter = 50; ss.line_search_max_iter = 5; ss.regularisation = 1
    return ss


@pytest.mark.parametrize("pivot", [0, 1, 3])
def test_sqp_constrained_rosenbrock(oracle, pivot):  # :78-97
    x, lam, info = oracle.nlp_solve(oracle.NLP_CONSTRAINED_ROSENBROCK, [2.01, 1.01], sqp_settings=_nlp_settings(oracle), pivot=pivot)
    assert _is_approx(x, np.array([0.7864, 0.6177]), 1e-2) and info.iter < 50


@pytest.mark.parametrize("pivot", [0, 1, 3])
def test_sqp_rosenbrock(oracle, pivot):  # :119-137
    x, lam, info = oracle.nlp_solve(oracle.NLP_ROSENBROCK, [2.01, 1.01], sqp_settings=_nlp_settings(oracle), pivot=pivot)
    assert _is_approx(x, np.array([1.0, 1.0]), 1e-2) and info.iter < 50


@pytest.mark.parametrize("pivot", [0, 1, 3])
def test_sqp_simple_nlp(oracle, pivot):  # :165-186
    x, lam, info = oracle.nlp_solve(oracle.NLP_SIMPLE, [1.0, 1.0], lbg=[1.0], ubg=[2.0], sqp_settings=_nlp_settings(oracle), pivot=pivot)
    assert _is_approx(x, np.array([1.0, 1.0]), 1e-2) and info.iter < 50


@pytest.mark.parametrize("pivot", [0, 1, 3])
def test_sqp_hs071_solution(oracle, pivot):  # :223-246
    x, lam, info = oracle.nlp_solve(oracle.NLP_HS071, [1.0, 5.0, 5.0, 1.0], lbx=[1.0] * 4, ubx=[5.0] * 4, lbg=[25.0], ubg=[inf],
                                    sqp_settings=_nlp_settings(oracle), pivot=pivot)
    assert _is_approx(x, np.array([1.0, 4.74299963, 3.82114998, 1.37940829]), 1e-2)


def test_sqp_hs071_iteration_bound_is_a_last_bit_property(oracle):
    """sqp_test_autodiff.cpp:245 also asserts `iter < 50`. On HS071 every QP stops at the SQP constructor's cap of 100 ADMM iterations
    (sqp_base.hpp:87) with inexact multipliers, the dual step norm then hovers around the 1e-3 threshold, and WHEN the termination test fires is
    decided by rounding: perturbing the first coordinate of the start point by 1e-13 — or choosing another, equally valid, elimination order —
    moves the count of the restatement between 32 and more than 100, every run ending at the same optimum. The reference binary sits at one point
    of that ensemble (Eigen's own summation orders in LDLT, EigenSolver and its products, none of which is available here); the restatement's
    unperturbed Eigen-order run needs 55. What CAN be pinned: the optimum is reached by every member, the spread straddles the reference's bound,
    and members that satisfy `iter < 50` exist for the Eigen-style order itself."""
    sol = np.array([1.0, 4.74299963, 3.82114998, 1.37940829])
    ss = oracle.sqp_default_settings(); ss.max_iter = 200; ss.line_search_max_iter = 5; ss.regularisation = 1
    iters = {}
    for pivot in (oracle.PIVOT_EIGEN, oracle.PIVOT_STATIC, oracle.PIVOT_SWEEP1):
        for dx in (0.0, 1e-13, -1e-13):
            x, lam, info = oracle.nlp_solve(oracle.NLP_HS071, [1.0 + dx, 5.0, 5.0, 1.0], lbx=[1.0] * 4, ubx=[5.0] * 4, lbg=[25.0], ubg=[inf],
                                            sqp_settings=ss, pivot=pivot)
            assert info.status == oracle.SQP_SOLVED and _is_approx(x, sol, 1e-2), (pivot, dx)
            iters[(pivot, dx)] = info.iter
    # (round 5) with every linear solve of the ADMM carried to exact arithmetic (PIVOT_EXACT) the unperturbed run takes 41 iterations — inside the reference's
    # bound — and its 1e-13 neighbours 64: even then the count is decided by rounding elsewhere (the eigenvalue mirroring, the products), not by the algorithm
    x, lam, info = oracle.nlp_solve(oracle.NLP_HS071, [1.0, 5.0, 5.0, 1.0], lbx=[1.0] * 4, ubx=[5.0] * 4, lbg=[25.0], ubg=[inf], sqp_settings=ss, pivot=oracle.PIVOT_EXACT)
    assert info.status == oracle.SQP_SOLVED and _is_approx(x, sol, 1e-2) and info.iter < 50
    eig = [v for (p, _), v in iters.items() if p == oracle.PIVOT_EIGEN]
    assert min(eig) < 50, iters                                # the reference's bound is met inside the Eigen-order ensemble ...
    assert min(iters.values()) < 50 <= max(iters.values()), iters   # ... and the ensemble straddles it: the bound is not a property of the algorithm


def _robot_bounds(nn, x0):
    n = 5 * nn
    lbx = np.full(n, -inf); ubx = np.full(n, inf)
    lbx[3 * nn - 3:3 * nn] = x0; ubx[3 * nn - 3:3 * nn] = x0
    lbx[3 * nn:] = np.tile([-1.5, -0.75], nn); ubx[3 * nn:] = np.tile([1.5, 0.75], nn)
    return lbx[None], ubx[None]


@pytest.mark.parametrize("pivot", [0, 1, 3])
def test_sqp_codegen_robot(oracle, pivot):  # codegen_test.cpp:402-438 — exact Hessian every iteration, qp max_iter 1000
    ss = oracle.sqp_default_settings(); ss.max_iter = 10; ss.line_search_max_iter = 10; ss.exact_hessian_every_iter = 1
    qs = oracle.sqp_qp_default_settings(); qs.max_iter = 1000
    lbx, ubx = _robot_bounds(11, [0.5, 0.5, 0.5])
    x, lam, info = oracle.sqp_solve_batch(oracle.MODEL_ROBOT, 5, 2, 0.0, 1.0, 1, [[1.0]], lbx, ubx, sqp_settings=ss, qp_settings=qs, pivot=pivot)
    assert info[0].status == oracle.SQP_SOLVED and info[0].iter < 10


@pytest.mark.parametrize("pivot", [0, 1, 3])
def test_sqp_robot_mpc_warm_start(oracle, pivot):  # mpc_wrapper_test.cpp:120-166 (dense BFGS variant)
    ss = oracle.sqp_default_settings(); ss.max_iter = 10; ss.line_search_max_iter = 10
    lbx, ubx = _robot_bounds(16, [0.5, 0.5, 0.5])
    kw = dict(sqp_settings=ss, pivot=pivot, mparams=[2.0])
    x, lam, i1 = oracle.sqp_solve_batch(oracle.MODEL_ROBOT, 5, 3, 0.0, 2.0, 1, [[2.0]], lbx, ubx, **kw)
    assert i1[0].status == oracle.SQP_SOLVED
    lbx2, ubx2 = _robot_bounds(16, [0.3, 0.4, 0.5])
    x2, lam2, i2 = oracle.sqp_solve_batch(oracle.MODEL_ROBOT, 5, 3, 0.0, 2.0, 1, [[2.0]], lbx2, ubx2, x_guess=x, lam_guess=lam, **kw)
    assert i2[0].status == oracle.SQP_SOLVED and i2[0].iter < i1[0].iter
    # initial condition is honoured on the LAST nx entries of the x block (mpc_wrapper.hpp:89-93)
    assert np.abs(x2[0, 45:48] - [0.3, 0.4, 0.5]).max() < 1e-3
    assert np.all(x2[0, 48:] <= np.tile([1.5, 0.75], 16) + 1e-3) and np.all(x2[0, 48:] >= -np.tile([1.5, 0.75], 16) - 1e-3)


def _minimal_time_parking(nn=11):
    """minimal_time_test.cpp:146-184 (polympc_amd/workloads.py minimal_time_parking)"""
    from polympc_amd import workloads
    return workloads.minimal_time_parking(nn)


def _parking_batch(B, nn=11, seed=5):
    """B minimal-time parking problems (NP = 1) around the reference's (polympc_amd/workloads.py parking_batch)"""
    from polympc_amd import workloads
    return workloads.parking_batch(B, nn, seed)


def test_bordered_block_structured_solve_against_a_dense_solve(oracle):
    """PIVOT_SCHUR with one parameter (round 5): the KKT matrix of a collocation QP whose Hessian has the arrow shape (node blocks, a border row / column, a corner)
    and whose A has a dense parameter column — the bordered range-space solve of the block-structured kernel against numpy's LU of the full matrix."""
    rng = np.random.default_rng(0)
    nx, nu, nn, P = 3, 2, 11, 5
    d = nx + nu; n0 = d * nn; n = n0 + 1; m = nx * nn
    sg = lambda k, c: k * nx + c if c < nx else nx * nn + k * nu + (c - nx)
    H = np.zeros((n, n))
    for k in range(nn):
        Bk = rng.standard_normal((d, d)); idx = [sg(k, c) for c in range(d)]
        H[np.ix_(idx, idx)] = Bk @ Bk.T + np.eye(d)
    b = 0.3 * rng.standard_normal(n0); H[n0, :n0] = b; H[:n0, n0] = b; H[n0, n0] = 5.0 + b @ b
    A = np.zeros((m, n))
    for r in range(m):
        ni, si = divmod(r, nx)
        kb = nn - 1 - P if ni == nn - 1 else (ni // P) * P
        for k in range(kb, kb + P + 1): A[r, k * nx + si] = rng.standard_normal()
        for c in range(d): A[r, sg(ni, c)] = rng.standard_normal()
        A[r, n0] = rng.standard_normal()
    for rho in (0.1, 100.0, 1e4):
        rv = np.full(m, rho)
        K = np.zeros((n + m, n + m)); K[:n, :n] = H + np.eye(n) * (1e-6 + 0.1 * rho); K[n:, :n] = A; K[:n, n:] = A.T; K[n:, n:] = -np.diag(1 / rv)
        rhs = rng.standard_normal(n + m)
        sol = oracle.kkt_solve(K, rv, rhs, pivot=oracle.PIVOT_SCHUR, structure=(nx, nu, nn, P, 1))
        ref = np.linalg.solve(K, rhs)
        assert np.abs(sol - ref).max() <= 1e-11 * max(1.0, np.abs(ref).max()), rho
    with pytest.raises(ValueError):
        oracle.kkt_solve(K, rv, rhs, pivot=oracle.PIVOT_SCHUR, structure=(nx, nu, nn, P))   # the structure must name the parameter


def test_sqp_parking_batch_block_structured_order_against_the_reference_order(oracle):
    """The bordered block-structured order (PIVOT_SCHUR, NP = 1) inside the SQP on 12 minimal-time parking problems around the reference's. (a) As
    minimal_time_test.cpp configures it (exact Hessians + Gershgorin, adaptive rho): EVERY instance meets the order's conditioning gate once the ADMM penalty adapts
    upwards — the states hardly enter these dynamics, so the state columns of the collocation Jacobian are nearly singular — and is finished in the static order: the
    outcome, the SQP iteration counts and the solutions (1e-5) of Eigen's pivoted order, flag set. (Without the gate: three instances need one to three more
    iterations and end 2e-3 .. 3e-2 away.) Which is why the product keeps this problem on its dense kernel. (b) With the penalty held at 0.1 nothing trips, and the
    bordered order is CLOSER to exact arithmetic (PIVOT_EXACT) than the reference order is."""
    lbx, ubx, xg, d = _parking_batch(12)
    ss = oracle.sqp_default_settings(); ss.max_iter = 20; ss.line_search_max_iter = 10; ss.regularisation = 2; ss.exact_hessian_every_iter = 1
    run = lambda pv, qs=None: oracle.sqp_solve_batch(oracle.MODEL_PARKING, 5, 2, 0.0, 1.0, 12, d, lbx, ubx, x_guess=xg, sqp_settings=ss, qp_settings=qs, pivot=pv, threads=8)
    xe, _, ie = run(oracle.PIVOT_EIGEN)
    xs, _, is_ = run(oracle.PIVOT_SCHUR)
    assert all(i.flags & oracle.FLAG_ILLCOND for i in is_) and not any(i.flags for i in ie)
    assert [i.status for i in ie] == [i.status for i in is_] and [i.iter for i in ie] == [i.iter for i in is_]
    solved = np.array([i.status == oracle.SQP_SOLVED for i in ie])
    assert solved.sum() >= 10
    assert np.abs(xe[solved] - xs[solved]).max() <= 1e-5 and np.abs(xe[solved, 55] - xs[solved, 55]).max() <= 1e-6
    qs = oracle.sqp_qp_default_settings(); qs.adaptive_rho = 0
    R = {pv: run(pv, qs) for pv in (oracle.PIVOT_EIGEN, oracle.PIVOT_SCHUR, oracle.PIVOT_EXACT)}
    assert not any(i.flags for i in R[oracle.PIVOT_SCHUR][2])
    assert [i.qp_solver_iter for i in R[oracle.PIVOT_SCHUR][2]] == [i.qp_solver_iter for i in R[oracle.PIVOT_EXACT][2]]
    ds = np.abs(R[oracle.PIVOT_SCHUR][0] - R[oracle.PIVOT_EXACT][0]).max(); de = np.abs(R[oracle.PIVOT_EIGEN][0] - R[oracle.PIVOT_EXACT][0]).max()
    assert ds <= 1e-10 and ds <= de, (ds, de)


@pytest.mark.parametrize("cfg", ["cstr_11", "robot_16", "robot_11"])
def test_block_structured_order_under_its_conditioning_gate_follows_exact_arithmetic_like_the_reference_order(oracle, cfg):
    """PIVOT_SCHUR with its gate (max S_ii max |(S^-1)_ii| > 1e7: give up, redo in PIVOT_STATIC) against the same solves carried to exact arithmetic
    (PIVOT_EXACT), next to Eigen's pivoted order, with the block BFGS and the QP penalty started at 0.1 .. 1e3: on every instance the order KEEPS, the
    trajectories are those of exact arithmetic and the solutions lie within 1e-7 (scaled) of it — at worst 16 times the reference order's own distance (CSTR, rho0 = 100: 1e-7 against 6e-9), mostly closer than the reference order; a gate of 3e6 would make that factor 1 but trips on 10 of config R's 2048 instances. Without the gate
    the range-space solve — a difference of quantities ~ rho_eq times its result — is off by 2e-5 (robot, rho0 = 100) to 2e-3 (CSTR, rho0 = 1e3)."""
    from polympc_amd import workloads
    B = 16
    wl = {"cstr_11": workloads.cstr_batch(B), "robot_16": workloads.robot_batch(B, P=5, S=3), "robot_11": workloads.robot_batch(B, P=5, S=2)}[cfg]
    kept = flagged = 0
    for rho0 in (0.1, 1.0, 10.0, 30.0, 100.0, 1e3):
        R = {}
        for pv in (oracle.PIVOT_SCHUR, oracle.PIVOT_EIGEN, oracle.PIVOT_EXACT):
            ss = oracle.sqp_default_settings(); ss.max_iter = wl["max_iter"]; ss.line_search_max_iter = wl["ls_max_iter"]; ss.hessian_update = 1
            qs = oracle.sqp_qp_default_settings(); qs.rho = rho0
            x, _, info = oracle.sqp_solve_batch(wl["model"], wl["P"], wl["S"], wl["t0"], wl["tf"], B, wl["d"], wl["lbx"], wl["ubx"], sqp_settings=ss, qp_settings=qs, pivot=pv, threads=8)
            R[pv] = (x, np.array([i.qp_solver_iter for i in info]), np.array([i.flags for i in info]))
        sc = np.maximum(1.0, np.abs(R[oracle.PIVOT_EXACT][0]).max(axis=0))
        keep = (R[oracle.PIVOT_SCHUR][2] & oracle.FLAG_ILLCOND) == 0
        kept += int(keep.sum()); flagged += int((~keep).sum())
        if rho0 <= 1.0: assert keep.all()
        if not keep.any(): continue
        assert np.array_equal(R[oracle.PIVOT_SCHUR][1][keep], R[oracle.PIVOT_EXACT][1][keep])
        ds = (np.abs(R[oracle.PIVOT_SCHUR][0] - R[oracle.PIVOT_EXACT][0]) / sc).max(axis=1)[keep].max()
        de = (np.abs(R[oracle.PIVOT_EIGEN][0] - R[oracle.PIVOT_EXACT][0]) / sc).max(axis=1)[keep].max()
        assert ds <= 2e-7 and ds <= max(20.0 * de, 1e-9), (rho0, ds, de)
    assert kept >= 3 * B and flagged >= B


@pytest.mark.parametrize("pivot", [0, 1, 3, 7])
def test_sqp_minimal_time_valet_parking(oracle, pivot):  # minimal_time_test.cpp:146-188 — exact Hessian every iteration + Gershgorin
    ss = oracle.sqp_default_settings(); ss.max_iter = 20; ss.line_search_max_iter = 10
    ss.regularisation = 2; ss.exact_hessian_every_iter = 1
    lbx, ubx, xg = _minimal_time_parking()
    x, lam, info = oracle.sqp_solve_batch(oracle.MODEL_PARKING, 5, 2, 0.0, 1.0, 1, [[1.0]], lbx, ubx, x_guess=xg, sqp_settings=ss, pivot=pivot)
    assert info[0].status == oracle.SQP_SOLVED and info[0].iter < 20          # the reference's two assertions (:186-187)
    assert 0.0 < x[0, 55] < 10.0 and np.abs(x[0, 0:3]).max() <= 0.05 + 1e-3   # a time inside its bounds, parked within tolerance


def _ng_settings(st):
    st.max_iter = 20; st.line_search_max_iter = 10; st.regularisation = 2; st.exact_hessian_every_iter = 1
    return st


def _parking_ng(oracle, ubg, pivot, qp_max_iter=100, max_iter=20):
    lbx, ubx, xg = _minimal_time_parking()
    nn = 11
    ss = _ng_settings(oracle.sqp_default_settings()); ss.max_iter = max_iter
    qs = oracle.sqp_qp_default_settings(); qs.max_iter = qp_max_iter
    x, lam, info = oracle.sqp_solve_batch(oracle.MODEL_PARKING_NG, 5, 2, 0.0, 1.0, 1, [[1.0]], lbx, ubx, lbg=np.full((1, nn), -10.0),
                                          ubg=np.full((1, nn), ubg), x_guess=xg, sqp_settings=ss, qp_settings=qs, pivot=pivot)
    u = x[0, 33:55].reshape(nn, 2)
    return x[0], lam[0], info[0], u[:, 0] ** 2 * np.cos(u[:, 1])


@pytest.mark.parametrize("pivot", [0, 1, 3])
def test_sqp_parking_with_nonlinear_path_constraint(oracle, pivot):
    """nonlinear_constraints_test.cpp:159-184 — the minimal-time parking problem with g = u0^2 cos(u1) in [-10, 10] at every node
    (NP = 1 and NG = 1 together; exact linearisation every iteration + Gershgorin, :97-145). The reference program only prints
    its result and asserts nothing. With |u0| <= 1.5 the constraint can never bind (g <= 2.25), so the optimum is that of
    minimal_time_test.cpp. What this restatement does with the reference's settings (SQP 20 iterations, QP 100): every QP stops at
    its iteration cap, after 20 iterations the iterate sits within 0.1 % of the minimal time and 2e-3 of feasibility, and the SQP
    termination test is not met; with a QP budget of 300 the same solve is SOLVED in < 20 iterations. Both are recorded."""
    lbx, ubx, xg = _minimal_time_parking()
    ss = _ng_settings(oracle.sqp_default_settings())
    x0, _, i0 = oracle.sqp_solve_batch(oracle.MODEL_PARKING, 5, 2, 0.0, 1.0, 1, [[1.0]], lbx, ubx, x_guess=xg, sqp_settings=ss, pivot=pivot)
    x, lam, info, g = _parking_ng(oracle, 10.0, pivot)
    assert info.iter == 20 and info.qp_solver_iter == 20 * 101 and info.status == oracle.SQP_MAX_ITER_EXCEEDED
    assert abs(x[55] - x0[0, 55]) <= 1e-3 * x0[0, 55] and np.abs(x[0:3]).max() <= 0.05 + 2e-3 and g.max() <= 2.25 + 1e-9
    x, lam, info, g = _parking_ng(oracle, 10.0, pivot, qp_max_iter=300)
    assert info.status == oracle.SQP_SOLVED and info.iter < 20
    assert abs(x[55] - x0[0, 55]) <= 1e-3 * x0[0, 55]
    assert np.abs(lam[33:44]).max() <= 1e-6                              # rows 33..43 are g: inactive, no multiplier


@pytest.mark.parametrize("pivot", [0, 1, 3])
def test_sqp_parking_active_nonlinear_path_constraint(oracle, pivot):
    """The same problem with the bound tightened until it binds (u0^2 cos(u1) <= 1.2; the free optimum reaches 2.19): SOLVED with
    the reference's settings, g at the bound on the nodes, non-zero multipliers on the g rows, and a longer manoeuvre."""
    free, _, _, gfree = _parking_ng(oracle, 10.0, pivot, qp_max_iter=300)
    x, lam, info, g = _parking_ng(oracle, 1.2, pivot)
    assert gfree.max() > 2.0
    assert info.status == oracle.SQP_SOLVED and info.iter < 20
    assert g.max() <= 1.2 + 1e-3 and g.max() >= 1.2 - 1e-3
    assert x[55] > 1.2 * free[55] and np.abs(lam[33:44]).max() > 1e-2


def test_sweep_policy_rejects_systems_over_64_rows(oracle):
    """PIVOT_SWEEP restates the register-resident kernel (at most 64 KKT rows); larger systems must be refused, not mis-solved."""
    lbx, ubx, xg = _minimal_time_parking()
    with pytest.raises(ValueError):
        oracle.sqp_solve_batch(oracle.MODEL_PARKING, 5, 2, 0.0, 1.0, 1, [[1.0]], lbx, ubx, x_guess=xg, pivot=oracle.PIVOT_SWEEP)


def _valet_bounds(x0):
    nn = 11; n = 5 * nn
    lbx = np.full(n, -inf); ubx = np.full(n, inf)
    lbx[3 * nn:] = np.tile([-1.5, -0.75], nn); ubx[3 * nn:] = np.tile([1.5, 0.75], nn)
    lbx[30:33] = x0; ubx[30:33] = x0
    return lbx[None], ubx[None]


@pytest.mark.parametrize("pivot", [0, 1, 3])
def test_sqp_valet_parking_with_ruiz(oracle, pivot):
    """valet_parking_mpc_test.cpp:183-240 — SQP with the RuizEquilibration preconditioner, QP max_iter 1000, cold solve then a
    warm-started solve from a moved initial state; both must be SOLVED in < 10 iterations. (Variant: the reference's solver
    there also swaps in a filter line search and the block BFGS of ContinuousOCP, which are out of scope; this runs the
    default l1 line search and dense damped BFGS.)"""
    ss = oracle.sqp_default_settings(); ss.max_iter = 10; ss.line_search_max_iter = 10; ss.preconditioner = 1
    qs = oracle.sqp_qp_default_settings(); qs.max_iter = 1000
    kw = dict(sqp_settings=ss, qp_settings=qs, pivot=pivot, mparams=[1.0])
    lbx, ubx = _valet_bounds([0.5, 0.5, 0.5])
    x, lam, i1 = oracle.sqp_solve_batch(oracle.MODEL_ROBOT, 5, 2, 0.0, 2.0, 1, [[2.0]], lbx, ubx, **kw)
    assert i1[0].status == oracle.SQP_SOLVED and i1[0].iter < 10
    lbx, ubx = _valet_bounds([0.3, 0.4, 0.45])
    x2, lam2, i2 = oracle.sqp_solve_batch(oracle.MODEL_ROBOT, 5, 2, 0.0, 2.0, 1, [[2.0]], lbx, ubx, x_guess=x, lam_guess=lam, **kw)
    assert i2[0].status == oracle.SQP_SOLVED and i2[0].iter < 10


@pytest.mark.parametrize("pivot", [0, 1, 3])
def test_sqp_valet_parking_as_the_reference_runs_it(oracle, pivot):
    """valet_parking_mpc_test.cpp:183-240 with every hook that test installs: RuizEquilibration preconditioner, QP max_iter 1000,
    the filter line search on LSFilter with beta = 0.1 (:116-158, :192; line_search.hpp:31-98) and ContinuousOCP's block BFGS
    (:160-165). The filter is a member of the solver, so it is carried from the cold solve into the warm-started one. Both solves
    must be SOLVED in < 10 iterations (:216-217, :238-239)."""
    ss = oracle.sqp_default_settings(); ss.max_iter = 10; ss.line_search_max_iter = 10
    ss.preconditioner = 1; ss.hessian_update = 1; ss.line_search = 1; ss.filter_beta = 0.1
    filt = np.zeros((1, oracle.FILTER_STATE_DOUBLES)); oracle.bind_filter_state(ss, filt)
    qs = oracle.sqp_qp_default_settings(); qs.max_iter = 1000
    kw = dict(sqp_settings=ss, qp_settings=qs, pivot=pivot, mparams=[1.0])
    lbx, ubx = _valet_bounds([0.5, 0.5, 0.5])
    x, lam, i1 = oracle.sqp_solve_batch(oracle.MODEL_ROBOT, 5, 2, 0.0, 2.0, 1, [[2.0]], lbx, ubx, **kw)
    assert i1[0].status == oracle.SQP_SOLVED and i1[0].iter < 10
    assert 1 <= filt[0, 0] <= 10 and filt[0, 1] > 0                        # the filter now holds the accepted (cost, violation) pairs
    lbx, ubx = _valet_bounds([0.3, 0.4, 0.45])
    x2, lam2, i2 = oracle.sqp_solve_batch(oracle.MODEL_ROBOT, 5, 2, 0.0, 2.0, 1, [[2.0]], lbx, ubx, x_guess=x, lam_guess=lam, **kw)
    assert i2[0].status == oracle.SQP_SOLVED and i2[0].iter < 10
    assert np.abs(x2[0, 30:33] - [0.3, 0.4, 0.45]).max() < 1e-3


def test_ls_filter_list_semantics(oracle):
    """LSFilter (line_search.hpp:31-98) through the state the solver exports: a dominated entry is removed when a better point is
    added (below max_depth), the newest pair sits in front, and a full filter drops its oldest entry instead."""
    ss = oracle.sqp_default_settings(); ss.max_iter = 1; ss.line_search_max_iter = 10; ss.line_search = 1; ss.filter_beta = 0.1
    lbx, ubx = _valet_bounds([0.5, 0.5, 0.5])
    # a pre-loaded filter whose single entry is dominated by the start point (cost 0, violation 1.5): replaced, then the step is added in front
    filt = np.zeros((1, oracle.FILTER_STATE_DOUBLES)); filt[0, :3] = [1, 5.0, 9.0]; oracle.bind_filter_state(ss, filt)
    oracle.sqp_solve_batch(oracle.MODEL_ROBOT, 5, 2, 0.0, 2.0, 1, [[2.0]], lbx, ubx, sqp_settings=ss, mparams=[1.0])
    assert filt[0, 0] == 2 and filt[0, 3] == 0.0 and abs(filt[0, 4] - 1.5) < 1e-12 and filt[0, 2] < 1.5
    # a full filter (10 entries that nothing dominates and that block nothing): the oldest entry leaves, the count stays 10
    full = np.zeros((1, oracle.FILTER_STATE_DOUBLES)); full[0, 0] = 10
    for i in range(10): full[0, 1 + 2 * i], full[0, 2 + 2 * i] = 100.0 + i, 100.0 + i
    oracle.bind_filter_state(ss, full)
    oracle.sqp_solve_batch(oracle.MODEL_ROBOT, 5, 2, 0.0, 2.0, 1, [[2.0]], lbx, ubx, sqp_settings=ss, mparams=[1.0])
    assert full[0, 0] == 10 and full[0, 19] == 100.0 + 7 and full[0, 3] == 0.0 and abs(full[0, 4] - 1.5) < 1e-12


@pytest.mark.parametrize("pivot", [0, 1, 3])
def test_sqp_robot_mpc_warm_start_block_bfgs(oracle, pivot):
    """mpc_wrapper_test.cpp:120-166 with the Hessian update that test actually selects (MySolver::hessian_update_impl ->
    ContinuousOCP::hessian_update_impl, the block BFGS of continuous_ocp.hpp:2304-2431): SOLVED, and the warm-started second solve
    needs fewer iterations (:159)."""
    ss = oracle.sqp_default_settings(); ss.max_iter = 10; ss.line_search_max_iter = 10; ss.hessian_update = 1
    lbx, ubx = _robot_bounds(16, [0.5, 0.5, 0.5])
    kw = dict(sqp_settings=ss, pivot=pivot, mparams=[2.0])
    x, lam, i1 = oracle.sqp_solve_batch(oracle.MODEL_ROBOT, 5, 3, 0.0, 2.0, 1, [[2.0]], lbx, ubx, **kw)
    assert i1[0].status == oracle.SQP_SOLVED
    lbx2, ubx2 = _robot_bounds(16, [0.3, 0.4, 0.5])
    x2, lam2, i2 = oracle.sqp_solve_batch(oracle.MODEL_ROBOT, 5, 3, 0.0, 2.0, 1, [[2.0]], lbx2, ubx2, x_guess=x, lam_guess=lam, **kw)
    assert i2[0].status == oracle.SQP_SOLVED and i2[0].iter < i1[0].iter


def test_block_bfgs_keeps_the_hessian_block_diagonal(oracle):
    """Started from a block-diagonal exact Hessian, SQP iterates with hessian_update = 1 must reach the same optimum as dense BFGS
    (the problem is the same) — and the two updates are different algorithms, so the trajectories may differ."""
    ss = oracle.sqp_default_settings(); ss.max_iter = 20; ss.line_search_max_iter = 10
    lbx, ubx = _robot_bounds(7, [0.5, 0.5, 0.5])
    x0, _, i0 = oracle.sqp_solve_batch(oracle.MODEL_ROBOT, 6, 1, 0.0, 2.0, 1, [[2.0]], lbx, ubx, sqp_settings=ss)
    ss.hessian_update = 1
    x1, _, i1 = oracle.sqp_solve_batch(oracle.MODEL_ROBOT, 6, 1, 0.0, 2.0, 1, [[2.0]], lbx, ubx, sqp_settings=ss)
    assert i0[0].status == oracle.SQP_SOLVED and i1[0].status == oracle.SQP_SOLVED
    assert np.abs(x0 - x1).max() < 5e-2


@pytest.mark.parametrize("pivot", [0, 1, 3])
def test_sqp_with_admm_qp_solver(oracle, pivot):
    """Solver<Problem, ADMM<...>> (the admm_solver alias of mpc_wrapper_test.cpp:109-110): the OSQP-form QP solver inside the SQP
    loop reaches the optimum boxADMM reaches, cold and warm-started (mpc_wrapper_test.cpp:120-166's assertions)."""
    ss = oracle.sqp_default_settings(); ss.max_iter = 10; ss.line_search_max_iter = 10
    lbx, ubx = _robot_bounds(16, [0.5, 0.5, 0.5])
    kw = dict(pivot=pivot, mparams=[2.0])
    xb, _, ib = oracle.sqp_solve_batch(oracle.MODEL_ROBOT, 5, 3, 0.0, 2.0, 1, [[2.0]], lbx, ubx, sqp_settings=ss, **kw)
    ss.qp_solver = 1
    x, lam, i1 = oracle.sqp_solve_batch(oracle.MODEL_ROBOT, 5, 3, 0.0, 2.0, 1, [[2.0]], lbx, ubx, sqp_settings=ss, **kw)
    assert i1[0].status == oracle.SQP_SOLVED and np.abs(x - xb).max() < 5e-3
    lbx2, ubx2 = _robot_bounds(16, [0.3, 0.4, 0.5])
    x2, lam2, i2 = oracle.sqp_solve_batch(oracle.MODEL_ROBOT, 5, 3, 0.0, 2.0, 1, [[2.0]], lbx2, ubx2, x_guess=x, lam_guess=lam, sqp_settings=ss, **kw)
    assert i2[0].status == oracle.SQP_SOLVED and i2[0].iter < i1[0].iter


def _cstr_reference_scenario(oracle, pivot, regularisation=0, hessian_update=1, perturb=0.0):
    """cstr_control_test.cpp:137-183 as the reference runs it: MySolver overrides hessian_update_impl with the problem's sparsity-preserving block BFGS
    (:128-132 -> continuous_ocp.hpp:2304-2431; `hessian_update = 1`), max_iter = ls_max = 20, cold solve from x0 = (1, 0.5, 100, 100), then a second
    solve warm-started from its primal / dual solution with x0 = (1.1, 0.508, 100.5, 100.1). `perturb` moves the first coordinate of the cold x0."""
    ss = oracle.sqp_default_settings(); ss.max_iter = 20; ss.line_search_max_iter = 20; ss.regularisation = regularisation
    ss.hessian_update = hessian_update
    n = 66
    lbx = np.full(n, -inf); ubx = np.full(n, inf)
    lbx[40:44] = ubx[40:44] = [1.0 + perturb, 0.5, 100.0, 100.0]
    lbx[44:] = np.tile([3.0, -9000.0], 11); ubx[44:] = np.tile([35.0, 0.0], 11)
    d = np.zeros((1, 1))
    x, lam, i1 = oracle.sqp_solve_batch(oracle.MODEL_CSTR, 5, 2, 0.0, 100.0, 1, d, lbx[None], ubx[None], sqp_settings=ss, pivot=pivot)
    lbx[40:44] = ubx[40:44] = [1.1, 0.508, 100.5, 100.1]
    x2, lam2, i2 = oracle.sqp_solve_batch(oracle.MODEL_CSTR, 5, 2, 0.0, 100.0, 1, d, lbx[None], ubx[None], x_guess=x, lam_guess=lam, sqp_settings=ss, pivot=pivot)
    return (x, i1[0]), (x2, i2[0])


def test_sqp_cstr(oracle):
    """cstr_control_test.cpp:137-183 with the Hessian update that test selects (block BFGS, :128-132). The test is SPARSE: its QP solver is
    boxADMM<VAR_SIZE, NUM_EQ, double, SPARSE, SimplicialLDLT> (:139-140, helpers.hpp:44-49) and SimplicialLDLT does NOT pivot numerically (a
    fill-reducing symmetric permutation, then plain LDL^T) — the restatement nearest to it is the static, non-pivoted order (PIVOT_STATIC), with
    glibc's exp. There the scenario runs as the reference asserts (:177): cold SOLVED in 7 SQP / 380 ADMM iterations, warm SOLVED in 4 / 403.
    (The assembly of the SPARSE members was read against the DENSE ones the restatement follows — _cost_grad_hess_sparse_update :1687-1905,
    _equalities_linearised_sparse_update :958-1022, lagrangian_gradient_hessian<SPARSE> :2178-2301: the same values, stored by column.)"""
    with oracle.libm():
        (x1, i1), (x2, i2) = _cstr_reference_scenario(oracle, oracle.PIVOT_STATIC)
    assert (i1.iter, i1.qp_solver_iter, i1.status) == (7, 380, oracle.SQP_SOLVED)
    assert (i2.iter, i2.qp_solver_iter, i2.status) == (4, 403, oracle.SQP_SOLVED) and np.isfinite(x2).all() and i2.max_violation <= 1e-3


def test_sqp_cstr_warm_solve_is_a_last_bit_property(oracle):
    """OPEN DISCREPANCY, pinned as what it is (next to HS071's `iter < 50`). The warm solve of cstr_control_test.cpp starts from multipliers for which the
    exact Lagrangian Hessian is INDEFINITE (five node blocks have an eigenvalue of -6 ... -45; the Hessian reduced to the null space of the linearised
    dynamics is positive definite, smallest eigenvalue 1e-6 from R = 5e-7) and the reference applies no regularisation (hessian_regularisation_*_impl is
    the no-op default, sqp_base.hpp:304-305). boxADMM on that non-convex QP runs to its cap with multipliers of 1e14; the line search then takes
    alpha = 2^-17, and whether the block-BFGS iteration recovers is decided by rounding: over {Eigen-pivoted, static, swept} orders x {glibc, IEEE-only
    exp} x a 1e-13 perturbation of x0 the restatement ends SOLVED (4 ... 15 iterations) in some members and at MAX_ITER_EXCEEDED (20 iterations, every QP
    at its cap) in most. The reference binary is one member (SimplicialLDLT's AMD order is not available here); `EXPECT_TRUE(SOLVED)` (:177) holds in the
    member nearest to it (test_sqp_cstr) and is not a property of the algorithm. What IS robust, in every member: the cold solve (7 / 380, SOLVED), and
    the warm solve once the Gershgorin shift of dense_sparse_compare.cpp:109-122 makes the QP convex (4 / 240, SOLVED, same optimum to 1e-7)."""
    outcomes = {}
    for glibc in (True, False):
        prev = oracle.set_libm(glibc)
        try:
            for pivot in (oracle.PIVOT_EIGEN, oracle.PIVOT_STATIC, oracle.PIVOT_SWEEP2, oracle.PIVOT_SCHUR):
                for dx in (0.0, 1e-13, -1e-13):
                    (x1, i1), (x2, i2) = _cstr_reference_scenario(oracle, pivot, perturb=dx)
                    assert (i1.iter, i1.qp_solver_iter, i1.status) == (7, 380, oracle.SQP_SOLVED), (glibc, pivot, dx)
                    outcomes[(glibc, pivot, dx)] = (i2.iter, i2.status)
                (x1, i1), (x2, i2) = _cstr_reference_scenario(oracle, pivot, regularisation=2)
                assert (i2.iter, i2.qp_solver_iter, i2.status) == (4, 240, oracle.SQP_SOLVED) and np.isfinite(x2).all(), (glibc, pivot)
        finally:
            oracle.set_libm(prev)
    solved = [k for k, v in outcomes.items() if v[1] == oracle.SQP_SOLVED]
    capped = [k for k, v in outcomes.items() if v[1] != oracle.SQP_SOLVED]
    assert solved and capped, outcomes                                   # the ensemble straddles the reference's assertion
    assert (True, oracle.PIVOT_STATIC, 0.0) in solved                    # the non-pivoted order with glibc (nearest to SimplicialLDLT) meets it
    assert all(outcomes[k][0] == 20 for k in capped)


def test_sqp_cstr_dense_bfgs_variant(oracle, transcendental_functions):
    """The same scenario with the DENSE default update (bfgs.hpp; NOT what cstr_control_test.cpp runs — kept as the round-2/3 record): the warm solve is
    equally ill-posed without regularisation (steps of 1e4..1e30; with glibc's exp and the Eigen-style pivoted order it returns to the optimum in 6
    steps, other members overflow to NaN, after which the reference's termination test — norms that drop NaNs — reports SOLVED), and robust with the
    Gershgorin shift: 4 iterations, 240 ADMM iterations, the same optimum, in every order and with either function set."""
    ref = None
    for pivot in (oracle.PIVOT_EIGEN, oracle.PIVOT_STATIC, oracle.PIVOT_SWEEP2):
        (x1, i1), (x2, i2) = _cstr_reference_scenario(oracle, pivot, regularisation=2, hessian_update=0)
        assert (i1.iter, i1.status) == (7, oracle.SQP_SOLVED)
        assert (i2.iter, i2.qp_solver_iter, i2.status) == (4, 240, oracle.SQP_SOLVED) and np.isfinite(x2).all()
        ref = x2 if ref is None else ref
        assert (np.abs(x2 - ref) / np.maximum(1.0, np.abs(ref))).max() <= 1e-7
        (x1, i1), (x2u, i2u) = _cstr_reference_scenario(oracle, pivot, regularisation=0, hessian_update=0)
        assert (i1.iter, i1.qp_solver_iter, i1.status) == (7, 401, oracle.SQP_SOLVED)
        assert i2u.status == oracle.SQP_SOLVED
        if pivot == oracle.PIVOT_EIGEN and transcendental_functions == "glibc":
            assert i2u.iter == 6 and np.isfinite(x2u).all() and abs(i2u.cost - 11662.3) < 1.0


def test_static_and_eigen_pivot_agree_on_config_A(oracle):
    """The two GPU-order restatements (static LDL^T, swept inverse) and Eigen's pivoted order give the same SQP trajectory
    to rounding on the benchmark configuration (H positive definite throughout)."""
    ss = oracle.sqp_default_settings(); ss.max_iter = 10; ss.line_search_max_iter = 10
    rng = np.random.default_rng(0)
    B = 8
    lb, ub = [], []
    for b in range(B):
        l, u = _robot_bounds(7, 0.5 + 0.4 * rng.uniform(-1, 1, 3)); lb.append(l[0]); ub.append(u[0])
    d = np.full((B, 1), 2.0)
    xe, le, ie = oracle.sqp_solve_batch(oracle.MODEL_ROBOT, 6, 1, 0.0, 2.0, B, d, np.array(lb), np.array(ub), sqp_settings=ss, pivot=0)
    xs, ls, is_ = oracle.sqp_solve_batch(oracle.MODEL_ROBOT, 6, 1, 0.0, 2.0, B, d, np.array(lb), np.array(ub), sqp_settings=ss, pivot=1)
    xw, lw, iw = oracle.sqp_solve_batch(oracle.MODEL_ROBOT, 6, 1, 0.0, 2.0, B, d, np.array(lb), np.array(ub), sqp_settings=ss, pivot=2)
    for b in range(B):
        assert ie[b].iter == is_[b].iter == iw[b].iter and ie[b].status == is_[b].status == iw[b].status
        assert ie[b].qp_solver_iter == is_[b].qp_solver_iter == iw[b].qp_solver_iter
    assert np.abs(xe - xs).max() < 1e-8 and np.abs(xe - xw).max() < 1e-8
    assert np.abs(le - ls).max() < 1e-8 and np.abs(le - lw).max() < 1e-8   # (measured 2e-10 / 3e-10)


def _cross_order(oracle, cfg, B=None, full=False, kernel_glibc=False):
    """Config `cfg` (oracle/cross_order.py) solved in the order of the kernel that serves it, with the IEEE-only functions the kernels share, and as
    the reference computes (Eigen-style pivoted LDL^T, glibc): the record of polympc_amd/parity_stats.py over EVERY instance, no mask."""
    from oracle import cross_order as tco
    wl, _ = tco.config_workload(cfg, B=B, full=full)
    n = wl["lbx"].shape[0]
    xk, lk, ik = tco.oracle_run(oracle, wl, n, tco.kernel_order(oracle, cfg, wl), kernel_glibc, 8)
    xr, lr, ir = tco.oracle_run(oracle, wl, n, oracle.PIVOT_EIGEN, True, 8)
    return tco.cross_order_stats(cfg, wl, xk, lk, ik, xr, lr, ir)


@pytest.mark.parametrize("cfg,min_qps", [("A", 1024), ("D", 512), ("B", 512), ("R", 512), ("C", 48)])
def test_kernel_orders_against_the_reference_order_one_qp_at_a_time(oracle, cfg, min_qps):
    """north_star's criterion on its own unit (one box-ADMM solve): the QPs of the reference-order SQP trajectories of every BASELINE
    configuration, each solved ONCE in the order of the kernel that serves its size at the QP entry point and once as the reference computes
    (PIVOT_EIGEN). Every QP: identical ADMM iteration counts / status / rho updates, residuals within 1e-8 (measured: <= 4.3e-10). The GPU test of
    the same name (tests/test_gpu_parity.py) puts the kernels themselves through it at larger counts."""
    from oracle import cross_order as tco
    q = tco.traced_qp_stream(oracle, cfg, min_qps)
    rows = q["n"] + q["m"]
    order = oracle.PIVOT_SWEEP if rows <= 64 else (oracle.PIVOT_SWEEP2 if rows <= oracle.SWEEP2_MAX_ROWS else oracle.PIVOT_BLOCKED)
    x, y, i = oracle.qp_solve_batch(q["H"], q["h"], q["A"], q["Alb"], q["Aub"], q["xlb"], q["xub"], settings=oracle.sqp_qp_default_settings(),
                                    pivot=order, threads=8)
    xr, yr, ir = tco.reference_qp_solve(oracle, q, threads=8)
    rec = tco.qp_level_stats(x, y, i, xr, yr, ir)
    assert rec["different_iter"] == 0 and rec["different_status"] == 0 and rec["different_rho_updates"] == 0, rec
    assert rec["max_abs_d_res_prim"] <= 1e-8 and rec["max_abs_d_res_dual"] <= 1e-8, rec


@pytest.mark.parametrize("cfg,min_qps,B", [("A", 1024, 256), ("B", 768, 256), ("R", 512, 128)])
def test_block_structured_order_against_the_reference_order(oracle, cfg, min_qps, B):
    """PIVOT_SCHUR — the order of the block-structured kernel (pmpc_qp_schur.hpp): per-node inverses of H + sigma I + rho_box, the m x m Schur
    complement swept like PIVOT_SWEEP, one step of iterative refinement on the constraint rows — admitted like the other kernel orders: on the QPs of the
    reference-order SQP with the block BFGS the reference's control tests select (config A's, config B's and the reference's 16-node grid) every QP keeps
    its ADMM iteration count, status and rho updates, the residuals lie within 1e-8 of PIVOT_EIGEN's (measured: <= 1.4e-11 — CLOSER than the swept
    orders), and whole SQP trajectories are identical with |dx| <= 1e-8."""
    from oracle import cross_order as tco
    q = tco.traced_qp_stream(oracle, cfg, min_qps, hessian_update=1)
    x, y, i = oracle.qp_solve_batch(q["H"], q["h"], q["A"], q["Alb"], q["Aub"], q["xlb"], q["xub"], settings=oracle.sqp_qp_default_settings(),
                                    pivot=oracle.PIVOT_SCHUR, threads=8, structure=q["structure"])
    xr, yr, ir = tco.reference_qp_solve(oracle, q, threads=8)
    rec = tco.qp_level_stats(x, y, i, xr, yr, ir)
    assert rec["different_iter"] == 0 and rec["different_status"] == 0 and rec["different_rho_updates"] == 0, rec
    assert rec["max_abs_d_res_prim"] <= 1e-10 and rec["max_abs_d_res_dual"] <= 1e-10, rec
    wl, _ = tco.config_workload(cfg, B=B)
    xs, ls, is_ = tco.oracle_run(oracle, wl, B, oracle.PIVOT_SCHUR, False, 8, hessian_update=1)
    xe, le, ie = tco.oracle_run(oracle, wl, B, oracle.PIVOT_EIGEN, True, 8, hessian_update=1)
    tr = tco.cross_order_stats(cfg, wl, xs, ls, is_, xe, le, ie)
    assert tr["different_trajectories"] == 0 and tr["max_abs_dx"] <= 1e-8 and tr["max_abs_d_constraint_violation"] <= 1e-10, tr


@pytest.mark.parametrize("cfg,min_qps,B", [("B", 768, 256), ("R", 512, 128)])
def test_condensed_register_order_against_the_reference_order(oracle, cfg, min_qps, B):
    """PIVOT_CONDSWEEP — the order of the condensed register kernel (pmpc_qp_cond.hpp, round 4: config B's and the 16-node grid's default route): only
    S = H + sigma I + rho_box + A' diag(rho) A is swept (the pivots of the constraint-first sweep), t = r1 + A'(rho o r2), x = S^-1 t, nu = rho o (A x - r2)
    with fma chains over the structural entries. Admitted like the other kernel orders: on the QPs of the reference-order SQP (dense damped BFGS, the
    default) every QP keeps its ADMM iteration count, status and rho updates and its residuals lie within 1e-9 of PIVOT_EIGEN's (measured: 2.9e-10 on B —
    the two-rows-per-lane full inverse: 1.1e-9 —, 5e-13 on the 16-node grid); whole SQP trajectories are identical with |dx| <= 1e-8 where the full
    inverse's are."""
    from oracle import cross_order as tco
    q = tco.traced_qp_stream(oracle, cfg, min_qps)
    x, y, i = oracle.qp_solve_batch(q["H"], q["h"], q["A"], q["Alb"], q["Aub"], q["xlb"], q["xub"], settings=oracle.sqp_qp_default_settings(),
                                    pivot=oracle.PIVOT_CONDSWEEP, threads=8, structure=q["structure"])
    xr, yr, ir = tco.reference_qp_solve(oracle, q, threads=8)
    rec = tco.qp_level_stats(x, y, i, xr, yr, ir)
    assert rec["different_iter"] == 0 and rec["different_status"] == 0 and rec["different_rho_updates"] == 0, rec
    assert rec["max_abs_d_res_prim"] <= 1e-9 and rec["max_abs_d_res_dual"] <= 1e-9, rec
    with pytest.raises(ValueError):   # the sparse products walk the nodes: the structure is part of the order
        oracle.qp_solve_batch(q["H"][:1], q["h"][:1], q["A"][:1], q["Alb"][:1], q["Aub"][:1], q["xlb"][:1], q["xub"][:1], pivot=oracle.PIVOT_CONDSWEEP)
    if cfg == "R":
        wl, _ = tco.config_workload(cfg, B=B)
        xs, ls, is_ = tco.oracle_run(oracle, wl, B, oracle.PIVOT_CONDSWEEP, False, 8)
        xe, le, ie = tco.oracle_run(oracle, wl, B, oracle.PIVOT_EIGEN, True, 8)
        tr = tco.cross_order_stats(cfg, wl, xs, ls, is_, xe, le, ie)
        assert tr["different_trajectories"] == 0 and tr["max_abs_dx"] <= 1e-8 and tr["max_abs_d_constraint_violation"] <= 1e-10, tr


def test_condensed_register_order_solves_a_traced_kkt_system(oracle):
    """One KKT solve of PIVOT_CONDSWEEP against numpy (refined in extended precision) on config-B systems, before and after a rho update, with a RANDOM
    right-hand side — the worst case for the condensed form: t = r1 + A'(rho o r2) carries the rounding of a product of size rho_eq |A| |r2| into a solve
    whose softest directions are 1 / (sigma + rho_box + lambda_min(H)): a loss of about rho_eq / rho_box x |A| = 1e3 x |A| over the quasi-definite KKT
    form, INDEPENDENT of rho (rho_box scales with it). Measured: <= 1.8e-9 relative at rho = 3.6 (pivoted LDL^T: 4e-12), 3e-13 at the initial rho; on
    the right-hand sides the ADMM actually produces (r2 = z - y / rho) the QP-level test above measures 2.9e-10 on the residuals — better than the
    full two-rows-per-lane inverse. settings.kkt_form = 1 keeps the KKT form for problems that need it."""
    from oracle import cross_order as tco
    q = tco.traced_qp_stream(oracle, "B", 420)
    n, m = q["n"], q["m"]
    rng = np.random.default_rng(7)
    worst = {0.1: 0.0, 3.6: 0.0}
    for w in (5, 100, 300, 419):
        H = q["H"][w].reshape(n, n).T; A = q["A"][w].reshape(n, m).T
        for rho in (0.1, 3.6):
            rho_vec = np.full(m, 1e3 * rho)   # equality rows (box_admm.hpp:357-396)
            K = np.block([[H + (1e-6 + rho) * np.eye(n), A.T], [A, -np.diag(1.0 / rho_vec)]])
            rhs = rng.normal(size=n + m)
            Kl = K.astype(np.longdouble); x = np.linalg.solve(K, rhs).astype(np.longdouble)
            for _ in range(3):
                x = x + np.linalg.solve(K, (rhs.astype(np.longdouble) - Kl @ x).astype(np.float64))
            x = x.astype(np.float64)
            xc = oracle.kkt_solve(np.tril(K), rho_vec, rhs, oracle.PIVOT_CONDSWEEP, structure=q["structure"])
            worst[rho] = max(worst[rho], np.abs(xc[:n] - x[:n]).max() / np.abs(x[:n]).max())
    assert worst[0.1] <= 1e-11 and worst[3.6] <= 1e-8, worst


def test_condensed_register_order_under_a_large_penalty(oracle):
    """Unbounded variables carry rho_box = RHO_MIN = 1e-6 (qp_base.hpp:195-222; the states of every BASELINE workload), and the condensed form's loss
    grows with rho_eq / (sigma + rho_box + lambda_min(H)): on a single solve with rho = 1e3 it is 1e-2 relative where the quasi-definite KKT form keeps
    1e-11. What that does to a whole box-ADMM solve — config B's QPs started from rho = 10 and rho = 1e3 instead of 0.1 (the benchmark
    streams themselves stay below rho = 70): every QP keeps its ADMM iteration count and status against the reference order, and the solutions are CLOSER
    to it than those of the full two-rows-per-lane inverse this order replaced (measured: rho0 = 10: 7.5e-9 against 3.5e-7; 1e3: 8.9e-7 against 1.9e-5;
    1e5: 3.1e-4 against 1.25). The acceptance test of the ADMM evaluates the true residuals (H x, A x, A' y from the data), so an inexact linear solve
    can cost iterations, never a wrong SOLVED."""
    from oracle import cross_order as tco
    q = tco.traced_qp_stream(oracle, "B", 150)
    for rho0, bound in ((10.0, 5e-8), (1e3, 5e-6)):
        s = oracle.sqp_qp_default_settings(); s.rho = rho0
        args = (q["H"], q["h"], q["A"], q["Alb"], q["Aub"], q["xlb"], q["xub"])
        xr, yr, ir = oracle.qp_solve_batch(*args, settings=s, pivot=oracle.PIVOT_EIGEN, threads=8)
        xc, yc, ic = oracle.qp_solve_batch(*args, settings=s, pivot=oracle.PIVOT_CONDSWEEP, threads=8, structure=q["structure"])
        xf, yf, if_ = oracle.qp_solve_batch(*args, settings=s, pivot=oracle.PIVOT_SWEEP2, threads=8)
        assert [i.iter for i in ic] == [i.iter for i in ir] and [i.status for i in ic] == [i.status for i in ir], rho0
        dc, df = np.abs(xc - xr).max(), np.abs(xf - xr).max()
        assert dc <= bound and dc <= df, (rho0, dc, df)


@pytest.mark.parametrize("cfg", ["A", "B", "R"])
def test_kernel_orders_follow_exact_arithmetic_more_closely_than_the_reference_order(oracle, cfg):
    """Which side is inexact when the kernel order and the reference order disagree at a large penalty? PIVOT_EXACT solves every linear system of the ADMM
    to working accuracy (the Eigen-order factor as a preconditioner, residuals in long double): the trajectory of exact arithmetic. On the QP streams of
    configs A, B and the reference's 16-node grid, penalty started at 10 / 1e3 / 1e5 (RHO_MAX = 1e6, RHO_EQ_FACTOR = 1e3: box_admm.hpp:56-59), the order of
    the default kernel — constraint-first sweep (A), condensed register kernel (B, R) — keeps every ADMM iteration count of the exact run and lies CLOSER
    to it than the reference's pivoted LDL^T of the quasi-definite matrix does (measured max |dx|, kernel order / Eigen order: A 1e-10 / 5e-10, 2e-9 /
    1e-7, 4e-6 / 5e-4; B 2e-9 / 5e-9, 5e-8 / 5e-7, 6e-5 / 2e-4; R 7e-11 / 1e-8, 2e-9 / 2e-5, 3e-5 / 2e-2): the directions A leaves free are the bounded
    controls, whose rho_box scales with rho, so cond(S) stays ~1e5 whatever rho is — while diagonal pivoting on a matrix with entries 1e-9 .. 1e1 is not
    backward stable. The disagreement with the reference order at large rho that round 4 recorded is therefore the REFERENCE order's rounding; matching it
    to 1e-8 there would mean reproducing its error. The conditioning gate stays silent on all of these."""
    from oracle import cross_order as tco
    kern = oracle.PIVOT_SWEEP if cfg == "A" else oracle.PIVOT_CONDSWEEP
    q = tco.traced_qp_stream(oracle, cfg, 32)
    args = (q["H"], q["h"], q["A"], q["Alb"], q["Aub"], q["xlb"], q["xub"])
    for rho0 in (10.0, 1e3, 1e5):
        s = oracle.sqp_qp_default_settings(); s.rho = rho0
        xe, ye, ie = oracle.qp_solve_batch(*args, settings=s, pivot=oracle.PIVOT_EXACT, threads=8)
        xr, yr, ir = oracle.qp_solve_batch(*args, settings=s, pivot=oracle.PIVOT_EIGEN, threads=8)
        xk, yk, ik = oracle.qp_solve_batch(*args, settings=s, pivot=kern, threads=8, structure=q["structure"])
        assert [i.iter for i in ik] == [i.iter for i in ie] and [i.status for i in ik] == [i.status for i in ie], rho0
        assert all(i.flags == 0 for i in ik), rho0
        dk, de = np.abs(xk - xe).max(), np.abs(xr - xe).max()
        print(cfg, rho0, "kernel order vs exact", dk, "reference order vs exact", de)
        assert dk <= max(2 * de, 1e-9), (cfg, rho0, dk, de)         # never noticeably farther from exact arithmetic than the reference order ...
        if rho0 >= 1e3: assert dk <= 0.5 * de, (cfg, rho0, dk, de)   # ... and at a large penalty several times to four orders of magnitude closer
        assert dk <= 1e-8 * max(1.0, rho0 / 10.0), (cfg, rho0, dk)   # (what is left grows with rho for ANY fp64 solve: the exact run's own conditioning)


@pytest.mark.parametrize("cfg,kern", [("A", "PIVOT_SWEEP"), ("B", "PIVOT_CONDENSED")])
def test_conditioning_gate_of_the_condensed_orders(oracle, cfg, kern):
    """When the condensed form IS the inexact one, and what the numeric gate does about it (the QP entry point's one-row-per-lane kernels: PIVOT_SWEEP; the
    large-instance kernel: PIVOT_CONDENSED — here on config B's QPs, the policy restates any size). With every bound removed the directions A leaves free
    carry rho_box = RHO_MIN = 1e-6 instead of rho, cond(S) = rho_eq |A|^2 / lambda_min grows with rho, and a condensed solve loses cond(S) eps (single solves
    at rho = 1e3: x 1e-5, nu 1 relative; the quasi-definite form 1e-10). The estimate — max S_ii * max |(S^-1)_ii| (swept inverse) or max S_ii / min |d_k|
    (LDL^T); within a factor of two of each other and 2 .. 10 below cond(S) — trips the gate at 1e10: the QP is given up (UNSOLVED + the flag); the QP entry point
    solves it again on the LDS-resident static LDL^T (PIVOT_SWEEP -> PIVOT_STATIC, as the product's redo launch does), the SQP driver the whole instance.
    Silent on the benchmark streams at every rho (cond(S) ~ 1e5: the bounded controls), set on every QP of the unbounded streams at rho0 = 1e5."""
    from oracle import cross_order as tco
    kern = getattr(oracle, kern)
    q = tco.traced_qp_stream(oracle, cfg, 32)
    free_l, free_u = np.full_like(q["xlb"], -np.inf), np.full_like(q["xub"], np.inf)
    for rho0 in (0.1, 1e3, 1e5):
        s = oracle.sqp_qp_default_settings(); s.rho = rho0
        xb, yb, ib = oracle.qp_solve_batch(q["H"], q["h"], q["A"], q["Alb"], q["Aub"], q["xlb"], q["xub"], settings=s, pivot=kern, threads=8)
        assert all(i.flags == 0 for i in ib), (cfg, rho0)
        args = (q["H"], q["h"], q["A"], q["Alb"], q["Aub"], free_l, free_u)
        xk, yk, ik = oracle.qp_solve_batch(*args, settings=s, pivot=kern, threads=8)
        flagged = np.array([i.flags & oracle.FLAG_ILLCOND for i in ik]) != 0
        if rho0 == 1e5:
            assert flagged.all(), (cfg, rho0)
        if cfg == "A" and flagged.any():   # re-solved in the full KKT form (static LDL^T): as close to exact arithmetic as the reference order is
            xe, ye, ie = oracle.qp_solve_batch(*args, settings=s, pivot=oracle.PIVOT_EXACT, threads=8)
            xr, yr, ir = oracle.qp_solve_batch(*args, settings=s, pivot=oracle.PIVOT_EIGEN, threads=8)
            dk, de = np.abs(xk - xe)[flagged].max(), np.abs(xr - xe)[flagged].max()
            print(cfg, rho0, int(flagged.sum()), "flagged: full KKT form vs exact", dk, "reference order vs exact", de)
            assert dk <= max(100 * de, 1e-7), (rho0, dk, de)
        if cfg != "A":   # a QP that gave up reports UNSOLVED (the SQP driver never uses it)
            assert all(i.status == oracle.QP_UNSOLVED for i, f in zip(ik, flagged) if f)


def test_sqp_conditioning_gate_redo(oracle):
    """The instance-level rule of the register-resident SQP kernels and their redo launch: the reference's 11-node robot grid with the control bounds removed
    (the pinned initial state stays) — an unbounded control means rho_box = RHO_MIN in a direction the collocation Jacobian leaves free, so the instance is
    solved in the full KKT form from the start: bit for bit the PIVOT_SWEEP2 solve, flag set, whatever the penalty; with the control bounds in place the
    flag stays clear. (Decided from the bounds once per instance: a numeric gate at every factorisation cost those kernels 4 .. 10 %.)"""
    from polympc_amd import workloads
    B = 6
    wl = workloads.robot_batch(B, P=5, S=2)
    ss = oracle.sqp_default_settings(); ss.max_iter = 4; ss.line_search_max_iter = wl["ls_max_iter"]
    qs = oracle.sqp_qp_default_settings(); qs.rho = 1e4
    run = lambda lbx, ubx, pivot: oracle.sqp_solve_batch(wl["model"], wl["P"], wl["S"], wl["t0"], wl["tf"], B, wl["d"], lbx, ubx, sqp_settings=ss, qp_settings=qs, pivot=pivot, threads=4)
    x0, l0, i0 = run(wl["lbx"], wl["ubx"], oracle.PIVOT_CONDSWEEP)
    assert all(i.flags == 0 for i in i0)
    nn = wl["P"] * wl["S"] + 1
    lbx, ubx = wl["lbx"].copy(), wl["ubx"].copy()
    lbx[:, 3 * nn:] = -np.inf; ubx[:, 3 * nn:] = np.inf
    xc, lc, ic = run(lbx, ubx, oracle.PIVOT_CONDSWEEP)
    xf, lf, if_ = run(lbx, ubx, oracle.PIVOT_SWEEP2)
    assert all(i.flags & oracle.FLAG_ILLCOND for i in ic) and all(i.flags == 0 for i in if_)
    assert np.array_equal(xc, xf) and np.array_equal(lc, lf)
    assert [(i.iter, i.status, i.qp_solver_iter) for i in ic] == [(i.iter, i.status, i.qp_solver_iter) for i in if_]


def test_block_structured_order_needs_its_refinement_step(oracle):
    """Why PIVOT_SCHUR refines: a config-B KKT system after a rho update (rho = 3.6: cond(S) = 6e5, S = 1/rho + A Q A'). The swept inverse of S has an
    isotropic forward error ~ eps cond(S) |nu|, but x = Q (r1 - A' nu) tolerates errors of nu only where Q^(1/2) A' nearly vanishes — without the step
    the kernel order was 2e-9 .. 2e-8 off in x (1.8e-6 in the reported residuals of config B's QPs, 3.6 in a trajectory); with it the solve is MORE
    accurate than the dense orders. Reference: numpy solve refined in extended precision."""
    from oracle import cross_order as tco
    q = tco.traced_qp_stream(oracle, "B", 420, hessian_update=1)
    n, m = q["n"], q["m"]
    worst = 0.0
    for w in (5, 100, 300, 419):
        H = q["H"][w].reshape(n, n).T; A = q["A"][w].reshape(n, m).T; lx, ux = q["xlb"][w], q["xub"][w]
        for rho in (0.1, 3.629):
            rb = np.where((lx < -1e10) & (ux > 1e10), 1e-6, np.where(ux - lx < 1e-4, 1e3 * rho, rho))
            Pm = np.tril(H) + np.tril(H, -1).T + np.diag(1e-6 + rb)
            rv = np.full(m, 1e3 * rho)
            K = np.block([[Pm, A.T], [A, -np.diag(1 / rv)]])
            rhs = np.random.default_rng(w).uniform(-1, 1, n + m)
            ld = np.longdouble
            sol = np.linalg.solve(K, rhs).astype(ld)
            for _ in range(6):
                sol = sol + np.linalg.solve(K, (rhs.astype(ld) - K.astype(ld) @ sol).astype(float)).astype(ld)
            ref = sol.astype(float)
            got = oracle.kkt_solve(K, rv, rhs, oracle.PIVOT_SCHUR, structure=q["structure"])
            dense = oracle.kkt_solve(K, rv, rhs, oracle.PIVOT_SWEEP2)
            sc = max(1.0, np.abs(ref).max())
            worst = max(worst, np.abs(got - ref).max() / sc)
            assert np.abs(got - ref).max() <= 1e-12 * sc and np.abs(got[:n] - ref[:n]).max() <= np.abs(dense[:n] - ref[:n]).max() + 1e-13 * sc, (w, rho)
    assert worst > 0.0


def test_kernel_orders_against_the_reference_order_on_the_full_benchmark_streams(oracle, transcendental_functions):
    """The cross-order table of DESIGN.md §5, asserted: every BASELINE configuration at its FULL size (C: 64 of its 1024 instances — 57 QP/s on this
    container) in the kernel's order and function set against the reference's (PIVOT_EIGEN + glibc). What holds on every instance: the same SQP
    iterations, status and total ADMM iterations (config B: on all but 2 of 16 384 — see below), constraint violation within 1e-8 and cost within 1e-8
    relative. What north_star's 1e-8 does NOT bound is the worst instance of the SOLUTION difference: the measured tails are asserted as such
    (A 1.7e-8, D 7.4e-8 absolute; B 5.8e-6 of a variable's magnitude), next to the percentiles that are far inside it."""
    if transcendental_functions != "glibc":
        pytest.skip("this test chooses the function set of each run itself: one pass")
    r = _cross_order(oracle, "A", full=True)
    assert r["instances"] == 4096 and r["different_trajectories"] == 0
    assert r["max_abs_dx"] <= 2e-8 and r["abs_dx_per_instance"]["p99"] <= 1e-10 and r["abs_dx_per_instance"]["p50"] <= 1e-12   # measured 1.72e-8 / 2.4e-11 / 2.0e-13
    assert r["scaled_dlam_per_instance"]["max"] <= 5e-8 and r["scaled_dlam_per_instance"]["p99"] <= 1e-10                       # 3.2e-8 / 4.8e-11
    assert r["max_abs_d_constraint_violation"] <= 1e-10 and r["max_rel_d_cost"] <= 1e-10                                        # 2.4e-11 / 1.9e-11
    r = _cross_order(oracle, "D", full=True)
    assert r["instances"] == 8192 and r["different_trajectories"] == 0
    assert r["max_abs_dx"] <= 1e-7 and r["abs_dx_per_instance"]["p99"] <= 1e-10            # measured 7.4e-8 (2 instances above 1e-8) / 2.0e-11
    assert r["instances_dx_over_1e-8"] <= 4
    assert r["max_abs_d_constraint_violation"] <= 1e-9 and r["max_rel_d_cost"] <= 1e-9     # 1.3e-10 / 5.1e-11
    r = _cross_order(oracle, "R", full=True)   # the reference's own 16-node grid (mpc_wrapper_test.cpp:24-25), 2048 instances
    assert r["different_trajectories"] == 0 and r["max_abs_dx"] <= 1e-8 and r["max_abs_d_constraint_violation"] <= 1e-10   # measured 1.5e-9 / 2.1e-12
    r = _cross_order(oracle, "C")
    assert r["instances"] == 64 and r["different_trajectories"] == 0 and r["max_abs_dx"] <= 1e-10 and r["max_rel_d_cost"] <= 1e-10   # 8.4e-12 / 8.1e-13


def test_config_B_cross_order_spread_is_the_conditioning_of_the_trajectory(oracle, transcendental_functions):
    """Config B at its full 16 384 instances. The CSTR iteration is much less contractive than the robot's (59 ADMM iterations per QP stopped at 1e-4,
    up to 20 SQP iterations, Arrhenius terms exp(E / (273.15 + T))): last-bit differences grow along the trajectory. Measured, and asserted here:
      * kernel order (since round 4 the condensed register kernel, PIVOT_CONDSWEEP) + IEEE-only exp vs reference order + glibc: 2 of 16 384 instances
        take another branch (the two runs of such an instance end up to 1.6 apart; one of the two keeps its iteration counts); on the other 16 382
        the scaled solution difference is p50 4e-13, p90 7e-12, p99 7.5e-10 (round 3's full inverse: 7e-13 / 1.2e-11 / 1.6e-9, worst 5.8e-6);
      * the SAME order (PIVOT_EIGEN) with glibc's exp against the IEEE-only exp — two correctly-rounded-to-1-ulp implementations of one function, no
        linear algebra involved — already flips one trajectory and moves the others by up to 1.4e-5 scaled.
    So the tail is the conditioning of a 20-iteration SQP trajectory with respect to ANY last-bit change, not an accuracy deficit of an elimination
    order (the orders solve a traced KKT system equally well: test_*_policy_matches_factorisations). The comparison that isolates the kernels —
    same order, same functions — is exact, and tested bit for bit on the GPU."""
    if transcendental_functions != "glibc":
        pytest.skip("this test chooses the function set of each run itself: one pass")
    r = _cross_order(oracle, "B", full=True)
    i = r["unbranched_only"]   # (identical counts AND the two runs within 1e-4: one instance keeps its counts on another branch and ends 1.55 apart)
    assert r["instances"] == 16384 and r["different_trajectories"] + i["same_counts_but_over_1e-4"] <= 4   # measured 1 + 1 (round 3's full inverse: 2 + 0)
    p = r["scaled_dx_per_instance"]
    assert p["p50"] <= 1e-11 and p["p90"] <= 1e-9 and p["p99"] <= 1e-8                                     # 4.4e-13 / 7.0e-12 / 7.5e-10 (round 3: 6.7e-13 / 1.2e-11 / 1.6e-9)
    assert i["max_scaled_dx"] <= 1e-4 and i["max_abs_d_constraint_violation"] <= 1e-5 and i["max_rel_d_cost"] <= 1e-5
    # the function set alone, in the reference's own order
    from oracle import cross_order as tco
    wl, _ = tco.config_workload("B", full=True)
    xa, la, ia = tco.oracle_run(oracle, wl, 16384, oracle.PIVOT_EIGEN, False, 8)
    xb, lb, ib = tco.oracle_run(oracle, wl, 16384, oracle.PIVOT_EIGEN, True, 8)
    q = tco.cross_order_stats("B", wl, xa, la, ia, xb, lb, ib)
    assert 1 <= q["different_trajectories"] <= 4                                                          # measured 1: exp's last bit flips a branch
    assert q["identical_trajectories_only"]["max_scaled_dx"] >= 1e-7                                      # measured 1.4e-5: as large as the cross-order tail
    assert q["scaled_dx_per_instance"]["p99"] <= 1e-8


# ---------------------------------------------------------------- PIVOT_SWEEP2: the two-rows-per-lane register kernel's order (65..112 rows)
@pytest.mark.parametrize("n,m", [(66, 44), (55, 33), (41, 24), (70, 42), (80, 48)])
def test_sweep2_policy_matches_factorisations(oracle, n, m):
    """The blocked sweep carried to 5..7 tiles per dimension with the mat-vec order of pmpc_qp_reg2.hpp (four chains per row, columns
    j = q mod 4) against numpy and against both LDL^T policies, on quasi-definite KKT matrices with the conditioning of the ADMM; sizes:
    config B (110 rows), the 11-node robot grid (88), the smallest size the path takes (65), the limit of the all-register variant (112) and the
    limit of the path (128: the reference's 16-node robot grid)."""
    rng = np.random.default_rng(n * 100 + m)
    G = rng.normal(size=(n, n)); H = G @ G.T / n + (1e-6 + 0.1) * np.eye(n); A = rng.normal(size=(m, n))
    K = np.block([[H, A.T], [A, -np.diag(1.0 / rng.choice([0.1, 100.0], m))]])
    b = rng.normal(size=n + m)
    x_ref = np.linalg.solve(K, b)
    scale = np.abs(x_ref).max()
    xs = oracle.ldlt_solve(np.tril(K), b, oracle.PIVOT_SWEEP2)
    assert np.abs(xs - x_ref).max() < 1e-9 * scale
    for piv in (oracle.PIVOT_EIGEN, oracle.PIVOT_STATIC):
        assert np.abs(xs - oracle.ldlt_solve(np.tril(K), b, piv)).max() < 1e-9 * scale


def test_sweep2_policy_rejects_systems_over_128_rows(oracle):
    from polympc_amd import workloads
    wl = workloads.robot_batch(1, P=4, S=4)   # 17 nodes: 136 rows
    with pytest.raises(ValueError):
        oracle.sqp_solve_batch(oracle.MODEL_ROBOT, 4, 4, 0.0, 2.0, 1, wl["d"], wl["lbx"], wl["ubx"], pivot=oracle.PIVOT_SWEEP2)


def test_sweep2_policy_on_the_11_node_robot_grid(oracle):
    """The reference's 11-node robot grid (88 KKT rows, P = 5, S = 2) at 1024 instances: the swept-inverse order of the two-rows-per-lane kernel follows
    the Eigen-pivoted trajectory — identical SQP and ADMM iteration counts and statuses on every instance, solutions within 1e-8. (Config B's
    own stream: test_config_B_cross_order_spread_is_the_conditioning_of_the_trajectory, all 16 384 instances.)"""
    from polympc_amd import workloads
    B = 1024
    wl = workloads.robot_batch(B, P=5, S=2)
    ss = oracle.sqp_default_settings(); ss.max_iter = wl["max_iter"]; ss.line_search_max_iter = wl["ls_max_iter"]
    r = {}
    for piv in (oracle.PIVOT_EIGEN, oracle.PIVOT_SWEEP2):
        x, l, info = oracle.sqp_solve_batch(wl["model"], wl["P"], wl["S"], wl["t0"], wl["tf"], B, wl["d"], wl["lbx"], wl["ubx"],
                                            sqp_settings=ss, pivot=piv, threads=8)
        r[piv] = (x, [i.iter for i in info], [i.qp_solver_iter for i in info], [i.status for i in info])
    a, b = r[oracle.PIVOT_EIGEN], r[oracle.PIVOT_SWEEP2]
    assert a[1] == b[1] and a[2] == b[2] and a[3] == b[3]
    assert np.abs(a[0] - b[0]).max() <= 1e-8


# ---------------------------------------------------------------- PIVOT_BLOCKED: the blocked tile LDL^T of the large-instance kernel
@pytest.mark.parametrize("n,m", [(256, 208), (100, 60), (40, 25), (130, 70)])
def test_blocked_policy_matches_factorisations(oracle, n, m):
    """PIVOT_STATIC's factor and forward pass with the backward pass by column dot products in blocks of 16 (64 partial sums per column): against
    numpy and both LDL^T policies; sizes: config C (464 rows), ragged sizes around the 64-row slots, a partial last block."""
    rng = np.random.default_rng(n * 100 + m)
    G = rng.normal(size=(n, n)); H = G @ G.T / n + (1e-6 + 0.1) * np.eye(n); A = rng.normal(size=(m, n))
    K = np.block([[H, A.T], [A, -np.diag(1.0 / rng.choice([0.1, 100.0], m))]])
    b = rng.normal(size=n + m)
    x_ref = np.linalg.solve(K, b)
    scale = np.abs(x_ref).max()
    xs = oracle.ldlt_solve(np.tril(K), b, oracle.PIVOT_BLOCKED)
    assert np.abs(xs - x_ref).max() < 1e-9 * scale
    for piv in (oracle.PIVOT_EIGEN, oracle.PIVOT_STATIC):
        assert np.abs(xs - oracle.ldlt_solve(np.tril(K), b, piv)).max() < 1e-9 * scale


def test_blocked_policy_on_config_C_stream(oracle):
    """The kite stand-in (464 KKT rows): the blocked order follows the Eigen-pivoted trajectory — identical SQP / ADMM iteration counts and statuses,
    solutions within 1e-9."""
    from polympc_amd import workloads
    B = 8
    wl = workloads.kite_standin_batch(B)
    ss = oracle.sqp_default_settings(); ss.max_iter = wl["max_iter"]; ss.line_search_max_iter = wl["ls_max_iter"]
    r = {}
    for piv in (oracle.PIVOT_EIGEN, oracle.PIVOT_BLOCKED):
        x, l, info = oracle.sqp_solve_batch(wl["model"], wl["P"], wl["S"], wl["t0"], wl["tf"], B, wl["d"], wl["lbx"], wl["ubx"], sqp_settings=ss, pivot=piv, threads=8)
        r[piv] = (x, [i.iter for i in info], [i.qp_solver_iter for i in info], [i.status for i in info])
    a, b = r[oracle.PIVOT_EIGEN], r[oracle.PIVOT_BLOCKED]
    assert a[1:] == b[1:] and np.abs(a[0] - b[0]).max() <= 1e-9


def test_sqp_iteration_records(oracle):
    """sqp_settings_t::iteration_callback (sqp_base.hpp:33, :685-686) restated as records: one per SQP iteration, in order, the last one equal to
    what the solver reports; the first iteration takes the full step of its (feasible-by-construction) QP unless the line search cuts it."""
    from polympc_amd import workloads
    B, cap = 8, 12
    wl = workloads.robot_batch(B)
    ss = oracle.sqp_default_settings(); ss.max_iter = 10; ss.line_search_max_iter = 10
    tr = np.zeros((B, cap, oracle.TRACE_DOUBLES)); oracle.bind_iteration_trace(ss, tr)
    x, lam, info = oracle.sqp_solve_batch(wl["model"], wl["P"], wl["S"], wl["t0"], wl["tf"], B, wl["d"], wl["lbx"], wl["ubx"], sqp_settings=ss)
    for b in range(B):
        k = info[b].iter
        assert np.array_equal(tr[b, :k, 0], np.arange(1, k + 1)) and np.all(tr[b, k:] == 0)
        last = tr[b, k - 1]
        assert (last[2], last[3], last[4], last[7]) == (info[b].primal_norm, info[b].dual_norm, info[b].cost, info[b].max_violation)
        assert tr[b, :k, 5].sum() == info[b].qp_solver_iter
        assert np.all((tr[b, :k, 1] > 0) & (tr[b, :k, 1] <= 1))
    # a capacity smaller than the iteration count keeps the first records only
    tr2 = np.zeros((B, 3, oracle.TRACE_DOUBLES)); oracle.bind_iteration_trace(ss, tr2)
    oracle.sqp_solve_batch(wl["model"], wl["P"], wl["S"], wl["t0"], wl["tf"], B, wl["d"], wl["lbx"], wl["ubx"], sqp_settings=ss)
    assert np.array_equal(tr2, tr[:, :3])


def test_box_admm_single_precision_float_fixture(oracle):
    """tests/solvers/qp/box_admm_test.cpp:85-115 (box_admmSinglePrecisionFloat): boxADMM<2, 1, float>, max_iter = 150 — the solution within 1e-2 of
    (0.3, 0.7) in isApprox's sense, SOLVED, fewer than 150 iterations; the float restatement (oracle/qp_f32.hpp) under Eigen-style pivoting and in the
    static order of the HIP kernel, both with the iteration count of the double solve of the same problem. On random QPs the float restatement
    agrees with the double one to single-precision accuracy."""
    H = np.array([[4, 1, 1, 2]], dtype=np.float32); h = [[1, 1]]; A = [[1, 1]]
    s = oracle.qp_default_settings(); s.max_iter = 150
    xd, yd, infod = oracle.qp_solve_batch(H, h, A, [[1]], [[1]], [[0, 0]], [[0.7, 0.7]], settings=s)
    sol = np.array([0.3, 0.7], dtype=np.float32)
    for piv in (oracle.PIVOT_EIGEN, oracle.PIVOT_STATIC):
        x, y, info = oracle.qp_solve_batch_f32(H, h, A, [[1]], [[1]], [[0, 0]], [[0.7, 0.7]], settings=s, pivot=piv)
        assert x.dtype == np.float32
        assert np.linalg.norm(x[0] - sol) <= 1e-2 * min(np.linalg.norm(x[0]), np.linalg.norm(sol))   # Eigen's isApprox
        assert info[0].status == oracle.QP_SOLVED and info[0].iter < 150 and info[0].iter == infod[0].iter
        assert np.abs(x[0] - xd[0]).max() < 1e-5
    rng = np.random.default_rng(5)
    n, m, B = 7, 3, 6
    Hs, hs, As, al, au, xl, xu = [], [], [], [], [], [], []
    for _ in range(B):
        G = rng.standard_normal((n, n)); Hm = G @ G.T + n * np.eye(n); Am = rng.standard_normal((m, n)); c = rng.standard_normal(m)
        Hs.append(Hm.T.ravel()); hs.append(rng.standard_normal(n)); As.append(Am.T.ravel()); al.append(c - 1.0); au.append(c + 1.0)
        xl.append(-np.ones(n)); xu.append(np.ones(n))
    s2 = oracle.qp_default_settings(); s2.max_iter = 500
    xd, yd, infod = oracle.qp_solve_batch(Hs, hs, As, al, au, xl, xu, settings=s2)
    for piv in (oracle.PIVOT_EIGEN, oracle.PIVOT_STATIC):
        x, y, info = oracle.qp_solve_batch_f32(Hs, hs, As, al, au, xl, xu, settings=s2, pivot=piv)
        assert all(i.status == oracle.QP_SOLVED for i in info)
        assert np.abs(x - xd).max() < 2e-3   # both stop at eps = 1e-3; the iterates agree to single precision until then


def test_admm_single_precision_float_fixture(oracle):
    """tests/solvers/qp/admm_solver_test.cpp:84-113 (admmSinglePrecisionFloat): ADMM<2, 1, float> with the default settings — within 1e-2 of (0.3, 0.7),
    SOLVED, fewer than max_iter iterations, under both factorisation orders of the float restatement; the iteration count of the double solve."""
    H = np.array([[4, 1, 1, 2]], dtype=np.float32)
    s = oracle.qp_default_settings()
    xd, yd, infod = oracle.qp_admm_solve_batch(H, [[1, 1]], [[1, 1]], [[1]], [[1]], [[0, 0]], [[0.7, 0.7]], settings=s)
    sol = np.array([0.3, 0.7], dtype=np.float32)
    for piv in (oracle.PIVOT_EIGEN, oracle.PIVOT_STATIC):
        x, y, info = oracle.qp_solve_batch_f32(H, [[1, 1]], [[1, 1]], [[1]], [[1]], [[0, 0]], [[0.7, 0.7]], settings=s, pivot=piv, osqp_form=True)
        assert np.linalg.norm(x[0] - sol) <= 1e-2 * min(np.linalg.norm(x[0]), np.linalg.norm(sol))
        assert info[0].status == oracle.QP_SOLVED and info[0].iter < s.max_iter and info[0].iter == infod[0].iter
        assert np.abs(x[0] - xd[0]).max() < 1e-5


# ---------------------------------------------------------------- PIVOT_CONDENSED: the condensed form of the large-instance kernel (round 3)
@pytest.mark.parametrize("n,m", [(8, 5), (35, 21), (66, 44), (120, 96)])
def test_condensed_policy_solves_the_same_qps(oracle, n, m):
    """boxADMM with the constraint block eliminated in closed form (S = H + sigma I + rho_box + A' diag(rho) A, x = S^-1 (r1 + A'(rho o r2)),
    nu = rho o (A x - r2): pmpc_qp_big.hpp, condensed mode) against the Eigen-style pivoted LDL^T of the full KKT matrix on random strictly convex
    QPs with equality, inequality and loose rows: the same ADMM iteration counts, statuses and rho updates, solutions within 1e-9 of the scale."""
    rng = np.random.default_rng(1000 * n + m)
    B = 6
    G = rng.normal(size=(B, n, n)); H = G @ G.transpose(0, 2, 1) / n + 0.05 * np.eye(n)
    A = rng.normal(size=(B, m, n)) * (rng.random(size=(B, m, n)) < 0.3)   # sparse rows, as a collocation Jacobian has
    x0 = rng.normal(size=(B, n))
    Ax = np.einsum("bij,bj->bi", A, x0)
    Alb = Ax.copy(); Aub = Ax.copy()
    Aub[:, m // 2:] += 0.5; Alb[:, m // 2:] -= 0.5          # second half: inequality rows
    Alb[:, -1] = -1e20; Aub[:, -1] = 1e20                   # one loose row
    h = rng.normal(size=(B, n)); xlb = np.full((B, n), -2.0); xub = np.full((B, n), 2.0)
    s = oracle.qp_default_settings(); s.max_iter = 200; s.adaptive_rho = 1; s.adaptive_rho_interval = 25; s.check_termination = 10; s.eps_abs = 1e-5; s.eps_rel = 1e-5
    col = lambda M: np.ascontiguousarray(M.transpose(0, 2, 1)).reshape(B, -1)
    xr, yr, ir = oracle.qp_solve_batch(col(H), h, col(A), Alb, Aub, xlb, xub, settings=s, pivot=oracle.PIVOT_EIGEN)
    xc, yc, ic = oracle.qp_solve_batch(col(H), h, col(A), Alb, Aub, xlb, xub, settings=s, pivot=oracle.PIVOT_CONDENSED)
    assert [i.iter for i in ir] == [i.iter for i in ic] and [i.status for i in ir] == [i.status for i in ic]
    assert [i.rho_updates for i in ir] == [i.rho_updates for i in ic]
    scale = max(1.0, np.abs(xr).max(), np.abs(yr).max())
    assert np.abs(xc - xr).max() <= 1e-9 * scale and np.abs(yc - yr).max() <= 1e-8 * scale


def test_condensed_policy_follows_the_eigen_order_trajectories(oracle):
    """The condensed order on the benchmark streams — config C (the kernel that uses it), the reference's 16-node robot grid and config A: identical SQP
    iterations, statuses and total ADMM iterations as the Eigen-pivoted order on every instance, solutions within north_star's 1e-8 (measured: C 4.9e-12,
    R 3.2e-10, A 3.9e-9 — closer to the Eigen order than the sweep / blocked orders of round 2 are)."""
    from oracle import cross_order as tco
    for cfg, B, tol in (("C", 16, 1e-10), ("R", 128, 1e-8), ("A", 512, 1e-8)):
        wl, _ = tco.config_workload(cfg, B=B)
        xr, lr, ir = tco.oracle_run(oracle, wl, B, oracle.PIVOT_EIGEN, True, 8)
        xc, lc, ic = tco.oracle_run(oracle, wl, B, oracle.PIVOT_CONDENSED, False, 8)
        r = tco.cross_order_stats(cfg, wl, xc, lc, ic, xr, lr, ir)
        assert r["different_trajectories"] == 0, (cfg, r)
        assert r["max_abs_dx"] <= tol and r["max_abs_d_constraint_violation"] <= 1e-9 and r["max_rel_d_cost"] <= 1e-9, (cfg, r)


def test_null_space_form_cannot_serve_the_minimal_time_problem(oracle):
    """VERDICT r5 item 4 asked for a null-space / reduced-Hessian variant of the block-structured kernel — eliminate the STATES through the state block A_x of
    the collocation Jacobian instead of the multipliers through S = 1/rho + A Q A' — for the reference's NP = 1 problems, "or show why it cannot work". It
    cannot, on exactly the problem it was meant for (minimal_time_test.cpp:146-188): A_x = D (x) I - t_s df/dx is SQUARE (NX nn x NX nn) and SINGULAR there — the
    differentiation matrix annihilates constant profiles and the parking dynamics do not depend on the position states at all — at the reference's guess and at
    its solution alike (sigma_min ~ 1e-16, rank deficiency >= 2). What makes the problem well posed are the BOUNDS that pin the initial state (lbx = ubx on the last
    node): with those three columns removed the remaining 33 x 30 block has full column rank and a condition number below 100 — but which columns are pinned is
    known to boxADMM only through rho_box (an active-set property), not to an elimination of the equality rows. The range-space form's own difficulty on this
    problem (every instance meets its conditioning gate once rho adapts beyond ~25, EXPERIMENTS.md round 5) is the same singularity seen from the other side: S
    is regular only through 1/rho_eq and the bounded controls' Q ~ 1/rho. The dense kernels factorise the quasi-definite (n + m)-row matrix, where the pinned
    states' rho_box = 1e3 rho does the regularising, and stay the default for this problem."""
    lbx, ubx, xg = _minimal_time_parking(11)
    nn, nx = 11, 3
    ss = oracle.sqp_default_settings(); ss.max_iter = 20; ss.line_search_max_iter = 10; ss.exact_hessian_every_iter = 1; ss.regularisation = 2
    d = np.array([[1.0]])
    x, lam, info = oracle.sqp_solve_batch(oracle.MODEL_PARKING, 5, 2, 0.0, 1.0, 1, d, lbx, ubx, x_guess=xg, sqp_settings=ss, pivot=oracle.PIVOT_EIGEN)
    assert info[0].status == 0 and info[0].iter < 20
    pinned = [i for i in range(nx * nn) if lbx[0, i] == ubx[0, i]]
    assert len(pinned) == nx
    free = [i for i in range(nx * nn) if i not in pinned]
    for pt in (xg[0], x[0]):
        J = oracle.ocp_eval(oracle.MODEL_PARKING, 5, 2, 0.0, 1.0, pt, d[0], lam=np.zeros(33 + 56))["jac"]
        Ax = J[:, :nx * nn]
        sv = np.linalg.svd(Ax, compute_uv=False)
        assert Ax.shape == (33, 33) and sv[-1] < 1e-12 * sv[0] and sv[-2] < 1e-12 * sv[0]      # singular, twice over
        svf = np.linalg.svd(Ax[:, free], compute_uv=False)
        assert svf[0] / svf[-1] < 100.0                                                          # regular once the pinned columns are known


def test_dual_residual_from_the_kkt_identity_changes_no_trajectory(oracle):
    """Round 6: the condensed register kernels take H x of boxADMM's dual residual from the KKT identity
    H x~ = (rhs_1 - A' nu) - (sigma + rho_box) o x~ instead of multiplying by H again (qp_base.hpp:240-252) — PIVOT_CONDSWEEP restates exactly that. Admission test,
    with the order held fixed and the identity switched on and off (orc_set_hx_identity, dense products): on the instance streams of configs A, B and the 16-node
    robot grid NO instance changes its SQP iterations, status or total ADMM iterations; the iterates move by the rounding of the adaptive-rho estimate only
    (<= 1e-9 on A and R; config B: <= 1e-4 absolute on controls bounded by 9000, i.e. 1e-8 of their range — measured 3.7e-5 under the Eigen order, 2e-7 under the kernel's)."""
    import ctypes as C
    from polympc_amd import workloads
    L = oracle.lib(); L.orc_set_hx_identity.argtypes = [C.c_int]; L.orc_set_hx_identity.restype = C.c_int
    for wl, B, model, tol in ((workloads.robot_batch(1024), 1024, oracle.MODEL_ROBOT, 1e-9), (workloads.cstr_batch(512), 512, oracle.MODEL_CSTR, 1e-4),
                              (workloads.robot_batch(256, P=5, S=3), 256, oracle.MODEL_ROBOT, 1e-9)):
        ss = oracle.sqp_default_settings(); ss.max_iter = wl["max_iter"]; ss.line_search_max_iter = wl["ls_max_iter"]
        res = []
        for on in (0, 1):
            old = L.orc_set_hx_identity(on)
            try:
                res.append(oracle.sqp_solve_batch(model, wl["P"], wl["S"], wl["t0"], wl["tf"], B, wl["d"], wl["lbx"], wl["ubx"], sqp_settings=ss, pivot=oracle.PIVOT_EIGEN, threads=4))
            finally:
                L.orc_set_hx_identity(old)
        (x0, l0, i0), (x1, l1, i1) = res
        assert [(i.iter, i.status, i.qp_solver_iter) for i in i0] == [(i.iter, i.status, i.qp_solver_iter) for i in i1]
        assert np.abs(x0 - x1).max() <= tol and np.abs(x0 - x1).max() > 0.0    # (the switch does something)


@pytest.mark.parametrize("ubg", [10.0, 1.2])
def test_condensed_register_order_with_a_path_constraint_and_a_parameter(oracle, ubg):
    """Round 6: PIVOT_CONDSWEEP with NG = 1 beside NP = 1 — the path-constraint rows behind the equality rows, each holding its own node's block only (no row of the
    differentiation matrix) — the order of the condensed register kernel that serves nonlinear_constraints_test.cpp:159-184 by default since round 6. Admission on 16 perturbed
    instances of that problem with the reference's bound (10: inactive) and a binding one (1.2), configured as the reference does (exact Hessians, Gershgorin): every instance keeps
    the SQP / ADMM iteration counts and the status of the run in the reference order (Eigen-style pivoted LDL^T) and of the run refined to exact arithmetic (PIVOT_EXACT), and the condensed
    order is the one that stays with exact arithmetic — 1e-8 against the reference order's 1e-6 and the full two-rows-per-lane inverse's (the kernel it replaces) 1e-5 .. 1e-4."""
    B, nn = 16, 11
    lbx, ubx, xg, d = _parking_batch(B, nn)
    lbg = np.full((B, nn), -10.0); ubgv = np.full((B, nn), ubg)
    ss = oracle.sqp_default_settings(); ss.max_iter = 20; ss.line_search_max_iter = 10; ss.regularisation = 2; ss.exact_hessian_every_iter = 1
    run = lambda piv: oracle.sqp_solve_batch(oracle.MODEL_PARKING_NG, 5, 2, 0.0, 1.0, B, d, lbx, ubx, lbg=lbg, ubg=ubgv, x_guess=xg, sqp_settings=ss, pivot=piv, threads=4)
    xc, lc, ic = run(oracle.PIVOT_CONDSWEEP)
    xe, le, ie = run(oracle.PIVOT_EIGEN)
    xx, lx, ix = run(oracle.PIVOT_EXACT)
    x2, l2, i2 = run(oracle.PIVOT_SWEEP2)
    key = lambda info: [(i.iter, i.status, i.qp_solver_iter) for i in info]
    assert key(ic) == key(ie) == key(ix)
    assert all(i.flags == 0 for i in ic)
    dc, de, d2 = np.abs(xc - xx).max(), np.abs(xe - xx).max(), np.abs(x2 - xx).max()
    assert dc < 1e-7 and dc < 0.05 * de and dc < 0.01 * d2, (dc, de, d2)


@pytest.mark.parametrize("policy", [dict(preconditioner=1), dict(preconditioner=1, line_search=1, hessian_update=1)], ids=["ruiz", "valet_parking_hooks"])
def test_condensed_register_order_under_the_ruiz_preconditioner(oracle, policy):
    """Late round 6: PIVOT_CONDSWEEP on a Ruiz-equilibrated QP (qp_preconditioners.hpp:114-220 either side of the solve, sqp_base.hpp:605-611) — the order of the condensed hook
    kernels that serve valet_parking_mpc_test.cpp's policy set by default (the kernel reads its D~ tables, one set per state index, and the node blocks back from the scaled workspace;
    the restatement reads the scaled KKT matrix anyway). Admission on 32 robot OCPs of that test's grid (11 nodes), Ruiz alone and with the filter line search + block BFGS: every
    instance keeps the SQP / ADMM iteration counts and the status of the run in the reference order and of the run refined to exact arithmetic, and the condensed order is at least as
    close to exact arithmetic as the reference order."""
    from polympc_amd import workloads
    B = 32
    wl = workloads.robot_batch(B, P=5, S=2)
    ss = oracle.sqp_default_settings(); ss.max_iter = 10; ss.line_search_max_iter = 10
    for k, v in policy.items(): setattr(ss, k, v)
    qs = oracle.sqp_qp_default_settings(); qs.max_iter = 1000
    run = lambda piv: oracle.sqp_solve_batch(oracle.MODEL_ROBOT, 5, 2, 0.0, 2.0, B, wl["d"], wl["lbx"], wl["ubx"], sqp_settings=ss, qp_settings=qs, pivot=piv, threads=4)
    xc, lc, ic = run(oracle.PIVOT_CONDSWEEP)
    xe, le, ie = run(oracle.PIVOT_EIGEN)
    xx, lx, ix = run(oracle.PIVOT_EXACT)
    key = lambda info: [(i.iter, i.status, i.qp_solver_iter) for i in info]
    assert key(ic) == key(ie) == key(ix)
    dc, de = np.abs(xc - xx).max(), np.abs(xe - xx).max()
    assert dc < 1e-9 and dc <= 2 * de, (dc, de)


def test_condensed_register_order_with_one_parameter_on_the_minimal_time_problem(oracle):
    """Round 6: PIVOT_CONDSWEEP with NP = 1 (the parameter's dense column of A' u as the wavefront's tree sum, its term last in every row of A x) — the order of the
    condensed register kernel that now serves the reference's minimal-time parking test (minimal_time_test.cpp:146-188: exact Hessians, Gershgorin, NP = 1) by default.
    Admission on 32 perturbed instances of that problem, as the reference configures it: every instance keeps the SQP / ADMM iteration counts and the status of the run
    as the reference computes (Eigen-style pivoted LDL^T, glibc) — and it is the condensed order that stays with exact arithmetic (PIVOT_EXACT: every linear solve refined
    in long double): 1e-7 against the reference order's own 1e-5 .. 1e-4. (No conditioning rule trips: controls and the parameter are bounded.)"""
    B = 32
    lbx, ubx, xg, d = _parking_batch(B)
    ss = oracle.sqp_default_settings(); ss.max_iter = 20; ss.line_search_max_iter = 10; ss.regularisation = 2; ss.exact_hessian_every_iter = 1
    run = lambda piv: oracle.sqp_solve_batch(oracle.MODEL_PARKING, 5, 2, 0.0, 1.0, B, d, lbx, ubx, x_guess=xg, sqp_settings=ss, pivot=piv, threads=4)
    xc, lc, ic = run(oracle.PIVOT_CONDSWEEP)
    xe, le, ie = run(oracle.PIVOT_EIGEN)
    xx, lx, ix = run(oracle.PIVOT_EXACT)
    key = lambda info: [(i.iter, i.status, i.qp_solver_iter) for i in info]
    assert key(ic) == key(ie) == key(ix)
    assert all(i.flags == 0 for i in ic) and sum(i.status == 0 for i in ic) >= B - 4
    dc, de = np.abs(xc - xx).max(), np.abs(xe - xx).max()
    assert dc < 1e-7 and dc < 0.01 * de, (dc, de)
