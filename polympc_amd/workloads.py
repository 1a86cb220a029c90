"""Synthetic batched OCP instances: counter-based splitmix64 -> U(-1,1), identical on every host so
the CPU and GPU runs see the same problem data. Pure input generation — no solver arithmetic here."""
import numpy as np

SEED = 20260929
_M = (1 << 64) - 1


def splitmix64(x):
    x = (np.asarray(x, dtype=np.uint64) + np.uint64(0x9E3779B97F4A7C15))
    z = x
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def uniform_pm1(seed, instance, j, K=1024):
    """U(-1,1) for (instance, j) from the counter seed ^ (instance*K + j)."""
    with np.errstate(over="ignore"):
        ctr = np.uint64(seed) ^ (np.asarray(instance, dtype=np.uint64) * np.uint64(K) + np.asarray(j, dtype=np.uint64))
        z = splitmix64(ctr)
    return (z >> np.uint64(11)).astype(np.float64) * (2.0 ** -53) * 2.0 - 1.0


def robot_batch(B, P=6, S=1, seed=SEED, perturb_d=False, first=0):
    """Config A / D: mobile robot, u in [-1.5,1.5]x[-0.75,0.75], states free, x0 = (0.5,0.5,0.5) + 0.4*U^3 pinned on
    the LAST nx entries of the x block (mpc_wrapper.hpp:89-93), d = 2 (A) or 2*(1+0.1*U) (D, perturbed wheel base)."""
    nn = P * S + 1
    n = 5 * nn
    inst = np.arange(first, first + B, dtype=np.uint64)
    lbx = np.full((B, n), -np.inf); ubx = np.full((B, n), np.inf)
    for j in range(3):
        x0 = 0.5 + 0.4 * uniform_pm1(seed, inst, j)
        lbx[:, 3 * nn - 3 + j] = x0; ubx[:, 3 * nn - 3 + j] = x0
    lbx[:, 3 * nn:] = np.tile([-1.5, -0.75], nn); ubx[:, 3 * nn:] = np.tile([1.5, 0.75], nn)
    d = np.full((B, 1), 2.0)
    if perturb_d:
        d[:, 0] = 2.0 * (1.0 + 0.1 * uniform_pm1(seed, inst, 3))
    return dict(model=0, P=P, S=S, t0=0.0, tf=2.0, d=d, lbx=lbx, ubx=ubx, n=n, m=3 * nn, max_iter=10, ls_max_iter=10)


def cstr_batch(B, seed=SEED, first=0):
    """Config B: CSTR (cstr_control_test.cpp:137-154), P=5 S=2, t in [0,100], u in [3,35]x[-9000,0],
    x0 = (1,0.5,100,100).*(1+0.05*U^4)."""
    P, S, nn = 5, 2, 11
    n = 6 * nn
    inst = np.arange(first, first + B, dtype=np.uint64)
    lbx = np.full((B, n), -np.inf); ubx = np.full((B, n), np.inf)
    base = [1.0, 0.5, 100.0, 100.0]
    for j in range(4):
        x0 = base[j] * (1.0 + 0.05 * uniform_pm1(seed, inst, j))
        lbx[:, 4 * nn - 4 + j] = x0; ubx[:, 4 * nn - 4 + j] = x0
    lbx[:, 4 * nn:] = np.tile([3.0, -9000.0], nn); ubx[:, 4 * nn:] = np.tile([35.0, 0.0], nn)
    return dict(model=1, P=P, S=S, t0=0.0, tf=100.0, d=np.zeros((B, 1)), lbx=lbx, ubx=ubx, n=n, m=4 * nn, max_iter=20, ls_max_iter=20)


def minimal_time_parking(nn=11):
    """minimal_time_test.cpp:146-184: parking OCP with a free time-scaling parameter (NP = 1), P=5 S=2 (nn = 11 nodes), d = 1,
    x0 = (1.5, .5, .5) pinned on the last node, final state within +-0.05 (first nx entries, mpc_wrapper.hpp:132-137), p in [0, 10],
    guesses p = 0.5 and x = x0 at every node. -> (lbx, ubx, x_guess), one instance each."""
    n = 5 * nn + 1
    lbx = np.full(n, -np.inf); ubx = np.full(n, np.inf)
    lbx[3 * nn:5 * nn] = np.tile([-1.5, -0.75], nn); ubx[3 * nn:5 * nn] = np.tile([1.5, 0.75], nn)
    lbx[5 * nn] = 0.0; ubx[5 * nn] = 10.0
    lbx[0:3] = -0.05; ubx[0:3] = 0.05
    lbx[3 * nn - 3:3 * nn] = [1.5, 0.5, 0.5]; ubx[3 * nn - 3:3 * nn] = [1.5, 0.5, 0.5]
    xg = np.zeros(n); xg[:3 * nn] = np.tile([1.5, 0.5, 0.5], nn); xg[5 * nn] = 0.5
    return lbx[None], ubx[None], xg[None]


def parking_batch(B, nn=11, seed=5):
    """B minimal-time parking problems (NP = 1) around the reference's: start states within +-0.2 of (1.5, .5, .5), wheel bases d in [0.8, 1.2].
    -> (lbx, ubx, x_guess, d)"""
    rng = np.random.default_rng(seed)
    lbx, ubx, xg = (np.repeat(a, B, 0) for a in minimal_time_parking(nn))
    x0 = np.array([1.5, 0.5, 0.5]) + 0.2 * rng.uniform(-1, 1, (B, 3))
    lbx[:, 3 * nn - 3:3 * nn] = x0; ubx[:, 3 * nn - 3:3 * nn] = x0
    xg[:, :3 * nn] = np.tile(x0, nn)
    return lbx, ubx, xg, 1.0 + 0.2 * rng.uniform(-1, 1, (B, 1))


def parking_reference_tests_batch(B, path_constraint=False, ubg=1.2):
    """The problems of the reference's two NP = 1 control tests as batches of perturbed instances, configured as those tests configure the solver (exact Hessians every
    iteration + Gershgorin shift, max_iter 20, ls 10): minimal_time_test.cpp:146-188 (model 2) and, with path_constraint, nonlinear_constraints_test.cpp:159-184
    (model 5: g = u0^2 cos u1 in [-10, ubg] per node). P=5 S=2."""
    nn = 11
    lbx, ubx, xg, d = parking_batch(B, nn)
    wl = dict(model=5 if path_constraint else 2, P=5, S=2, t0=0.0, tf=1.0, d=d, lbx=lbx, ubx=ubx, x_guess=xg, n=5 * nn + 1, m=(4 if path_constraint else 3) * nn,
              max_iter=20, ls_max_iter=10, settings=dict(regularisation=2, exact_hessian_every_iter=1))
    if path_constraint:
        wl["lbg"] = np.full((B, nn), -10.0); wl["ubg"] = np.full((B, nn), ubg)
    return wl


def valet_parking_policy_batch(B):
    """The solver configuration of valet_parking_mpc_test.cpp:161-165,183-241 — Ruiz preconditioner, filter line search, ContinuousOCP's block BFGS, QP iteration cap 1000 — on B
    randomised robot OCPs of that test's grid (P=5 S=2, 11 nodes; robot_batch's start states and bounds)."""
    wl = robot_batch(B, P=5, S=2)
    wl["settings"] = dict(preconditioner=1, line_search=1, hessian_update=1)
    wl["qp_max_iter"] = 1000
    return wl


def kite_standin_batch(B, seed=SEED, first=0):
    """Config C dimension stand-in: SYNTHETIC 13-state / 3-input smooth dynamics (the reference's KiteDynamics is not in
    the reference tree), P=5 S=3 -> 16 nodes, n=256, m=208, KKT 464 rows; u in [-1,1]^3, x0 = 0.3*U^13."""
    P, S, nn = 5, 3, 16
    n = 16 * nn
    inst = np.arange(first, first + B, dtype=np.uint64)
    lbx = np.full((B, n), -np.inf); ubx = np.full((B, n), np.inf)
    for j in range(13):
        x0 = 0.3 * uniform_pm1(seed, inst, j)
        lbx[:, 13 * nn - 13 + j] = x0; ubx[:, 13 * nn - 13 + j] = x0
    lbx[:, 13 * nn:] = -1.0; ubx[:, 13 * nn:] = 1.0
    return dict(model=4, P=P, S=S, t0=0.0, tf=1.0, d=np.zeros((B, 1)), lbx=lbx, ubx=ubx, n=n, m=13 * nn, max_iter=5, ls_max_iter=10)


def random_qp_batch(B, n, m, seed=SEED):
    """Dense strictly convex QPs with mixed equality / inequality / loose rows and boxes (any n, m)."""
    rng = np.random.default_rng(seed)
    G = rng.normal(size=(B, n, n))
    H = np.einsum("bij,bkj->bik", G, G) / n + 0.1 * np.eye(n)
    h = rng.normal(size=(B, n))
    A = rng.normal(size=(B, m, n))
    xf = rng.uniform(-0.5, 0.5, size=(B, n))              # a feasible point
    Ax = np.einsum("bij,bj->bi", A, xf)
    kind = rng.integers(0, 3, size=(B, m))
    Alb = np.where(kind == 0, Ax, np.where(kind == 1, Ax - rng.uniform(0.1, 1, (B, m)), -np.inf))
    Aub = np.where(kind == 0, Ax, np.where(kind == 1, Ax + rng.uniform(0.1, 1, (B, m)), np.inf))
    bk = rng.integers(0, 3, size=(B, n))
    xlb = np.where(bk == 0, xf, np.where(bk == 1, xf - rng.uniform(0.1, 1, (B, n)), -np.inf))
    xub = np.where(bk == 0, xf, np.where(bk == 1, xf + rng.uniform(0.1, 1, (B, n)), np.inf))
    Hc = np.ascontiguousarray(H.transpose(0, 2, 1)).reshape(B, n * n)      # column-major per instance
    Ac = np.ascontiguousarray(A.transpose(0, 2, 1)).reshape(B, m * n)
    return dict(H=Hc, h=h, A=Ac, Alb=Alb, Aub=Aub, xlb=xlb, xub=xub)
