"""GPU parity tests: the HIP path (through the C ABI of include/polympc_amd.h) against the oracle (static elimination
order — the order the kernels use) on the same seeded inputs, against the reference's golden vectors, and — at the
benchmark's full size — through size-independent KKT properties. Tolerances are stated per assertion; all fp64."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))   # oracle.cross_order (the shared cross-order record)

GOLD = os.path.join(os.path.dirname(__file__), "golden", "casadi_robot_P5S2.npz")
inf = np.inf


@pytest.fixture(scope="module")
def ctx():
    import polympc_amd as pa
    c = pa.Context(0)
    yield c
    c.close()


def _mat(Hc, n, m=None):
    """column-major flat -> [rows, cols]"""
    return Hc.reshape(n if m is None else n, -1).T if m is None else Hc.reshape(n, m).T


BIG_KKT_MIN_ROWS = 96   # pmpc_launch.hpp: from this many KKT rows the fused SQP kernel keeps its factor in HBM (blocked tile LDL^T)


QP_BIG_MIN_ROWS = 112   # pmpc_api.hip: the QP entry point's threshold for the same kernel family


def _lds_order(oracle, rows, qp_entry=False, ruiz=False, kkt_form=0):
    """The LDS-resident kernels (boxADMM above 64 KKT rows without a register specialisation, every policy the register paths do not carry, the
    stacked system of the OSQP-form ADMM): static right-looking LDL^T with fma substitutions (PIVOT_STATIC). Larger systems — 96 rows and more in the
    fused SQP kernel, 112 and more at the QP entry point, and everything whose packed triangle does not fit LDS (config C) — run the blocked tile
    LDL^T with the factor in HBM: same factor and forward pass, backward pass by column dot products (PIVOT_BLOCKED). Inside the fused SQP kernel
    that kernel works in condensed form since round 3 — the n x n matrix H + sigma I + rho_box + A' diag(rho) A in the same blocked order
    (PIVOT_CONDENSED) — unless the Ruiz preconditioner rescaled the workspace or pmpc_sqp_settings::kkt_form = 1 asks for the full KKT matrix."""
    if rows >= (QP_BIG_MIN_ROWS if qp_entry else BIG_KKT_MIN_ROWS):
        return oracle.PIVOT_BLOCKED if (qp_entry or ruiz or kkt_form == 1) else oracle.PIVOT_CONDENSED
    return oracle.PIVOT_STATIC


REG2_QP_SHAPES = ((66, 44), (55, 33), (45, 27), (50, 30), (60, 36), (65, 39), (54, 36), (60, 40), (80, 48), (75, 45), (72, 48))
REG1_QP_SHAPES = ((35, 21), (20, 12), (25, 15), (30, 18), (40, 24), (24, 16), (30, 20), (36, 24))   # one KKT row per lane (pmpc_api.hip)   # QP entry point: two-rows-per-lane register specialisations (pmpc_qp_reg2.hip)


REG_NODE_COUNTS = (3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16)   # grids of the built-in models with register-resident SQP kernels (pmpc_launch.hpp, pmpc_grids.hpp)


SCHUR_GRIDS = {0: ((5, 2), (5, 3)), 1: ((5, 2), (6, 1))}   # model -> (P, S) routed to a block-structured kernel (pmpc_schur_*.hip; more than 64 KKT rows)


def _schur_route(model, P, S, hessian_update=0, exact_hessian_every_iter=0, regularisation=0, preconditioner=0, qp_solver=0, line_search=0, kkt_form=0,
                 linear_solver=0, **_):
    """Does the request take the block-structured kernel (pmpc_launch.hpp: schur_request_ok + the compiled grids)? Its restatement is PIVOT_SCHUR."""
    return (P, S) in SCHUR_GRIDS.get(model, ()) and (hessian_update == 1 or exact_hessian_every_iter) and regularisation in (0, 2) and \
        not preconditioner and not qp_solver and (not line_search or P * S + 1 in (11, 16)) and not kkt_form and not linear_solver   # (round 5: the filter line search on the 11- and 16-node builds)


def _gpu_order(oracle, n, m, nodes=None, block_bfgs=False, kkt_form=0, schur=False, ng=0):
    """Which of the oracle's GPU-order linear-solve restatements mirrors the kernel that serves this size: the register-resident QP
    (compile-time sizes with n+m <= 64: the (35, 21) QP entry point, SQP grids of 3 to 8 nodes) applies the inverse swept in blocks of
    four pivots (PIVOT_SWEEP); the two-rows-per-lane register path (65..112 rows: the (66, 44) and (55, 33) QP entry points) the same sweep with
    its own mat-vec order (PIVOT_SWEEP2); every other size runs an LDS/HBM-resident kernel (_lds_order: PIVOT_STATIC). All of them are tied to the
    reference's pivoted Eigen LDLT (PIVOT_EIGEN) in tests/test_oracle_pins.py."""
    if schur:
        return oracle.PIVOT_SCHUR
    if nodes is None:
        if (n, m) in REG2_QP_SHAPES:
            return oracle.PIVOT_SWEEP2
        return oracle.PIVOT_SWEEP if (n, m) in REG1_QP_SHAPES else _lds_order(oracle, n + m, qp_entry=True)
    if n + m <= 64 and nodes in ((5, 7) if block_bfgs else REG_NODE_COUNTS):   # (the block-BFGS one-row-per-lane specialisation exists for 5 and 7 nodes)
        return oracle.PIVOT_SWEEP
    if 64 < n + m <= 128 and nodes in REG_NODE_COUNTS:      # two-rows-per-lane register path (113..128 rows: part of the operand tiles in LDS) (the Hessian update is a run-time choice there)
        if n <= 112 and m <= 64 and kkt_form == 0 and (n % nodes == 0 or (n - 1) % nodes == 0):    # (at most one parameter — NP = 1 and path constraints since round 6) condensed register kernel (pmpc_qp_cond.hpp): only S = H + sigma I + rho_box + A' diag(rho) A is inverted
            return oracle.PIVOT_CONDSWEEP
        return oracle.PIVOT_SWEEP2
    return _lds_order(oracle, n + m, kkt_form=kkt_form)


POLICY_REG_NODE_COUNTS = (7, 11)   # grids whose register-resident kernels exist with the Ruiz / filter-line-search hooks compiled in (pmpc_launch.hpp, POL); since round 4 also the 16-node grid where it has 128 rows


def _policy_order(oracle, n, m, nodes, ruiz=False, block_bfgs=False, kkt_form=0, ng=0):
    """preconditioner = 1 / line_search = 1: on the grids of the reference's own tests (7 and 11 nodes) the register-resident kernels carry these hooks
    since round 3 (their sweep orders); every other grid takes the LDS / HBM-resident kernels for them."""
    if (nodes in POLICY_REG_NODE_COUNTS and n + m <= 112) or (nodes == 16 and n + m <= 128):   # (16 nodes, round 4: the reference's mpc_wrapper_test grid)
        if n + m > 64 and kkt_form == 0 and n <= 112 and m <= 64 and (n % nodes == 0 or (n - 1) % nodes == 0):
            return oracle.PIVOT_CONDSWEEP   # the hooks keep the condensed register QP (since round 5 also on at most 64 variables; since late round 6 also with the Ruiz preconditioner: tables and node blocks from the scaled workspace)
        return oracle.PIVOT_SWEEP if n + m <= 64 else oracle.PIVOT_SWEEP2
    return _lds_order(oracle, n + m, ruiz=ruiz, kkt_form=kkt_form)


def _qp_oracle(oracle, q, s, x0=None, y0=None):
    os_ = oracle.qp_default_settings()
    for f, _ in s._fields_:
        setattr(os_, f, getattr(s, f))
    n, m = q["h"].shape[1], q["Alb"].shape[1]
    return oracle.qp_solve_batch(q["H"], q["h"], q["A"], q["Alb"], q["Aub"], q["xlb"], q["xub"], settings=os_,
                                 pivot=_gpu_order(oracle, n, m), x0=x0, y0=y0)



def _assert_same_solve(info, io, x, xo, lam=None, lo=None, bit=True, tol=1e-8):
    """GPU vs CPU restatement on EVERY instance (no mask): identical SQP iteration counts, statuses and total ADMM iterations; with
    bit=True (the kernels and the restatement evaluate the models with the same IEEE-only sin / cos / exp — pmpc_math.hpp — and the same
    order of linear algebra) the primal / dual solutions and the reported KKT quantities must be BIT-IDENTICAL — every policy is tested that
    way; bit=False (within `tol`, 1e-8 = north_star's fp64 tolerance) is kept for comparisons against a DIFFERENT order of the linear algebra."""
    assert np.array_equal(info["iter"], np.array([i.iter for i in io])), "SQP iteration counts differ"
    assert np.array_equal(info["status"], np.array([i.status for i in io])), "statuses differ"
    assert np.array_equal(info["qp_solver_iter"], np.array([i.qp_solver_iter for i in io])), "total ADMM iterations differ"
    q = {f: np.array([getattr(i, f) for i in io]) for f in ("primal_norm", "dual_norm", "max_violation", "cost")}
    if bit:
        assert np.array_equal(x, xo), f"x not bit-identical: max |dx| = {np.abs(x - xo).max():.3e}"
        if lam is not None:
            assert np.array_equal(lam, lo), f"lam not bit-identical: max |dlam| = {np.abs(lam - lo).max():.3e}"
        for f, v in q.items():
            assert np.array_equal(info[f], v), f
    else:
        assert np.abs(x - xo).max() <= tol * max(1.0, np.abs(xo).max())
        if lam is not None:
            assert np.abs(lam - lo).max() <= tol * max(1.0, np.abs(lo).max())
        for f, v in q.items():
            assert np.abs(info[f] - v).max() <= tol * max(1.0, np.abs(v).max()), f


# -------------------------------------------------------------------------------------------- A16-A18: box-ADMM QP
def test_qp_reference_known_answers(ctx, oracle):
    """box_admm_test.cpp:15-45, :266-297, :299-334 through the GPU path."""
    import polympc_amd as pa
    H = np.array([[4.0, 1.0], [1.0, 2.0]]).T.ravel()[None]
    s = pa.qp_settings_default(); s.max_iter = 150
    x, y, info = ctx.qp_solve_batch(H, [[1.0, 1.0]], [[1.0, 1.0]], [[1.0]], [[1.0]], [[0.0, 0.0]], [[0.7, 0.7]], settings=s)
    assert np.linalg.norm(x[0] - [0.3, 0.7]) <= 1e-2 * np.linalg.norm([0.3, 0.7])
    assert info["iter"][0] < 150 and info["status"][0] == pa.QP_SOLVED
    z = np.zeros((1, 0))
    s = pa.qp_settings_default(); s.max_iter = 200; s.adaptive_rho = 1; s.check_termination = 10
    x, y, info = ctx.qp_solve_batch(np.zeros((1, 1)), np.ones((1, 1)), z, z, z, [[-1e6]], [[1e6]], settings=s)
    assert abs(x[0, 0] + 1e6) <= 1e4 and info["iter"][0] < 200 and info["status"][0] == pa.QP_SOLVED
    s.rho = 2
    x, y, info = ctx.qp_solve_batch(-np.ones((1, 1)), np.zeros((1, 1)), z, z, z, [[-1.0]], [[2.0]], settings=s, x0=[[0.1]], y0=[[0.1]])
    assert abs(x[0, 0] - 2.0) <= 2e-2 and info["iter"][0] < 200 and info["status"][0] == pa.QP_SOLVED


@pytest.mark.parametrize("n,m,B", [(2, 1, 8), (1, 0, 4), (7, 3, 33), (35, 21, 64), (55, 33, 16), (66, 44, 8), (80, 48, 4), (3, 70, 4), (60, 36, 5), (105, 63, 3), (256, 208, 3), (70, 43, 4), (64, 65, 3), (130, 0, 3), (5, 140, 2), (20, 12, 9), (40, 24, 5), (36, 24, 5), (45, 27, 4), (60, 36, 3), (65, 39, 3), (60, 40, 3), (75, 45, 3), (72, 48, 3)])
def test_qp_random_vs_oracle(ctx, oracle, n, m, B):
    """Random convex QPs of many shapes (incl. ragged n+m > 64, m > n, m = 0): same iteration count, status and
    rho updates as the oracle; x, y and the reported residuals bit-identical (register, two-rows-per-lane, LDS and HBM-factor kernels)."""
    import polympc_amd as pa
    from polympc_amd import workloads
    q = workloads.random_qp_batch(B, n, m, seed=n * 1000 + m)
    s = pa.qp_settings_sqp_default()
    x, y, info = ctx.qp_solve_batch(q["H"], q["h"], q["A"], q["Alb"], q["Aub"], q["xlb"], q["xub"], settings=s)
    xo, yo, io = _qp_oracle(oracle, q, s)
    assert [int(i) for i in info["iter"]] == [i.iter for i in io]
    assert [int(i) for i in info["status"]] == [i.status for i in io]
    assert [int(i) for i in info["rho_updates"]] == [i.rho_updates for i in io]
    assert np.array_equal(x, xo) and np.array_equal(y, yo), (np.abs(x - xo).max(), np.abs(y - yo).max())
    assert np.array_equal(info["res_prim"], np.array([i.res_prim for i in io])) and np.array_equal(info["res_dual"], np.array([i.res_dual for i in io]))


@pytest.mark.parametrize("n,m,B", [(35, 21, 8), (55, 33, 5), (30, 50, 4), (100, 60, 3)])
def test_qp_hessian_is_read_from_its_lower_triangle_on_every_kernel_family(ctx, oracle, n, m, B):
    """The reference factorises with Eigen::LDLT<Matrix, Lower> (helpers.hpp:38-43): the KKT matrix sees the LOWER triangle of H only, while the
    residuals use H as given (qp_base.hpp:240-252). With an H that is NOT bitwise symmetric (the upper triangle perturbed by 1e-3) the result must not
    depend on whether (n, m) happens to have a register-resident specialisation: one-row-per-lane (35, 21), two-rows-per-lane (55, 33), LDS (30, 50)
    and HBM-factor (100, 60) kernels against the restatement, bit for bit."""
    import polympc_amd as pa
    from polympc_amd import workloads
    q = workloads.random_qp_batch(B, n, m, seed=n * 1000 + m + 7)
    rng = np.random.default_rng(n + m)
    Hm = q["H"].reshape(B, n, n).transpose(0, 2, 1).copy()            # [b, row, col]
    Hm += 1e-3 * np.triu(rng.normal(size=(B, n, n)), 1)                 # upper triangle only
    q["H"] = np.ascontiguousarray(Hm.transpose(0, 2, 1)).reshape(B, n * n)
    s = pa.qp_settings_sqp_default()
    x, y, info = ctx.qp_solve_batch(q["H"], q["h"], q["A"], q["Alb"], q["Aub"], q["xlb"], q["xub"], settings=s)
    xo, yo, io = _qp_oracle(oracle, q, s)
    assert [int(i) for i in info["iter"]] == [i.iter for i in io] and [int(i) for i in info["status"]] == [i.status for i in io]
    assert np.array_equal(x, xo) and np.array_equal(y, yo), (np.abs(x - xo).max(), np.abs(y - yo).max())


@pytest.mark.parametrize("n,m,B", [(7, 3, 9), (35, 21, 16), (66, 44, 6), (20, 45, 4)])
def test_qp_pivoted_linear_solver_vs_eigen_order(ctx, oracle, n, m, B):
    """linear_solver = 1 (Eigen::LDLT's diagonal pivoting on the device, LDS-resident kernel) against the CPU restatement's Eigen-style policy —
    the restatement of what the reference itself runs: identical iteration counts, statuses and rho updates, bit-identical x, y and residuals."""
    import polympc_amd as pa
    from polympc_amd import workloads
    q = workloads.random_qp_batch(B, n, m, seed=n * 1000 + m + 1)
    s = pa.qp_settings_sqp_default(); s.linear_solver = 1
    x, y, info = ctx.qp_solve_batch(q["H"], q["h"], q["A"], q["Alb"], q["Aub"], q["xlb"], q["xub"], settings=s)
    os_ = oracle.qp_default_settings()
    for f, _ in os_._fields_:
        setattr(os_, f, getattr(s, f))
    xo, yo, io = oracle.qp_solve_batch(q["H"], q["h"], q["A"], q["Alb"], q["Aub"], q["xlb"], q["xub"], settings=os_, pivot=oracle.PIVOT_EIGEN)
    assert [int(i) for i in info["iter"]] == [i.iter for i in io] and [int(i) for i in info["status"]] == [i.status for i in io]
    assert [int(i) for i in info["rho_updates"]] == [i.rho_updates for i in io]
    assert np.array_equal(x, xo) and np.array_equal(y, yo)
    assert np.array_equal(info["res_prim"], [i.res_prim for i in io]) and np.array_equal(info["res_dual"], [i.res_dual for i in io])


def test_qp_pivoted_linear_solver_on_an_indefinite_hessian(ctx, oracle):
    """box_admm_test.cpp:299-334 (H = -1, -1 <= x <= 2, rho = 2: a non-convex QP the reference solves to x = 2) and an indefinite 6 x 6 Hessian with a
    zero leading entry — a zero first pivot in the static order (flagged non-finite there), factorised by the pivoted solver like the restatement."""
    import polympc_amd as pa
    z = np.zeros((1, 0))
    s = pa.qp_settings_default(); s.max_iter = 200; s.adaptive_rho = 1; s.check_termination = 10; s.rho = 2; s.linear_solver = 1
    x, y, info = ctx.qp_solve_batch(-np.ones((1, 1)), np.zeros((1, 1)), z, z, z, [[-1.0]], [[2.0]], settings=s, x0=[[0.1]], y0=[[0.1]])
    assert abs(x[0, 0] - 2.0) <= 2e-2 and info["status"][0] == pa.QP_SOLVED
    n = 6
    H = np.diag([-(1e-6 + 0.1), 1.0, 2.0, 3.0, 4.0, 5.0]); H[0, 1] = H[1, 0] = 1.0     # K(0,0) = H00 + sigma + rho_box = 0 exactly
    Hc = H.T.reshape(1, n * n).copy(); h = np.ones((1, n)); lb = -np.ones((1, n)); ub = np.ones((1, n))
    res = {}
    for ls in (0, 1):
        s = pa.qp_settings_default(); s.max_iter = 50; s.check_termination = 10; s.linear_solver = ls
        res[ls] = ctx.qp_solve_batch(Hc, h, z, z, z, lb, ub, settings=s)
    assert res[0][2]["flags"][0] == pa.capi.FLAG_NONFINITE           # static order: zero pivot, reported
    x1, y1, i1 = res[1]
    assert i1["flags"][0] == 0 and np.isfinite(x1).all()
    os_ = oracle.qp_default_settings(); os_.max_iter = 50; os_.check_termination = 10
    xo, yo, io = oracle.qp_solve_batch(Hc, h, z, z, z, lb, ub, settings=os_, pivot=oracle.PIVOT_EIGEN)
    assert int(i1["iter"][0]) == io[0].iter and np.array_equal(x1, xo) and np.array_equal(y1, yo)


def test_qp_warm_start_guess(ctx, oracle):
    import polympc_amd as pa
    from polympc_amd import workloads
    B, n, m = 8, 12, 5
    q = workloads.random_qp_batch(B, n, m, seed=77)
    rng = np.random.default_rng(1)
    x0 = rng.normal(size=(B, n)) * 0.1; y0 = rng.normal(size=(B, n + m)) * 0.1
    s = pa.qp_settings_sqp_default()
    x, y, info = ctx.qp_solve_batch(q["H"], q["h"], q["A"], q["Alb"], q["Aub"], q["xlb"], q["xub"], settings=s, x0=x0, y0=y0)
    xo, yo, io = _qp_oracle(oracle, q, s, x0=x0, y0=y0)
    assert [int(i) for i in info["iter"]] == [i.iter for i in io]
    assert np.abs(x - xo).max() <= 1e-9 and np.abs(y - yo).max() <= 1e-9


def test_qp_from_sqp_trace_vs_oracle(ctx, oracle):
    """The QPs the oracle's SQP emits for config-A robot instances (true collocation structure and conditioning)."""
    import polympc_amd as pa
    from polympc_amd import workloads
    wl = workloads.robot_batch(6)
    ss = oracle.sqp_default_settings(); ss.max_iter = 10; ss.line_search_max_iter = 10
    Hs, hs, As, al, au, lx, ux = [], [], [], [], [], [], []
    for b in range(6):
        t = oracle.sqp_trace_qps(oracle.MODEL_ROBOT, 6, 1, 0.0, 2.0, wl["d"][b:b + 1], wl["lbx"][b:b + 1], wl["ubx"][b:b + 1],
                                 sqp_settings=ss, pivot=oracle.PIVOT_STATIC)
        Hs.append(t["H"]); hs.append(t["h"]); As.append(t["A"]); al.append(t["al"]); au.append(t["au"]); lx.append(t["lx"]); ux.append(t["ux"])
    q = dict(H=np.concatenate(Hs), h=np.concatenate(hs), A=np.concatenate(As), Alb=np.concatenate(al), Aub=np.concatenate(au),
             xlb=np.concatenate(lx), xub=np.concatenate(ux))
    s = pa.qp_settings_sqp_default()
    x, y, info = ctx.qp_solve_batch(q["H"], q["h"], q["A"], q["Alb"], q["Aub"], q["xlb"], q["xub"], settings=s)
    xo, yo, io = _qp_oracle(oracle, q, s)
    assert [int(i) for i in info["iter"]] == [i.iter for i in io]
    assert np.abs(x - xo).max() <= 1e-9 and np.abs(y - yo).max() <= 1e-8
    assert np.abs(info["res_prim"] - [i.res_prim for i in io]).max() <= 1e-8    # north_star: KKT residual within 1e-8
    assert np.abs(info["res_dual"] - [i.res_dual for i in io]).max() <= 1e-8


QP_STREAMS = (("A", 4096), ("D", 1024), ("B", 2048), ("R", 1024), ("C", 128))   # configuration, least number of QPs (oracle/cross_order.traced_qp_stream)


@pytest.mark.parametrize("cfg,min_qps", QP_STREAMS)
def test_qp_level_parity_against_the_reference_order(ctx, oracle, cfg, min_qps):
    """north_star's literal criterion — "primal/dual KKT residual within 1e-8 of CPU reference" — on its own unit, ONE box-ADMM solve (SURVEY 8d:
    "max |D| of (x, y, res_prim, res_dual) GPU-vs-CPU"): the QPs the reference-order SQP emits for the BASELINE configurations (A >= 4096 QPs,
    D >= 1024, B >= 2048, the reference's 16-node grid >= 1024, C >= 128) solved by the DEFAULT kernels behind pmpc_qp_boxadmm_solve_batch (one row per
    lane, two rows per lane, HBM factor) against the restatement AS THE REFERENCE COMPUTES — Eigen-style pivoted LDL^T (PIVOT_EIGEN). Every QP, no mask:
    identical ADMM iteration counts, statuses and rho updates, and the reported residuals (qp_base.hpp:240-252, box_admm.hpp:398-431) within 1e-8."""
    import polympc_amd as pa
    from oracle import cross_order as tco
    q = tco.traced_qp_stream(oracle, cfg, min_qps)
    assert q["H"].shape[0] >= min_qps
    s = pa.qp_settings_sqp_default()
    x, y, info = ctx.qp_solve_batch(q["H"], q["h"], q["A"], q["Alb"], q["Aub"], q["xlb"], q["xub"], settings=s)
    xr, yr, ir = tco.reference_qp_solve(oracle, q, threads=8)
    rec = tco.qp_level_stats(x, y, info, xr, yr, ir)
    print(cfg, rec)
    assert rec["different_iter"] == 0 and rec["different_status"] == 0 and rec["different_rho_updates"] == 0, rec
    assert rec["max_abs_d_res_prim"] <= 1e-8 and rec["max_abs_d_res_dual"] <= 1e-8, rec      # north_star's tolerance
    assert rec["abs_dx_per_qp"]["max"] <= 1e-7 and rec["scaled_dy_per_qp"]["max"] <= 1e-7, rec   # (measured: <= 5e-9 / <= 2e-10)


def test_qp_kkt_properties_full_size(ctx):
    """Config-A-sized batch of 4096 QPs: every SOLVED instance satisfies the reference's own termination inequalities
    when the residuals are recomputed on the host from the returned (x, y) (size-independent property)."""
    import polympc_amd as pa
    from polympc_amd import workloads
    B, n, m = 4096, 35, 21
    q = workloads.random_qp_batch(B, n, m, seed=4096)
    s = pa.qp_settings_sqp_default()
    x, y, info = ctx.qp_solve_batch(q["H"], q["h"], q["A"], q["Alb"], q["Aub"], q["xlb"], q["xub"], settings=s)
    H = q["H"].reshape(B, n, n).transpose(0, 2, 1); A = q["A"].reshape(B, n, m).transpose(0, 2, 1)
    rd = np.abs(np.einsum("bij,bj->bi", H, x) + q["h"] + np.einsum("bji,bj->bi", A, y[:, :m]) + y[:, m:]).max(axis=1)
    ok = info["status"] == pa.QP_SOLVED
    assert ok.sum() > 0.3 * B     # random dense QPs: many need more than the SQP-default cap of 100 ADMM iterations
    assert np.all(info["iter"][~ok] == 101) and np.all(info["status"][~ok] == pa.QP_MAX_ITER_EXCEEDED)   # Q5: iter = max_iter+1
    assert np.abs(rd[ok] - info["res_dual"][ok]).max() <= 1e-9
    Ax = np.einsum("bij,bj->bi", A, x)
    assert np.all(Ax[ok] >= q["Alb"][ok] - 1e-2) and np.all(Ax[ok] <= q["Aub"][ok] + 1e-2)
    assert np.all(x[ok] >= q["xlb"][ok] - 1e-2) and np.all(x[ok] <= q["xub"][ok] + 1e-2)


# -------------------------------------------------------------------------------------------- A1-A11: collocation
def test_collocation_against_reference_golden(ctx):
    """GPU assembly vs the vectors produced by the reference's own CasADi fixtures (tests/golden/make_golden.py)."""
    import polympc_amd as pa
    g = np.load(GOLD)
    K = len(g["x"])
    lam = np.concatenate([g["lam"], np.zeros((K, 55))], axis=1)
    ev = ctx.ocp_linearise_batch(pa.MODEL_ROBOT, 5, 2, 0.0, 1.0, g["x"], np.ones((K, 1)), lam=lam)
    assert np.abs(ev["cost"] - g["cost"]).max() <= 1e-13
    assert np.abs(ev["cost_values_only"] - g["cost"]).max() <= 1e-13
    assert np.abs(ev["c"] - g["c"]).max() <= 1e-13
    assert np.abs(ev["jac"] - g["jac"]).max() <= 1e-13
    assert np.abs(ev["cost_grad"] - g["cost_grad"]).max() <= 1e-13
    assert np.abs(ev["lag_grad"] - g["lag_grad"]).max() <= 1e-13
    assert np.abs(ev["lag_hess"] - g["lag_hess"]).max() <= 1e-13
    # A8 on its own: with zero multipliers the Lagrangian Hessian IS cost_gradient_hessian's block Hessian (continuous_ocp.hpp:1256-1367)
    ev0 = ctx.ocp_linearise_batch(pa.MODEL_ROBOT, 5, 2, 0.0, 1.0, g["x"], np.ones((K, 1)), lam=np.zeros((K, 33 + 55)))
    assert np.abs(ev0["lag_hess"] - g["cost_hess"]).max() <= 1e-13
    assert np.abs(ev0["lag_grad"] - g["cost_grad"]).max() <= 1e-13


@pytest.mark.parametrize("model,P,S,t0,tf,nd", [(0, 6, 1, 0.0, 2.0, 1), (0, 5, 3, 0.0, 2.0, 1), (1, 5, 2, 0.0, 100.0, 0), (2, 5, 2, 0.0, 1.0, 1),
                                                (3, 4, 2, 0.0, 1.5, 1), (0, 3, 2, 0.0, 2.0, 1), (0, 15, 1, 0.0, 2.0, 1)])
def test_collocation_vs_oracle(ctx, oracle, model, P, S, t0, tf, nd):
    """All models (NP = 1 border blocks, NG = 1 path constraints, exp-heavy CSTR), several (P, S): 1e-12 relative."""
    import polympc_amd as pa
    dm = pa.ocp_dims(model, P, S)
    assert dm == oracle.ocp_dims(model, P, S)
    n, m = dm["n"], dm["m"]
    rng = np.random.default_rng(model * 100 + P * 10 + S)
    B = 3
    var = rng.uniform(-1, 1, (B, n))
    if model == 1:
        var[:, :4 * dm["nn"]] = np.tile([2.0, 1.0, 110.0, 110.0], dm["nn"]) * (1 + 0.05 * var[:, :4 * dm["nn"]])
        var[:, 4 * dm["nn"]:] = np.tile([14.0, -1000.0], dm["nn"]) * (1 + 0.05 * var[:, 4 * dm["nn"]:])
    lam = rng.uniform(-1, 1, (B, m + n))
    d = np.full((B, max(nd, 1)), 2.0)
    ev = ctx.ocp_linearise_batch(model, P, S, t0, tf, var, d, lam=lam)
    for b in range(B):
        eo = oracle.ocp_eval(model, P, S, t0, tf, var[b], d[b], lam=lam[b])
        def close(a, r, tol=1e-12):
            assert np.abs(a - r).max() <= tol * max(1.0, np.abs(r).max()), (np.abs(a - r).max(), np.abs(r).max())
        close(ev["cost"][b], eo["cost"]); close(ev["cost_values_only"][b], eo["cost"])
        close(ev["c"][b], np.concatenate([eo["c"], eo["g"]]))
        close(ev["jac"][b], eo["jac"]); close(ev["cost_grad"][b], eo["cost_grad"]); close(ev["lag_grad"][b], eo["lag_grad"])
        close(ev["lag_hess"][b], eo["lag_hess"])


def test_pivot_reciprocal_equals_ieee_division():
    """The 8-operation reciprocal of the sweep pivots (v_rcp_f64 + 2 Newton steps + residual correction + v_div_fixup) gives the
    correctly rounded quotient: 0 mismatches against 1.0 / d on 2^26 doubles with exponents in [-1000, 1000] and the special
    values (tests/experiments/recip_check.hip, built by __graft_entry__.build())."""
    import subprocess
    exe = os.path.join(os.path.dirname(__file__), "experiments", "recip_check")
    if not os.path.exists(exe):
        pytest.skip("tests/experiments/recip_check not built")
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and " 0 mismatches in the normal range" in out.stdout, out.stdout + out.stderr


# -------------------------------------------------------------------------------------------- §8f-4: OSQP-style ADMM
def test_admm_reference_known_answers(ctx, oracle):
    """admm_solver_test.cpp:16-45, :303-334, :336-371 through the GPU path."""
    import polympc_amd as pa
    H = np.array([[4.0, 1.0], [1.0, 2.0]]).T.ravel()[None]
    s = pa.qp_settings_default(); s.max_iter = 1000
    x, y, info = ctx.qp_admm_solve_batch(H, [[1.0, 1.0]], [[1.0, 1.0]], [[1.0]], [[1.0]], [[0.0, 0.0]], [[0.7, 0.7]], settings=s)
    assert np.linalg.norm(x[0] - [0.3, 0.7]) <= 1e-2 * np.linalg.norm([0.3, 0.7]) and info["iter"][0] < 1000 and info["status"][0] == pa.QP_SOLVED
    z = np.zeros((1, 0))
    s = pa.qp_settings_default(); s.max_iter = 200; s.adaptive_rho = 1; s.check_termination = 10
    x, y, info = ctx.qp_admm_solve_batch(np.zeros((1, 1)), np.ones((1, 1)), z, z, z, [[-1e6]], [[1e6]], settings=s)
    assert abs(x[0, 0] + 1e6) <= 1e4 and info["iter"][0] < 200 and info["status"][0] == pa.QP_SOLVED
    s.rho = 2
    x, y, info = ctx.qp_admm_solve_batch(-np.ones((1, 1)), np.zeros((1, 1)), z, z, z, [[-1.0]], [[2.0]], settings=s, x0=[[0.1]], y0=[[0.1]])
    assert abs(x[0, 0] - 2.0) <= 2e-2 and info["iter"][0] < 200 and info["status"][0] == pa.QP_SOLVED


@pytest.mark.parametrize("n,m,B", [(2, 1, 8), (1, 0, 4), (7, 3, 33), (35, 21, 16), (20, 45, 4)])
def test_admm_random_vs_oracle(ctx, oracle, n, m, B):
    """Random QPs (mixed equality / inequality / loose rows and boxes), adaptive rho on: identical iteration counts, status and
    rho updates as the CPU restatement in the same static elimination order; |dx|, |dy|, residuals within 1e-9."""
    import polympc_amd as pa
    from polympc_amd import workloads
    q = workloads.random_qp_batch(B, n, m, seed=7)
    s = pa.qp_settings_default(); s.max_iter = 400; s.adaptive_rho = 1; s.adaptive_rho_interval = 25; s.check_termination = 25
    x, y, info = ctx.qp_admm_solve_batch(q["H"], q["h"], q["A"], q["Alb"], q["Aub"], q["xlb"], q["xub"], settings=s)
    os_ = oracle.qp_default_settings()
    for f, _ in s._fields_:
        setattr(os_, f, getattr(s, f))
    xo, yo, io = oracle.qp_admm_solve_batch(q["H"], q["h"], q["A"], q["Alb"], q["Aub"], q["xlb"], q["xub"], settings=os_, pivot=oracle.PIVOT_STATIC)
    assert list(info["iter"]) == [i.iter for i in io] and list(info["status"]) == [i.status for i in io]
    assert list(info["rho_updates"]) == [i.rho_updates for i in io]
    assert np.abs(x - xo).max() <= 1e-9 and np.abs(y - yo).max() <= 1e-9 * max(1.0, np.abs(yo).max())
    assert np.abs(info["res_prim"] - [i.res_prim for i in io]).max() <= 1e-9 and np.abs(info["res_dual"] - [i.res_dual for i in io]).max() <= 1e-9


# -------------------------------------------------------------------------------------------- §8f-2: Ruiz equilibration
def test_ruiz_reference_known_answer(ctx, oracle):
    """box_admm_test.cpp:47-83 through the GPU path: compute -> solve -> unscale gives (0.3, 0.7), SOLVED in < 150 iterations."""
    import polympc_amd as pa
    H = np.array([[4.0, 1.0], [1.0, 2.0]]).T.ravel()[None]
    Hs, hs, As, al, au, xl, xu, D, E, c = ctx.qp_ruiz_compute_batch(H, [[1.0, 1.0]], [[1.0, 1.0]], [[1.0]], [[1.0]], [[0.0, 0.0]], [[0.7, 0.7]])
    s = pa.qp_settings_default(); s.max_iter = 150
    x, y, info = ctx.qp_solve_batch(Hs, hs, As, al, au, xl, xu, settings=s)
    sol, dual = ctx.qp_ruiz_unscale_batch(D, E, c, x, y)
    assert np.linalg.norm(sol[0] - [0.3, 0.7]) <= 1e-2 * np.linalg.norm([0.3, 0.7])
    assert info["iter"][0] < 150 and info["status"][0] == pa.QP_SOLVED


@pytest.mark.parametrize("n,m,B,scale", [(2, 1, 4, 1.0), (7, 3, 9, 50.0), (35, 21, 16, 1e-3), (66, 44, 4, 1.0), (5, 0, 3, 1.0), (3, 70, 2, 7.0)])
def test_ruiz_compute_bit_exact_vs_oracle(ctx, oracle, n, m, B, scale):
    """Scaled problem data, D, E, c and the unscaled solution are IDENTICAL to the CPU restatement (sqrt, division and the
    association order of the diagonal products are the IEEE operations of the reference's expressions). Small `scale`
    keeps the norms below 1 so that all four sweeps run; ragged shapes (m = 0, m > 64) included."""
    from polympc_amd import workloads
    q = workloads.random_qp_batch(B, n, m, seed=11)
    args = (q["H"] * scale, q["h"], q["A"] * scale, q["Alb"], q["Aub"], q["xlb"], q["xub"])
    g = ctx.qp_ruiz_compute_batch(*args)
    o = oracle.ruiz_compute_batch(*args)
    for a, b_ in zip(g, o):
        assert np.array_equal(np.asarray(a).reshape(-1), np.asarray(b_).reshape(-1))
    rng = np.random.default_rng(1)
    x = rng.normal(size=(B, n)); y = rng.normal(size=(B, n + m))
    gx, gy = ctx.qp_ruiz_unscale_batch(g[7], g[8], g[9], x, y)
    ox, oy = oracle.ruiz_unscale_solution_batch(o[7], o[8], o[9], x, y)
    assert np.array_equal(gx, ox) and np.array_equal(gy, oy)


def test_sqp_with_ruiz_preconditioner_vs_oracle(ctx, oracle):
    """SQPBase<..., RuizEquilibration> (sqp_base.hpp:605-611, :661-665): the fused kernel scales / unscales the QP data in place
    around every QP exactly as the restatement does (7-node and 11-node grids, both on the LDS-resident QP kernels)."""
    from polympc_amd import workloads
    for P, S, B in ((6, 1, 48), (5, 2, 8)):
        (x, lam, info), (xo, lo, io) = _sqp_both(ctx, oracle, workloads.robot_batch(B, P=P, S=S), B, preconditioner=1)
        _assert_same_solve(info, io, x, xo, lam, lo)


@pytest.mark.gpu
@pytest.mark.parametrize("osqp_form", [False, True])
@pytest.mark.parametrize("n,m,B", [(2, 1, 1), (7, 3, 33), (35, 21, 16), (5, 0, 4), (66, 44, 3), (3, 70, 2), (40, 24, 3)])
def test_qp_single_precision_vs_oracle(ctx, oracle, n, m, B, osqp_form):
    """pmpc_qp_boxadmm_solve_batch_f32 = boxADMM<N, M, float> (box_admm_test.cpp:85-115) and pmpc_qp_admm_solve_batch_f32 = ADMM<N, M, float>
    (admm_solver_test.cpp:84-113, the stacked (2n+m)-row system): the reference's float fixture (n = 2, m = 1) and random QPs,
    cold and warm-started, with adaptive rho — identical iteration counts, statuses, rho updates and BIT-IDENTICAL float x / y / residuals against
    the float restatement in the kernel's static order; the fixture's own assertions hold on the GPU result."""
    import polympc_amd as pa
    if osqp_form and 2 * n + m > 128:
        with pytest.raises(RuntimeError):   # the stacked system must fit two rows per lane
            z = np.zeros((1, n), dtype=np.float32)
            ctx.qp_solve_batch_f32(np.eye(n, dtype=np.float32).reshape(1, -1), z, np.zeros((1, m * n)), np.zeros((1, m)), np.zeros((1, m)), z - 1, z + 1, osqp_form=True)
        return
    if (n, m, B) == (2, 1, 1):
        H = np.array([[4, 1, 1, 2]], dtype=np.float32); h = np.array([[1, 1]], dtype=np.float32); A = np.array([[1, 1]], dtype=np.float32)
        al = np.array([[1]], dtype=np.float32); au = al.copy(); xl = np.zeros((1, 2), dtype=np.float32); xu = np.full((1, 2), 0.7, dtype=np.float32)
    else:
        from polympc_amd import workloads
        q = workloads.random_qp_batch(B, n, m, seed=n * 1000 + m)
        H, h, A, al, au, xl, xu = (np.asarray(q[k], dtype=np.float32) for k in ("H", "h", "A", "Alb", "Aub", "xlb", "xub"))
        al = al.reshape(B, m); au = au.reshape(B, m)
    for adaptive, iters in ((0, 150), (1, 400)):
        s = pa.qp_settings_default(); s.max_iter = iters; s.adaptive_rho = adaptive; s.adaptive_rho_interval = 25
        so = oracle.qp_default_settings(); so.max_iter = iters; so.adaptive_rho = adaptive; so.adaptive_rho_interval = 25
        x0 = y0 = None
        for warm in (False, True):
            x, y, info = ctx.qp_solve_batch_f32(H, h, A, al, au, xl, xu, settings=s, x0=x0, y0=y0, osqp_form=osqp_form)
            xo, yo, io = oracle.qp_solve_batch_f32(H, h, A, al, au, xl, xu, settings=so, pivot=oracle.PIVOT_STATIC, x0=x0, y0=y0, osqp_form=osqp_form)
            assert np.array_equal(info["iter"], [i.iter for i in io]) and np.array_equal(info["status"], [i.status for i in io])
            assert np.array_equal(info["rho_updates"], [i.rho_updates for i in io])
            assert x.dtype == np.float32 and x.tobytes() == xo.tobytes() and y.tobytes() == yo.tobytes()
            assert np.array_equal(info["res_prim"], [i.res_prim for i in io]) and np.array_equal(info["res_dual"], [i.res_dual for i in io])
            assert np.all(info["flags"] == 0)
            x0, y0 = x, y
    if (n, m, B) == (2, 1, 1):
        sol = np.array([0.3, 0.7], dtype=np.float32)
        s = pa.qp_settings_default(); s.max_iter = 1000 if osqp_form else 150   # (the ADMM test keeps the default max_iter)
        x, y, info = ctx.qp_solve_batch_f32(H, h, A, al, au, xl, xu, settings=s, osqp_form=osqp_form)
        assert np.linalg.norm(x[0] - sol) <= 1e-2 * min(np.linalg.norm(x[0]), np.linalg.norm(sol))
        assert info["iter"][0] < s.max_iter and info["status"][0] == pa.QP_SOLVED
    with pytest.raises(RuntimeError):   # the matrix lives in LDS, two KKT rows per lane
        z = np.zeros((1, 130), dtype=np.float32)
        ctx.qp_solve_batch_f32(np.eye(130, dtype=np.float32).reshape(1, -1), z, np.zeros((1, 0)), np.zeros((1, 0)), np.zeros((1, 0)), z - 1, z + 1)


# -------------------------------------------------------------------------------------------- §8f-1: batched MPC step
def test_mpc_receding_horizon_device_resident(ctx, oracle):
    """Closed loop of 16 robots for 5 steps: pmpc_mpc_step_batch_dev (x0 pinned on the device, warm start from the previous
    solution held in HBM, first control extracted on the device, Euler plant in torch on the same stream — no host
    synchronisation inside the loop). Every step is re-solved by the CPU restatement from the SAME inputs (state, warm-start
    primal / dual; mpc_wrapper.hpp:89-93 / :298 / sqp_base.hpp:368-374): identical SQP iteration counts and BIT-IDENTICAL solutions on
    every instance at every step (cold and warm-started). Warm-started steps
    need fewer iterations than the cold one (mpc_wrapper_test.cpp:159) and the loop drives the robots towards the origin."""
    import torch
    import polympc_amd as pa
    from polympc_amd import workloads
    B, K, dt = 16, 5, 0.05
    wl = workloads.robot_batch(B)
    n, m, nn = wl["n"], wl["m"], 7
    dev = torch.device("cuda", 0)
    ss = pa.sqp_settings_default(); ss.max_iter = 10; ss.line_search_max_iter = 10
    qs = pa.qp_settings_sqp_default()
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    state0 = wl["lbx"][:, 3 * nn - 3:3 * nn].copy()
    d_state = t(state0); d_d, d_lbx, d_ubx = t(wl["d"]), t(wl["lbx"]), t(wl["ubx"])
    d_x = torch.zeros(B, n, dtype=torch.float64, device=dev); d_lam = torch.zeros(B, m + n, dtype=torch.float64, device=dev)
    d_info = torch.zeros(B, 48, dtype=torch.uint8, device=dev); d_u0 = torch.zeros(B, 2, dtype=torch.float64, device=dev)

    def plant(s, u):   # explicit Euler on the unicycle (mpc_wrapper_test.cpp:33-44), wheel base d = 2
        return torch.stack([s[:, 0] + dt * u[:, 0] * torch.cos(s[:, 2]) * torch.cos(u[:, 1]),
                            s[:, 1] + dt * u[:, 0] * torch.sin(s[:, 2]) * torch.cos(u[:, 1]),
                            s[:, 2] + dt * u[:, 0] * torch.sin(u[:, 1]) / 2.0], 1).contiguous()

    torch.cuda.synchronize(dev)
    stream = torch.cuda.Stream(dev)   # a real (non-null) HIP stream shared by torch (plant, snapshots) and the solver context
    c2 = pa.Context(0, stream=stream.cuda_stream)
    snaps = []   # device-side snapshots (clones are stream-ordered): inputs and outputs of every step
    with torch.cuda.stream(stream):
        for k in range(K):
            before = (d_state.clone(), d_x.clone(), d_lam.clone())
            c2.mpc_step_batch_dev(0, 6, 1, 0.0, 2.0, B, d_state, d_d, d_lbx, d_ubx, d_x, d_lam, d_info, ss, qs, u0=d_u0)
            snaps.append(before + (d_x.clone(), d_info.clone(), d_u0.clone()))
            d_state = plant(d_state, d_u0)
    torch.cuda.synchronize(dev)
    c2.close()

    oss = oracle.sqp_default_settings(); oss.max_iter = 10; oss.line_search_max_iter = 10
    lbx, ubx = wl["lbx"].copy(), wl["ubx"].copy()
    mean_iters = []
    for k, (st_k, xg, lg, xk, info_k, u0k) in enumerate(snaps):
        st_k, xg, lg, xk, u0k = (a.cpu().numpy() for a in (st_k, xg, lg, xk, u0k))
        it_gpu = np.frombuffer(info_k.cpu().numpy().tobytes(), dtype=pa.capi.SQP_INFO_DTYPE)["iter"]
        lbx[:, 3 * nn - 3:3 * nn] = st_k; ubx[:, 3 * nn - 3:3 * nn] = st_k
        xo, lo, io = oracle.sqp_solve_batch(oracle.MODEL_ROBOT, 6, 1, 0.0, 2.0, B, wl["d"], lbx, ubx, x_guess=xg, lam_guess=lg,
                                            sqp_settings=oss, pivot=oracle.PIVOT_SWEEP)
        assert np.array_equal(it_gpu, np.array([i.iter for i in io])), (k, it_gpu, [i.iter for i in io])
        assert np.array_equal(xk, xo), (k, np.abs(xk - xo).max())
        assert np.array_equal(u0k, xk[:, 3 * nn + 2 * (nn - 1):3 * nn + 2 * nn])      # u(t_start) = last node of the u block
        assert np.abs(xk[:, 3 * nn - 3:3 * nn] - st_k).max() <= 1e-3                  # the pinned initial state is honoured (eps_prim)
        mean_iters.append(it_gpu.mean())
    assert max(mean_iters[1:]) < mean_iters[0]
    final = d_state.cpu().numpy()
    assert np.linalg.norm(final, axis=1).mean() < np.linalg.norm(state0, axis=1).mean()


# -------------------------------------------------------------------------------------------- A12-A15: fused SQP
def _sqp_both(ctx, oracle, wl, B, **kw):
    import polympc_amd as pa
    ss = pa.sqp_settings_default(); ss.max_iter = wl["max_iter"]; ss.line_search_max_iter = wl["ls_max_iter"]
    for k, v in kw.items():
        setattr(ss, k, v)
    for k, v in wl.get("settings", {}).items():   # (settings of the workload itself, e.g. the minimal-time problem's exact Hessians + Gershgorin shift)
        if k not in kw: setattr(ss, k, v)
    gk = {k: wl[k] for k in ("x_guess", "lbg", "ubg") if k in wl}
    x, lam, info = ctx.sqp_solve_batch(wl["model"], wl["P"], wl["S"], wl["t0"], wl["tf"], B, wl["d"], wl["lbx"], wl["ubx"], sqp_settings=ss, **gk)
    oss = oracle.sqp_default_settings(); oss.max_iter = wl["max_iter"]; oss.line_search_max_iter = wl["ls_max_iter"]
    for k, v in wl.get("settings", {}).items():
        if k not in kw: setattr(oss, k, v)
    for k, v in kw.items():
        if k != "kkt_form": setattr(oss, k, v)   # (how the large-instance kernel arranges its linear algebra: a pivot policy on the CPU side)
    kf = kw.get("kkt_form", 0)
    n = wl["lbx"].shape[1]; dm = oracle.ocp_dims(wl["model"], wl["P"], wl["S"])
    # (preconditioner = 1, qp_solver = 1 and line_search = 1 are served by the LDS-resident QP kernels whatever the size: static LDL^T
    #  order; hessian_update = 1 has register-resident specialisations like the default)
    # (regularisation = 1, eigenvalue mirroring: the hook builds of the register kernels since round 6 — the same routing as the other two hooks)
    if kw.get("qp_solver", 0): order = oracle.PIVOT_STATIC
    elif kw.get("preconditioner", 0) or kw.get("line_search", 0) or kw.get("regularisation", 0) == 1: order = _policy_order(oracle, dm["n"], dm["m"], wl["P"] * wl["S"] + 1, ruiz=bool(kw.get("preconditioner", 0)), block_bfgs=bool(kw.get("hessian_update", 0)), kkt_form=kf, ng=dm["ng"])
    else: order = _gpu_order(oracle, dm["n"], dm["m"], wl["P"] * wl["S"] + 1, block_bfgs=bool(kw.get("hessian_update", 0)), kkt_form=kf, ng=dm["ng"],
                             schur=_schur_route(wl["model"], wl["P"], wl["S"], **kw))
    xo, lo, io = oracle.sqp_solve_batch(wl["model"], wl["P"], wl["S"], wl["t0"], wl["tf"], B, wl["d"], wl["lbx"], wl["ubx"],
                                        sqp_settings=oss, pivot=order, threads=8, **gk)
    return (x, lam, info), (xo, lo, io)


def test_sqp_config_A_vs_oracle(ctx, oracle):
    """1024 config-A instances: identical SQP iteration counts, statuses and total ADMM iterations on EVERY instance, and bit-identical
    x, lam and reported KKT quantities (primal / dual step norms, constraint violation, cost)."""
    from polympc_amd import workloads
    B = 1024
    (x, lam, info), (xo, lo, io) = _sqp_both(ctx, oracle, workloads.robot_batch(B), B)
    _assert_same_solve(info, io, x, xo, lam, lo)


def test_sqp_config_D_perturbed_params(ctx, oracle):
    from polympc_amd import workloads
    B = 64
    (x, lam, info), (xo, lo, io) = _sqp_both(ctx, oracle, workloads.robot_batch(B, perturb_d=True, first=5000), B)
    _assert_same_solve(info, io, x, xo, lam, lo)


def test_sqp_reference_robot_fixture_sizes(ctx, oracle):
    """The reference's own test sizes: P=5,S=2 (88 KKT rows) and P=5,S=3 (128 KKT rows) — more rows than lanes."""
    from polympc_amd import workloads
    for P, S in ((5, 2), (5, 3)):
        B = 8
        (x, lam, info), (xo, lo, io) = _sqp_both(ctx, oracle, workloads.robot_batch(B, P=P, S=S), B)
        _assert_same_solve(info, io, x, xo, lam, lo)


def test_sqp_five_node_register_path(ctx, oracle):
    """5 collocation nodes (P=4,S=1 and P=2,S=2): 40 KKT rows, i.e. the register-resident QP with 24 idle lanes and a
    48-row padded accumulator grid."""
    from polympc_amd import workloads
    for P, S in ((4, 1), (2, 2)):
        B = 32
        (x, lam, info), (xo, lo, io) = _sqp_both(ctx, oracle, workloads.robot_batch(B, P=P, S=S), B)
        _assert_same_solve(info, io, x, xo, lam, lo)


def test_sqp_codegen_robot_exact_hessian(ctx, oracle):
    """codegen_test.cpp:402-438: exact Hessian every iteration, QP max_iter 1000 -> SOLVED in < 10 iterations."""
    import polympc_amd as pa
    n = 55
    lbx = np.full((1, n), -inf); ubx = np.full((1, n), inf)
    lbx[0, 30:33] = ubx[0, 30:33] = 0.5
    lbx[0, 33:] = np.tile([-1.5, -0.75], 11); ubx[0, 33:] = np.tile([1.5, 0.75], 11)
    ss = pa.sqp_settings_default(); ss.max_iter = 10; ss.line_search_max_iter = 10; ss.exact_hessian_every_iter = 1
    qs = pa.qp_settings_sqp_default(); qs.max_iter = 1000
    x, lam, info = ctx.sqp_solve_batch(pa.MODEL_ROBOT, 5, 2, 0.0, 1.0, 1, [[1.0]], lbx, ubx, sqp_settings=ss, qp_settings=qs)
    assert info["status"][0] == pa.SQP_SOLVED and info["iter"][0] < 10
    oss = oracle.sqp_default_settings(); oss.max_iter = 10; oss.line_search_max_iter = 10; oss.exact_hessian_every_iter = 1
    oqs = oracle.sqp_qp_default_settings(); oqs.max_iter = 1000
    xo, lo, io = oracle.sqp_solve_batch(oracle.MODEL_ROBOT, 5, 2, 0.0, 1.0, 1, [[1.0]], lbx, ubx, sqp_settings=oss, qp_settings=oqs, pivot=_gpu_order(oracle, 55, 33, 11, schur=True))   # exact Hessians: block diagonal -> the block-structured kernel
    _assert_same_solve(info, io, x, xo, lam, lo)
    assert ctx.last_route() == pa.capi.ROUTE_SCHUR


def test_sqp_minimal_time_valet_parking(ctx, oracle):
    """minimal_time_test.cpp:146-188 through the GPU path (NP = 1 border blocks, exact Hessian every iteration, Gershgorin shift,
    parameter / final-state bounds, primal guess): SOLVED in < 20 iterations like the reference asserts, same iteration count
    as the CPU restatement, bit-identical solution."""
    import polympc_amd as pa
    from test_oracle_pins import _minimal_time_parking
    lbx, ubx, xg = _minimal_time_parking()
    ss = pa.sqp_settings_default(); ss.max_iter = 20; ss.line_search_max_iter = 10; ss.regularisation = 2; ss.exact_hessian_every_iter = 1
    x, lam, info = ctx.sqp_solve_batch(pa.MODEL_PARKING, 5, 2, 0.0, 1.0, 1, [[1.0]], lbx, ubx, x_guess=xg, sqp_settings=ss)
    oss = oracle.sqp_default_settings(); oss.max_iter = 20; oss.line_search_max_iter = 10; oss.regularisation = 2; oss.exact_hessian_every_iter = 1
    assert ctx.last_route() == pa.capi.ROUTE_CONDREG   # round 6: the condensed register kernel with one parameter (the dense column of A as a wave reduction) — round 5: the full two-rows-per-lane inverse
    xo, lo, io = oracle.sqp_solve_batch(oracle.MODEL_PARKING, 5, 2, 0.0, 1.0, 1, [[1.0]], lbx, ubx, x_guess=xg, sqp_settings=oss, pivot=_gpu_order(oracle, 56, 33, 11))
    assert info["status"][0] == pa.SQP_SOLVED and info["iter"][0] < 20
    _assert_same_solve(info, io, x, xo, lam, lo)


@pytest.mark.parametrize("P,S,ubg,qp_max", [(5, 2, 10.0, 100), (5, 2, 10.0, 300), (5, 2, 1.2, 100), (6, 1, 10.0, 300), (6, 1, 1.2, 100)])
def test_sqp_parking_nonlinear_path_constraint(ctx, oracle, P, S, ubg, qp_max):
    """nonlinear_constraints_test.cpp:159-184 through the GPU path (NP = 1 and NG = 1 together, exact linearisation every iteration
    + Gershgorin): the reference's grid (P=5, S=2: 100 KKT rows, LDS path, static-order CPU restatement) with the reference's bound
    (inactive) and a binding one, and a 7-node grid whose KKT system has exactly 64 rows — the largest the register-resident path
    takes (CPU restatement in the kernel's sweep order). Same outcome, same SQP and QP iteration counts, bit-identical x and lam
    (the steering-angle profile of this minimal-time problem is nearly flat in the cost and most QPs stop at their iteration cap, so
    last-bit sin/cos differences are carried un-damped through up to 20 iterations; observed up to 1.8e-6)."""
    import polympc_amd as pa
    from test_oracle_pins import _minimal_time_parking
    nn = P * S + 1
    lbx, ubx, xg = _minimal_time_parking(nn)
    lbg = np.full((1, nn), -10.0); ubgv = np.full((1, nn), ubg)
    ss = pa.sqp_settings_default(); oss = oracle.sqp_default_settings()
    for st in (ss, oss):
        st.max_iter = 20; st.line_search_max_iter = 10; st.regularisation = 2; st.exact_hessian_every_iter = 1
    qs = pa.qp_settings_sqp_default(); qs.max_iter = qp_max
    oqs = oracle.sqp_qp_default_settings(); oqs.max_iter = qp_max
    x, lam, info = ctx.sqp_solve_batch(pa.MODEL_PARKING_NG, P, S, 0.0, 1.0, 1, [[1.0]], lbx, ubx, lbg=lbg, ubg=ubgv, x_guess=xg,
                                       sqp_settings=ss, qp_settings=qs)
    pivot = _gpu_order(oracle, 5 * nn + 1, 4 * nn, nn, ng=1)   # 7 nodes: 64 rows, one row per lane; 11 nodes: 100 rows — round 6: the condensed register kernel (56 variables, one per lane; the path-constraint rows as own-node blocks), before: two rows per lane
    assert ctx.last_route() == (pa.capi.ROUTE_CONDREG if nn == 11 else pa.capi.ROUTE_REG1)
    assert pivot == (oracle.PIVOT_CONDSWEEP if nn == 11 else oracle.PIVOT_SWEEP)
    xo, lo, io = oracle.sqp_solve_batch(oracle.MODEL_PARKING_NG, P, S, 0.0, 1.0, 1, [[1.0]], lbx, ubx, lbg=lbg, ubg=ubgv, x_guess=xg,
                                        sqp_settings=oss, qp_settings=oqs, pivot=pivot)
    _assert_same_solve(info, io, x, xo, lam, lo)
    if not (ubg == 10.0 and qp_max == 100):
        assert info["status"][0] == pa.SQP_SOLVED
    u = x[0, 3 * nn:5 * nn].reshape(nn, 2)
    assert (u[:, 0] ** 2 * np.cos(u[:, 1])).max() <= ubg + 1e-3


def test_sqp_valet_parking_with_ruiz(ctx, oracle):
    """valet_parking_mpc_test.cpp:183-240 through the GPU path (Ruiz preconditioner, QP max_iter 1000, cold + warm-started
    solve, l1 / dense-BFGS variant — see tests/test_oracle_pins.py): both solves SOLVED in < 10 iterations as the reference
    asserts, same iteration counts as the CPU restatement, x within 1e-7."""
    import polympc_amd as pa
    from test_oracle_pins import _valet_bounds
    ss = pa.sqp_settings_default(); ss.max_iter = 10; ss.line_search_max_iter = 10; ss.preconditioner = 1
    qs = pa.qp_settings_sqp_default(); qs.max_iter = 1000
    oss = oracle.sqp_default_settings(); oss.max_iter = 10; oss.line_search_max_iter = 10; oss.preconditioner = 1
    oqs = oracle.sqp_qp_default_settings(); oqs.max_iter = 1000
    xg = lg = xo = lo = None
    for x0 in ([0.5, 0.5, 0.5], [0.3, 0.4, 0.45]):
        lbx, ubx = _valet_bounds(x0)
        xg, lg, info = ctx.sqp_solve_batch(pa.MODEL_ROBOT, 5, 2, 0.0, 2.0, 1, [[2.0]], lbx, ubx, x_guess=xg, lam_guess=lg, sqp_settings=ss,
                                           qp_settings=qs, mparams=[1.0])
        xo, lo, io = oracle.sqp_solve_batch(oracle.MODEL_ROBOT, 5, 2, 0.0, 2.0, 1, [[2.0]], lbx, ubx, x_guess=xo, lam_guess=lo, sqp_settings=oss,
                                            qp_settings=oqs, pivot=_policy_order(oracle, 55, 33, 11, ruiz=True), mparams=[1.0])
        assert ctx.last_route() == pa.capi.ROUTE_CONDREG and _policy_order(oracle, 55, 33, 11, ruiz=True) == oracle.PIVOT_CONDSWEEP   # late round 6: the condensed hook kernel; before: the full two-rows-per-lane inverse
        assert info["status"][0] == pa.SQP_SOLVED and info["iter"][0] < 10
        _assert_same_solve(info, io, xg, xo, lg, lo)


@pytest.mark.parametrize("case", ["robot_16_nodes", "cstr_11_nodes", "parking_np1_11_nodes", "parking_np1_ng1_11_nodes", "robot_11_nodes_all_hooks"])
def test_sqp_ruiz_on_the_condensed_hook_kernels(ctx, oracle, case):
    """Late round 6: preconditioner = 1 (RuizEquilibration either side of the QP, sqp_base.hpp:605-611) on the hook builds of the condensed register kernel — the D~ tables per state
    index and the node blocks read back from the scaled workspace (pmpc_qp_cond.hpp WS). Grids of 65 .. 128 KKT rows of four models, incl. one parameter and path-constraint rows, and the
    three hooks of valet_parking_mpc_test.cpp together: route PMPC_ROUTE_CONDREG, bit-identical to PIVOT_CONDSWEEP on the scaled KKT matrix on every instance."""
    import polympc_amd as pa
    from polympc_amd import workloads
    B = 6
    kw = dict(preconditioner=1)
    if case == "robot_16_nodes":
        wl = workloads.robot_batch(B, P=5, S=3); wl["max_iter"] = 6
    elif case == "robot_11_nodes_all_hooks":
        wl = workloads.valet_parking_policy_batch(B); wl["max_iter"] = 8; kw = dict(wl.pop("settings")); wl.pop("qp_max_iter")
    elif case == "cstr_11_nodes":
        lbx, ubx = _cstr_grid(B, 5, 2)
        wl = dict(model=1, P=5, S=2, t0=0.0, tf=100.0, d=np.zeros((B, 1)), lbx=lbx, ubx=ubx, max_iter=6, ls_max_iter=20)
    else:
        wl = workloads.parking_reference_tests_batch(B, path_constraint=(case == "parking_np1_ng1_11_nodes")); wl["max_iter"] = 6
    (x, lam, info), (xo, lo, io) = _sqp_both(ctx, oracle, wl, B, **kw)
    assert ctx.last_route() == pa.capi.ROUTE_CONDREG
    dm = oracle.ocp_dims(wl["model"], wl["P"], wl["S"])
    assert _policy_order(oracle, dm["n"], dm["m"], wl["P"] * wl["S"] + 1, ruiz=True, ng=dm["ng"]) == oracle.PIVOT_CONDSWEEP
    _assert_same_solve(info, io, x, xo, lam, lo)


def test_sqp_valet_parking_as_the_reference_runs_it(ctx, oracle):
    """valet_parking_mpc_test.cpp:183-240 through the GPU path with every hook that test installs: Ruiz preconditioner, QP max_iter
    1000, the filter line search on LSFilter (beta = 0.1, carried from the cold solve into the warm-started one in a device buffer)
    and the block BFGS. Both solves SOLVED in < 10 iterations as the reference asserts; same SQP / QP iteration counts, the same
    filter contents (1e-9) and x within 1e-7 of the CPU restatement."""
    import polympc_amd as pa
    from test_oracle_pins import _valet_bounds
    ss = pa.sqp_settings_default(); oss = oracle.sqp_default_settings()
    for st in (ss, oss):
        st.max_iter = 10; st.line_search_max_iter = 10; st.preconditioner = 1; st.hessian_update = 1; st.line_search = 1; st.filter_beta = 0.1
    qs = pa.qp_settings_sqp_default(); qs.max_iter = 1000
    oqs = oracle.sqp_qp_default_settings(); oqs.max_iter = 1000
    handle = ctx.filter_state_create(1); ss.filter_state = handle
    ofilt = np.zeros((1, oracle.FILTER_STATE_DOUBLES)); oracle.bind_filter_state(oss, ofilt)
    try:
        xg = lg = xo = lo = None
        for x0 in ([0.5, 0.5, 0.5], [0.3, 0.4, 0.45]):
            lbx, ubx = _valet_bounds(x0)
            xg, lg, info = ctx.sqp_solve_batch(pa.MODEL_ROBOT, 5, 2, 0.0, 2.0, 1, [[2.0]], lbx, ubx, x_guess=xg, lam_guess=lg, sqp_settings=ss,
                                               qp_settings=qs, mparams=[1.0])
            xo, lo, io = oracle.sqp_solve_batch(oracle.MODEL_ROBOT, 5, 2, 0.0, 2.0, 1, [[2.0]], lbx, ubx, x_guess=xo, lam_guess=lo, sqp_settings=oss,
                                                qp_settings=oqs, pivot=_policy_order(oracle, 55, 33, 11, ruiz=True, block_bfgs=True), mparams=[1.0])
            assert ctx.last_route() == pa.capi.ROUTE_CONDREG   # late round 6: all three hooks of the reference's test on the condensed register kernel
            assert info["status"][0] == pa.SQP_SOLVED and info["iter"][0] < 10
            _assert_same_solve(info, io, xg, xo, lg, lo)
            filt = ctx.filter_state_download(1, handle)
            assert np.array_equal(filt, ofilt) and filt[0, 0] >= 1
        ctx.filter_state_clear(1, handle)
        assert not ctx.filter_state_download(1, handle).any()
    finally:
        ctx.filter_state_destroy(handle)


def test_sqp_filter_line_search_batch_vs_oracle(ctx, oracle):
    """line_search = 1 on batches of randomised robot OCPs (P=5 S=3: 128 KKT rows — since round 4 on the two-rows-per-lane register kernel with the hooks; P=5 S=4: 168 rows, HBM-factor kernel; config A's grid and the 11-node grid on the register-resident kernels that carry the hook since round 3), with and
    without a carried filter: identical iteration counts, bit-identical x, lam and filter contents; a second solve from the
    first one's solution with the carried filter must again agree (the filter then holds the first solve's history)."""
    import polympc_amd as pa
    from polympc_amd import workloads
    for P, S, B in ((5, 3, 6), (5, 4, 4), (6, 1, 32)):
        wl = workloads.robot_batch(B, P=P, S=S)
        ss = pa.sqp_settings_default(); oss = oracle.sqp_default_settings()
        for st in (ss, oss):
            st.max_iter = 10; st.line_search_max_iter = 10; st.line_search = 1
        handle = ctx.filter_state_create(B); ss.filter_state = handle
        ofilt = np.zeros((B, oracle.FILTER_STATE_DOUBLES)); oracle.bind_filter_state(oss, ofilt)
        dm = oracle.ocp_dims(oracle.MODEL_ROBOT, P, S)
        try:
            xg = lg = xo = lo = None
            for rep in range(2):
                xg, lg, info = ctx.sqp_solve_batch(pa.MODEL_ROBOT, P, S, 0.0, 2.0, B, wl["d"], wl["lbx"], wl["ubx"], x_guess=xg, lam_guess=lg, sqp_settings=ss)
                xo, lo, io = oracle.sqp_solve_batch(oracle.MODEL_ROBOT, P, S, 0.0, 2.0, B, wl["d"], wl["lbx"], wl["ubx"], x_guess=xo, lam_guess=lo,
                                                    sqp_settings=oss, pivot=_policy_order(oracle, dm["n"], dm["m"], P * S + 1))
                _assert_same_solve(info, io, xg, xo, lg, lo)
                filt = ctx.filter_state_download(B, handle)
                assert np.array_equal(filt, ofilt)
                xo, lo = xg.copy(), lg.copy()   # continue both sides from the same point
                ofilt[:] = filt
        finally:
            ctx.filter_state_destroy(handle)
    # without a carried state every call starts from an empty filter
    wl = workloads.robot_batch(8, P=5, S=2)
    ss = pa.sqp_settings_default(); ss.max_iter = 10; ss.line_search_max_iter = 10; ss.line_search = 1
    oss = oracle.sqp_default_settings(); oss.max_iter = 10; oss.line_search_max_iter = 10; oss.line_search = 1
    x, lam, info = ctx.sqp_solve_batch(pa.MODEL_ROBOT, 5, 2, 0.0, 2.0, 8, wl["d"], wl["lbx"], wl["ubx"], sqp_settings=ss)
    xo, lo, io = oracle.sqp_solve_batch(oracle.MODEL_ROBOT, 5, 2, 0.0, 2.0, 8, wl["d"], wl["lbx"], wl["ubx"], sqp_settings=oss, pivot=_policy_order(oracle, 55, 33, 11))
    _assert_same_solve(info, io, x, xo, lam, lo)


def test_sqp_filter_line_search_on_the_block_structured_kernel(ctx, oracle):
    """line_search = 1 together with the block BFGS (two of the three hooks valet_parking_mpc_test.cpp:116-165 installs; the third, the Ruiz preconditioner, rescales a
    dense workspace this kernel family does not have and keeps the dense kernels): since round 5 on the block-structured kernel for the 11- and 16-node grids — route
    PMPC_ROUTE_SCHUR, identical iteration counts, bit-identical x, lambda and filter contents against PIVOT_SCHUR; a second solve from the first one's solution with
    the carried filter agrees again."""
    import polympc_amd as pa
    from polympc_amd import workloads
    unflagged = 0
    for model, P, S, B in ((pa.MODEL_ROBOT, 5, 2, 24), (pa.MODEL_ROBOT, 5, 3, 12), (pa.MODEL_CSTR, 5, 2, 12)):
        wl = workloads.robot_batch(B, P=P, S=S) if model == pa.MODEL_ROBOT else workloads.cstr_batch(B)
        ss = pa.sqp_settings_default(); oss = oracle.sqp_default_settings()
        for st in (ss, oss):
            st.max_iter = 10; st.line_search_max_iter = 10; st.line_search = 1; st.hessian_update = 1
        handle = ctx.filter_state_create(B); ss.filter_state = handle
        ofilt = np.zeros((B, oracle.FILTER_STATE_DOUBLES)); oracle.bind_filter_state(oss, ofilt)
        try:
            xg = lg = xo = lo = None
            for rep in range(2):
                xg, lg, info = ctx.sqp_solve_batch(model, P, S, wl["t0"], wl["tf"], B, wl["d"], wl["lbx"], wl["ubx"], x_guess=xg, lam_guess=lg, sqp_settings=ss)
                assert ctx.last_route() == pa.capi.ROUTE_SCHUR
                xo, lo, io = oracle.sqp_solve_batch(model, P, S, wl["t0"], wl["tf"], B, wl["d"], wl["lbx"], wl["ubx"], x_guess=xo, lam_guess=lo, sqp_settings=oss, pivot=oracle.PIVOT_SCHUR)
                _assert_same_solve(info, io, xg, xo, lg, lo)
                filt = ctx.filter_state_download(B, handle)
                assert np.array_equal(filt, ofilt) and np.all(filt[:, 0] >= 1)
                fo = np.array([i.flags for i in io])
                print(model, P, S, "solve", rep, "flags", info["flags"].tolist())
                assert np.array_equal(info["flags"], fo) and not np.any(info["flags"] & pa.capi.FLAG_NONFINITE)   # (a conditioning-gate trip sends the instance — filter included — to the redo launch: same rule on both sides)
                unflagged += int(np.count_nonzero(info["flags"] == 0))
                xo, lo = xg.copy(), lg.copy()
                ofilt[:] = filt
        finally:
            ctx.filter_state_destroy(handle)
    assert unflagged >= 48   # (most solves run on the block-structured kernel from start to end)
    # a grid without the hook build keeps its previous route
    wl = workloads.cstr_batch(4); lbx, ubx = _cstr_grid(4, 6, 1)
    ss = pa.sqp_settings_default(); ss.max_iter = 2; ss.line_search = 1; ss.hessian_update = 1
    ctx.sqp_solve_batch(pa.MODEL_CSTR, 6, 1, 0.0, 100.0, 4, np.zeros((4, 1)), lbx, ubx, sqp_settings=ss)
    assert ctx.last_route() != pa.capi.ROUTE_SCHUR


def test_filter_settings_are_validated(ctx):
    import polympc_amd as pa
    from polympc_amd import workloads
    wl = workloads.robot_batch(2, P=5, S=2)
    for ls, depth in ((2, 10), (1, 0), (1, 11)):
        ss = pa.sqp_settings_default(); ss.line_search = ls; ss.filter_max_depth = depth
        with pytest.raises(RuntimeError):
            ctx.sqp_solve_batch(pa.MODEL_ROBOT, 5, 2, 0.0, 2.0, 2, wl["d"], wl["lbx"], wl["ubx"], sqp_settings=ss)


def test_sqp_block_bfgs_vs_oracle(ctx, oracle):
    """hessian_update = 1 (ContinuousOCP's sparsity-preserving block BFGS, continuous_ocp.hpp:2304-2431): identical iteration counts
    and x within 1e-8 of the CPU restatement on the reference's MPC-test grid (P=5, S=3), on config A's grid and, with a free
    parameter (NP = 1 border), on the parking model."""
    from polympc_amd import workloads
    import polympc_amd as pa
    for P, S, B in ((5, 3, 6), (6, 1, 256), (4, 1, 64)):   # LDS path; register-resident specialisations for 7 and 5 nodes (sweep order)
        (x, lam, info), (xo, lo, io) = _sqp_both(ctx, oracle, workloads.robot_batch(B, P=P, S=S), B, hessian_update=1)
        _assert_same_solve(info, io, x, xo, lam, lo)
    from test_oracle_pins import _minimal_time_parking
    lbx, ubx, xg = _minimal_time_parking()
    # three iterations only: quasi-Newton updates are not what this minimal-time problem is solved with (the reference switches it
    # to exact linearisation, minimal_time_test.cpp:120-133) and the iteration diverges later — identically on both sides
    ss = pa.sqp_settings_default(); ss.max_iter = 3; ss.line_search_max_iter = 10; ss.regularisation = 2; ss.hessian_update = 1
    x, lam, info = ctx.sqp_solve_batch(pa.MODEL_PARKING, 5, 2, 0.0, 1.0, 1, [[1.0]], lbx, ubx, x_guess=xg, sqp_settings=ss)
    oss = oracle.sqp_default_settings(); oss.max_iter = 3; oss.line_search_max_iter = 10; oss.regularisation = 2; oss.hessian_update = 1
    xo, lo, io = oracle.sqp_solve_batch(oracle.MODEL_PARKING, 5, 2, 0.0, 1.0, 1, [[1.0]], lbx, ubx, x_guess=xg, sqp_settings=oss, pivot=_gpu_order(oracle, 56, 33, 11))
    assert info["iter"][0] == io[0].iter == 3 and info["qp_solver_iter"][0] == io[0].qp_solver_iter
    assert np.array_equal(x, xo) and np.array_equal(lam, lo)


@pytest.mark.parametrize("model,P,S,B", [(0, 5, 2, 256), (0, 5, 3, 256), (1, 5, 2, 256), (1, 6, 1, 64)])
def test_sqp_block_structured_kernel_vs_oracle(ctx, oracle, model, P, S, B):
    """The block-structured kernel (pmpc_qp_schur.hpp; route PMPC_ROUTE_SCHUR): a Hessian that is block diagonal per node — the block BFGS every control
    test of the reference selects (hessian_update = 1, continuous_ocp.hpp:2304-2431) or exact Hessians every iteration — kept as per-node blocks in LDS,
    the QP through the m x m Schur complement with one refinement step. Config A's, config B's and the reference's own grids: identical trajectories and
    bit-identical x, lambda and KKT quantities against the restatement in this order (PIVOT_SCHUR) on every instance; also with the Gershgorin shift and
    with exact Hessians; kkt_form = 1 keeps the dense kernels."""
    import polympc_amd as pa
    from polympc_amd import workloads
    if model == 0:
        wl = workloads.robot_batch(B, P=P, S=S)
    elif (P, S) == (5, 2):
        wl = workloads.cstr_batch(B)
    else:
        lbx, ubx = _cstr_grid(B, P, S)
        wl = dict(model=1, P=P, S=S, t0=0.0, tf=100.0, d=np.zeros((B, 1)), lbx=lbx, ubx=ubx, max_iter=8, ls_max_iter=20)
    for kw in (dict(hessian_update=1), dict(hessian_update=1, regularisation=2), dict(exact_hessian_every_iter=1, regularisation=2)):
        (x, lam, info), (xo, lo, io) = _sqp_both(ctx, oracle, wl, B, **kw)
        assert ctx.last_route() == pa.capi.ROUTE_SCHUR, kw
        _assert_same_solve(info, io, x, xo, lam, lo)
        assert np.all(info["flags"] == 0)
    (x, lam, info), (xo, lo, io) = _sqp_both(ctx, oracle, wl, min(B, 32), hessian_update=1, kkt_form=1)
    assert ctx.last_route() != pa.capi.ROUTE_SCHUR
    _assert_same_solve(info, io, x, xo, lam, lo)


def test_sqp_admm_qp_solver_vs_oracle(ctx, oracle):
    """qp_solver = 1 (Solver<Problem, ADMM<...>>, the OSQP-form QP of admm.hpp inside the fused SQP kernel): identical SQP and ADMM
    iteration counts and x within 1e-8 of the CPU restatement, on config A's grid (91-row stacked KKT) and on P=5, S=2 (143 rows)."""
    from polympc_amd import workloads
    for P, S, B in ((6, 1, 24), (5, 2, 6)):
        (x, lam, info), (xo, lo, io) = _sqp_both(ctx, oracle, workloads.robot_batch(B, P=P, S=S), B, qp_solver=1)
        _assert_same_solve(info, io, x, xo, lam, lo)


def test_sqp_cstr_config_B(ctx, oracle):
    """Config B (CSTR, 110 KKT rows, exp-heavy dynamics, badly scaled): identical trajectories and bit-identical solutions on every instance."""
    from polympc_amd import workloads
    B = 128
    (x, lam, info), (xo, lo, io) = _sqp_both(ctx, oracle, workloads.cstr_batch(B), B)
    _assert_same_solve(info, io, x, xo, lam, lo)


@pytest.mark.parametrize("hessian_update", [1, 0])
def test_sqp_cstr_reference_scenario(ctx, oracle, hessian_update):
    """cstr_control_test.cpp:137-183 through the C ABI (cold solve, then the warm-started solve from a moved initial state), with the Hessian update the
    reference's test selects — the block BFGS of ContinuousOCP (:128-132, `hessian_update = 1`; 0 = the DENSE default, kept as a second case). Cold:
    SOLVED in 7 iterations, bit-identical to the restatement in the kernel's order. Warm, as the reference runs it (no regularisation of an indefinite
    exact Hessian — the outcome is a last-bit property, tests/test_oracle_pins.py::test_sqp_cstr_warm_solve_is_a_last_bit_property): whatever the
    restatement in the kernel's order does, the kernel does it bit for bit — iteration counts, status, x, lambda —, and the info word says whether a
    non-finite value went through. Warm with the Gershgorin shift: SOLVED in 4 iterations / 240 ADMM iterations — the counts of the Eigen-pivoted
    order — at the same optimum (1e-7 relative). Then the same solves with Eigen::LDLT's pivoting on the device (linear_solver = 1)."""
    import polympc_amd as pa
    from test_oracle_pins import _cstr_reference_scenario
    n = 66
    order = _gpu_order(oracle, 66, 44, 11, schur=bool(hessian_update))   # block BFGS: the block-structured kernel (PIVOT_SCHUR)
    for reg in (0, 2):
        ss = pa.sqp_settings_default(); ss.max_iter = 20; ss.line_search_max_iter = 20; ss.regularisation = reg; ss.hessian_update = hessian_update
        oss = oracle.sqp_default_settings(); oss.max_iter = 20; oss.line_search_max_iter = 20; oss.regularisation = reg; oss.hessian_update = hessian_update
        lbx = np.full((1, n), -inf); ubx = np.full((1, n), inf)
        lbx[0, 40:44] = ubx[0, 40:44] = [1.0, 0.5, 100.0, 100.0]
        lbx[0, 44:] = np.tile([3.0, -9000.0], 11); ubx[0, 44:] = np.tile([35.0, 0.0], 11)
        d = np.zeros((1, 1))
        x, lam, i1 = ctx.sqp_solve_batch(pa.MODEL_CSTR, 5, 2, 0.0, 100.0, 1, d, lbx, ubx, sqp_settings=ss)
        xo, lo, io1 = oracle.sqp_solve_batch(oracle.MODEL_CSTR, 5, 2, 0.0, 100.0, 1, d, lbx, ubx, sqp_settings=oss, pivot=order)
        _assert_same_solve(i1, io1, x, xo, lam, lo)
        assert i1["iter"][0] == 7 and i1["status"][0] == pa.SQP_SOLVED and i1["flags"][0] == 0
        assert i1["qp_solver_iter"][0] == (380 if hessian_update else 401)
        lbx[0, 40:44] = ubx[0, 40:44] = [1.1, 0.508, 100.5, 100.1]
        x2, lam2, i2 = ctx.sqp_solve_batch(pa.MODEL_CSTR, 5, 2, 0.0, 100.0, 1, d, lbx, ubx, x_guess=x, lam_guess=lam, sqp_settings=ss)
        xo2, lo2, io2 = oracle.sqp_solve_batch(oracle.MODEL_CSTR, 5, 2, 0.0, 100.0, 1, d, lbx, ubx, x_guess=xo, lam_guess=lo, sqp_settings=oss, pivot=order)
        # whatever the restatement in the kernel's order does, the kernel does it bit for bit — and the outcome itself is pinned, not merely mirrored:
        assert i2["iter"][0] == io2[0].iter and i2["qp_solver_iter"][0] == io2[0].qp_solver_iter and i2["status"][0] == io2[0].status
        assert np.array_equal(x2, xo2, equal_nan=True) and np.array_equal(lam2, lo2, equal_nan=True)
        assert bool(i2["flags"][0] & pa.capi.FLAG_NONFINITE) == (not (np.isfinite(x2).all() and np.isfinite(lam2).all()))
        if reg == 0 and hessian_update:
            # OPEN DISCREPANCY against cstr_control_test.cpp:177, stated as what it is instead of being accepted silently: the reference asserts
            # SOLVED for this warm solve; the block-structured order ends it finite at MAX_ITER_EXCEEDED (every QP at its cap) — and so does the restatement
            # with every linear solve carried to exact arithmetic (PIVOT_EXACT, either function set): the unregularised indefinite Hessian, not an
            # elimination order, decides it. A change of this outcome — in either direction — must be looked at.
            assert i2["status"][0] == pa.SQP_MAX_ITER_EXCEEDED and (i2["iter"][0], i2["qp_solver_iter"][0]) == (20, 2020)
            # (round 5: the block-structured kernel's conditioning gate trips on a QP of this indefinite stream — the instance is finished by the redo launch in the
            #  static order, which ends it the same way: flag PMPC_FLAG_ILLCOND, nothing non-finite)
            assert np.isfinite(x2).all() and np.isfinite(lam2).all() and i2["flags"][0] == io2[0].flags == pa.capi.FLAG_ILLCOND
        if reg == 0 and not hessian_update:
            # the dense-BFGS variant: SOLVED as :177 asserts — after an overflow (the termination test's norms drop NaNs), which the info word reports
            assert i2["status"][0] == pa.SQP_SOLVED and (i2["iter"][0], i2["qp_solver_iter"][0]) == (4, 313)
            assert i2["flags"][0] == pa.capi.FLAG_NONFINITE
        # the same two solves with Eigen::LDLT's pivoting on the device (linear_solver = 1): bit-identical to the Eigen-order restatement
        qp = pa.qp_settings_sqp_default(); qp.linear_solver = 1
        lbx[0, 40:44] = ubx[0, 40:44] = [1.0, 0.5, 100.0, 100.0]
        xp, lp, ip1 = ctx.sqp_solve_batch(pa.MODEL_CSTR, 5, 2, 0.0, 100.0, 1, d, lbx, ubx, sqp_settings=ss, qp_settings=qp)
        xe1, le1, ie1 = oracle.sqp_solve_batch(oracle.MODEL_CSTR, 5, 2, 0.0, 100.0, 1, d, lbx, ubx, sqp_settings=oss, pivot=oracle.PIVOT_EIGEN)
        _assert_same_solve(ip1, ie1, xp, xe1, lp, le1)
        lbx[0, 40:44] = ubx[0, 40:44] = [1.1, 0.508, 100.5, 100.1]
        xp2, lp2, ip2 = ctx.sqp_solve_batch(pa.MODEL_CSTR, 5, 2, 0.0, 100.0, 1, d, lbx, ubx, x_guess=xp, lam_guess=lp, sqp_settings=ss, qp_settings=qp)
        xe2, le2, ie2 = oracle.sqp_solve_batch(oracle.MODEL_CSTR, 5, 2, 0.0, 100.0, 1, d, lbx, ubx, x_guess=xe1, lam_guess=le1, sqp_settings=oss, pivot=oracle.PIVOT_EIGEN)
        assert ip2["iter"][0] == ie2[0].iter and ip2["qp_solver_iter"][0] == ie2[0].qp_solver_iter and ip2["status"][0] == ie2[0].status
        assert np.array_equal(xp2, xe2, equal_nan=True) and np.array_equal(lp2, le2, equal_nan=True)
        if reg == 2:
            (_, _), (xe, ie) = _cstr_reference_scenario(oracle, oracle.PIVOT_EIGEN, regularisation=2, hessian_update=hessian_update)
            assert (i2["iter"][0], i2["qp_solver_iter"][0]) == (ie.iter, ie.qp_solver_iter) == (4, 240) and i2["flags"][0] == 0
            assert i2["status"][0] == pa.SQP_SOLVED
            assert (np.abs(x2 - xe) / np.maximum(1.0, np.abs(xe))).max() <= 1e-7


def test_sqp_kite_standin_config_C(ctx, oracle):
    """Config C (synthetic 13-state / 3-input stand-in, 16 nodes, n=256, m=208: 464 KKT rows): the KKT factor does not
    fit LDS, so the large-instance mode keeps it in an HBM workspace. Same algorithm, same parity bar."""
    from polympc_amd import workloads
    B = 3
    (x, lam, info), (xo, lo, io) = _sqp_both(ctx, oracle, workloads.kite_standin_batch(B), B)
    _assert_same_solve(info, io, x, xo, lam, lo)   # bit for bit, like every other same-order comparison


def test_collocation_kite_standin_vs_oracle(ctx, oracle):
    import polympc_amd as pa
    rng = np.random.default_rng(13)
    dm = pa.ocp_dims(4, 5, 3)
    var = rng.uniform(-1, 1, (2, dm["n"])); lam = rng.uniform(-1, 1, (2, dm["m"] + dm["n"]))
    ev = ctx.ocp_linearise_batch(4, 5, 3, 0.0, 1.0, var, np.zeros((2, 1)), lam=lam)
    for b in range(2):
        eo = oracle.ocp_eval(4, 5, 3, 0.0, 1.0, var[b], [0.0], lam=lam[b])
        for k in ("c", "jac", "cost_grad", "lag_grad", "lag_hess"):
            ref = np.concatenate([eo["c"], eo["g"]]) if k == "c" else eo[k]
            assert np.abs(ev[k][b] - ref).max() <= 1e-12 * max(1.0, np.abs(ref).max()), k


def _cstr_grid(B, P, S):
    """cstr_batch's bounds on another grid."""
    from polympc_amd import workloads
    nn = P * S + 1
    n = 6 * nn
    inst = np.arange(B, dtype=np.uint64)
    lbx = np.full((B, n), -np.inf); ubx = np.full((B, n), np.inf)
    for j, b in enumerate([1.0, 0.5, 100.0, 100.0]):
        x0 = b * (1.0 + 0.05 * workloads.uniform_pm1(workloads.SEED, inst, j))
        lbx[:, 4 * nn - 4 + j] = x0; ubx[:, 4 * nn - 4 + j] = x0
    lbx[:, 4 * nn:] = np.tile([3.0, -9000.0], nn); ubx[:, 4 * nn:] = np.tile([35.0, 0.0], nn)
    return lbx, ubx


@pytest.mark.parametrize("case", ["cstr_13_nodes", "parking_16_nodes", "parking_ng_16_nodes", "robot_13_nodes_ruiz", "robot_21_nodes"])
def test_sqp_builtin_models_on_the_hbm_factor_kernel(ctx, oracle, case):
    """From 96 KKT rows on (and off the 11-node register grids) the fused SQP kernel keeps its factor in HBM: blocked tile LDL^T with MFMA
    trailing updates. Every built-in model family is driven through it — CSTR on 13 nodes (130 rows), the minimal-time parking problem with
    its free parameter on 16 nodes (129 rows; with the nonlinear path constraint 145 rows), the robot with the Ruiz preconditioner on 13 nodes
    (104 rows) and on 21 nodes (168 rows: not a multiple of 16) — against the blocked-order CPU restatement: bit-identical."""
    import polympc_amd as pa
    from polympc_amd import workloads
    from test_oracle_pins import _minimal_time_parking
    kw = dict(); okw = dict(); ss = pa.sqp_settings_default(); oss = oracle.sqp_default_settings(); B = 6
    if case == "cstr_13_nodes":
        model, P, S, t0, tf = pa.MODEL_CSTR, 4, 3, 0.0, 100.0
        lbx, ubx = _cstr_grid(B, P, S); d = np.zeros((B, 1))
        for st in (ss, oss): st.max_iter = 8; st.line_search_max_iter = 20
    elif case in ("parking_16_nodes", "parking_ng_16_nodes"):
        B = 1
        model, P, S, t0, tf = (pa.MODEL_PARKING if case == "parking_16_nodes" else pa.MODEL_PARKING_NG), 5, 3, 0.0, 1.0
        lbx, ubx, xg = _minimal_time_parking(16); d = np.array([[1.0]])
        kw["x_guess"] = xg; okw["x_guess"] = xg
        if case == "parking_ng_16_nodes":
            kw["lbg"] = okw["lbg"] = np.full((1, 16), -10.0); kw["ubg"] = okw["ubg"] = np.full((1, 16), 1.2)
        for st in (ss, oss): st.max_iter = 12; st.line_search_max_iter = 10; st.regularisation = 2; st.exact_hessian_every_iter = 1
    else:
        P, S = (4, 3) if case == "robot_13_nodes_ruiz" else (5, 4)
        wl = workloads.robot_batch(B, P=P, S=S)
        model, t0, tf, lbx, ubx, d = pa.MODEL_ROBOT, 0.0, 2.0, wl["lbx"], wl["ubx"], wl["d"]
        for st in (ss, oss):
            st.max_iter = 8; st.line_search_max_iter = 10
            if case == "robot_13_nodes_ruiz": st.preconditioner = 1
    dm = oracle.ocp_dims(model, P, S)
    assert dm["n"] + dm["m"] >= BIG_KKT_MIN_ROWS
    x, lam, info = ctx.sqp_solve_batch(model, P, S, t0, tf, B, d, lbx, ubx, sqp_settings=ss, **kw)
    xo, lo, io = oracle.sqp_solve_batch(model, P, S, t0, tf, B, d, lbx, ubx, sqp_settings=oss, pivot=_lds_order(oracle, dm["n"] + dm["m"], ruiz=(case == "robot_13_nodes_ruiz")), **okw)
    _assert_same_solve(info, io, x, xo, lam, lo)
    assert np.all(info["flags"] == 0)
    if case != "robot_13_nodes_ruiz":   # the same instances with the full KKT matrix (kkt_form = 1): the blocked order of round 2
        ss.kkt_form = 1
        x, lam, info = ctx.sqp_solve_batch(model, P, S, t0, tf, B, d, lbx, ubx, sqp_settings=ss, **kw)
        xo, lo, io = oracle.sqp_solve_batch(model, P, S, t0, tf, B, d, lbx, ubx, sqp_settings=oss, pivot=oracle.PIVOT_BLOCKED, **okw)
        _assert_same_solve(info, io, x, xo, lam, lo)


@pytest.mark.parametrize("P,S,B", [(6, 1, 64), (5, 2, 16), (5, 3, 6)])
def test_sqp_iteration_records_vs_oracle(ctx, oracle, P, S, B):
    """pmpc_sqp_settings::iteration_trace — what the reference hands to sqp_settings_t::iteration_callback (sqp_base.hpp:33, :685-686), recorded per
    iteration by the fused kernel: bit-identical to the CPU restatement's records on the three kernel families (register path, two rows per
    lane, HBM factor), one record per iteration that ran, nothing written beyond them, validation of the capacity."""
    import polympc_amd as pa
    from polympc_amd import workloads
    cap = 12
    wl = workloads.robot_batch(B, P=P, S=S)
    ss = pa.sqp_settings_default(); ss.max_iter = 10; ss.line_search_max_iter = 10
    oss = oracle.sqp_default_settings(); oss.max_iter = 10; oss.line_search_max_iter = 10
    h = ctx.iteration_trace_create(B, cap)
    try:
        ss.iteration_trace = h; ss.iteration_trace_capacity = cap
        x, lam, info = ctx.sqp_solve_batch(wl["model"], P, S, wl["t0"], wl["tf"], B, wl["d"], wl["lbx"], wl["ubx"], sqp_settings=ss)
        tr = ctx.iteration_trace_download(B, cap, h)
        otr = np.zeros((B, cap, oracle.TRACE_DOUBLES)); oracle.bind_iteration_trace(oss, otr)
        dm = oracle.ocp_dims(wl["model"], P, S)
        xo, lo, io = oracle.sqp_solve_batch(wl["model"], P, S, wl["t0"], wl["tf"], B, wl["d"], wl["lbx"], wl["ubx"], sqp_settings=oss,
                                            pivot=_gpu_order(oracle, dm["n"], dm["m"], P * S + 1, ng=dm["ng"]))
        _assert_same_solve(info, io, x, xo, lam, lo)
        assert np.array_equal(tr, otr)
        for b in range(B):
            assert np.array_equal(tr[b, :info["iter"][b], 0], np.arange(1, info["iter"][b] + 1)) and np.all(tr[b, info["iter"][b]:] == 0)
        ss.iteration_trace_capacity = 0
        with pytest.raises(RuntimeError):
            ctx.sqp_solve_batch(wl["model"], P, S, wl["t0"], wl["tf"], B, wl["d"], wl["lbx"], wl["ubx"], sqp_settings=ss)
    finally:
        ctx.iteration_trace_destroy(h)


@pytest.mark.parametrize("model,P,S", [(0, 3, 1), (0, 5, 1), (0, 7, 1), (0, 2, 1), (0, 4, 2), (0, 3, 3), (0, 11, 1), (0, 4, 3), (0, 13, 1), (1, 5, 1), (1, 4, 2), (1, 3, 1), (0, 7, 2), (0, 5, 3), (0, 3, 5), (1, 11, 1)])
def test_sqp_register_paths_on_other_grids(ctx, oracle, model, P, S):
    """Register-resident SQP kernels beyond the 5-, 7- and 11-node grids (pmpc_grids_*.hip): robot on 4, 6, 8 and 3 nodes (one KKT row per lane),
    on 9, 10, 12 and 13 nodes (72 .. 104 rows, two rows per lane) and on 15 and 16 nodes (120 / 128 rows: 8 x 8 tiles, sixteen of them in LDS — the
    reference's mpc_wrapper_test grid P = 5, S = 3 and the same node count as 3 x 5); CSTR on 6 nodes (60 rows), 9 nodes (90 rows), 4 nodes and 12 nodes (120 rows). Identical trajectories and
    bit-identical solutions against the sweep-order restatements; the block BFGS on such a grid takes the LDS-resident kernel (static order) below 65
    rows and stays on the two-rows-per-lane kernel above."""
    from polympc_amd import workloads
    B = 12
    nn = P * S + 1
    if model == 0:
        wl = workloads.robot_batch(B, P=P, S=S)
    else:
        lbx, ubx = _cstr_grid(B, P, S)
        wl = dict(model=1, P=P, S=S, t0=0.0, tf=100.0, d=np.zeros((B, 1)), lbx=lbx, ubx=ubx, max_iter=8, ls_max_iter=20)
    dm = oracle.ocp_dims(model, P, S)
    assert nn in REG_NODE_COUNTS and dm["n"] + dm["m"] <= 128
    (x, lam, info), (xo, lo, io) = _sqp_both(ctx, oracle, wl, B)
    _assert_same_solve(info, io, x, xo, lam, lo)
    (x, lam, info), (xo, lo, io) = _sqp_both(ctx, oracle, wl, B, hessian_update=1)
    _assert_same_solve(info, io, x, xo, lam, lo)


def test_sqp_warm_start_and_gershgorin(ctx, oracle):
    """Second solve warm-started from the first (x, lam) with a moved initial state; Gershgorin regulariser on."""
    import polympc_amd as pa
    from polympc_amd import workloads
    B = 8
    wl = workloads.robot_batch(B, P=5, S=3)
    ss = pa.sqp_settings_default(); ss.max_iter = 10; ss.line_search_max_iter = 10; ss.regularisation = 2
    mp = [2.0]
    x1, l1, i1 = ctx.sqp_solve_batch(0, 5, 3, 0.0, 2.0, B, wl["d"], wl["lbx"], wl["ubx"], sqp_settings=ss, mparams=mp)
    lbx2 = wl["lbx"].copy(); ubx2 = wl["ubx"].copy()
    lbx2[:, 45:48] -= 0.1; ubx2[:, 45:48] -= 0.1
    x2, l2, i2 = ctx.sqp_solve_batch(0, 5, 3, 0.0, 2.0, B, wl["d"], lbx2, ubx2, x_guess=x1, lam_guess=l1, sqp_settings=ss, mparams=mp)
    oss = oracle.sqp_default_settings(); oss.max_iter = 10; oss.line_search_max_iter = 10; oss.regularisation = 2
    xo1, lo1, io1 = oracle.sqp_solve_batch(0, 5, 3, 0.0, 2.0, B, wl["d"], wl["lbx"], wl["ubx"], sqp_settings=oss, pivot=_gpu_order(oracle, 80, 48, nodes=16), mparams=mp)
    xo2, lo2, io2 = oracle.sqp_solve_batch(0, 5, 3, 0.0, 2.0, B, wl["d"], lbx2, ubx2, x_guess=xo1, lam_guess=lo1, sqp_settings=oss, pivot=_gpu_order(oracle, 80, 48, nodes=16), mparams=mp)
    _assert_same_solve(i1, io1, x1, xo1, l1, lo1)
    _assert_same_solve(i2, io2, x2, xo2, l2, lo2)
    assert np.mean(i2["status"] == pa.SQP_SOLVED) >= 0.75


def test_sqp_eigenvalue_mirroring_regulariser(ctx, oracle):
    """regularisation = 1 (the eigenvalue-mirroring hook of sqp_test_autodiff.cpp:29-45) on the device: exact Lagrangian Hessian every iteration
    — indefinite along the way — mirrored by the in-LDS Jacobi iteration, against the CPU restatement's Jacobi: identical trajectories and
    bit-identical solutions. Round 6: on the grids of the reference's own tests the policy runs on the hook builds of the REGISTER kernels (7 nodes: one KKT row
    per lane, the constraint-first sweep; 11 nodes: the condensed register kernel), the Jacobi workspace in their LDS; elsewhere (5 nodes) on the LDS-resident kernel."""
    from polympc_amd import workloads
    import polympc_amd as pa
    for P, S, B, route, order in ((6, 1, 24, pa.capi.ROUTE_REG1, oracle.PIVOT_SWEEP), (4, 1, 16, pa.capi.ROUTE_LDS, oracle.PIVOT_STATIC), (5, 2, 12, pa.capi.ROUTE_CONDREG, oracle.PIVOT_CONDSWEEP)):
        wl = workloads.robot_batch(B, P=P, S=S)
        ss = pa.sqp_settings_default(); oss = oracle.sqp_default_settings()
        for st in (ss, oss):
            st.max_iter = 6; st.line_search_max_iter = 10; st.regularisation = 1; st.exact_hessian_every_iter = 1
        x, lam, info = ctx.sqp_solve_batch(wl["model"], P, S, 0.0, 2.0, B, wl["d"], wl["lbx"], wl["ubx"], sqp_settings=ss)
        assert ctx.last_route() == route, (P, S, pa.capi.ROUTE_NAMES.get(ctx.last_route()))
        xo, lo, io = oracle.sqp_solve_batch(wl["model"], P, S, 0.0, 2.0, B, wl["d"], wl["lbx"], wl["ubx"], sqp_settings=oss, pivot=order, threads=8)
        _assert_same_solve(info, io, x, xo, lam, lo)
        assert np.all(info["flags"] == 0)
    # the Jacobi workspace is 16 n^2 bytes of LDS: a grid that does not leave room for it is refused, not mis-solved
    wl = workloads.kite_standin_batch(1)
    ss = pa.sqp_settings_default(); ss.max_iter = 1; ss.regularisation = 1
    with pytest.raises(RuntimeError):
        ctx.sqp_solve_batch(wl["model"], wl["P"], wl["S"], wl["t0"], wl["tf"], 1, wl["d"], wl["lbx"], wl["ubx"], sqp_settings=ss)


def test_sqp_full_size_properties(ctx):
    """BASELINE size (4096 config-A OCPs): every SOLVED instance honours the pinned initial state, the control box and
    the collocation equalities to the solver's own tolerance (checked with an independent numpy evaluation)."""
    import polympc_amd as pa
    from polympc_amd import workloads
    B = 4096
    wl = workloads.robot_batch(B)
    ss = pa.sqp_settings_default(); ss.max_iter = 10; ss.line_search_max_iter = 10
    x, lam, info = ctx.sqp_solve_batch(0, 6, 1, 0.0, 2.0, B, wl["d"], wl["lbx"], wl["ubx"], sqp_settings=ss)
    ok = info["status"] == pa.SQP_SOLVED
    assert ok.mean() > 0.5
    nodes, w, D = pa.chebyshev(6)
    X = x[:, :21].reshape(B, 7, 3); U = x[:, 21:].reshape(B, 7, 2)
    f = np.stack([U[..., 0] * np.cos(X[..., 2]) * np.cos(U[..., 1]), U[..., 0] * np.sin(X[..., 2]) * np.cos(U[..., 1]),
                  U[..., 0] * np.sin(U[..., 1]) / 2.0], axis=-1)
    c = np.einsum("ij,bjs->bis", D, X) - 1.0 * f       # t_scale = (2-0)/(2*1)
    viol = np.abs(c).reshape(B, -1).max(axis=1)
    viol = np.maximum(viol, np.maximum((wl["lbx"] - x).max(axis=1), (x - wl["ubx"]).max(axis=1)))   # sqp_base.hpp:448-474
    assert np.abs(viol[ok] - info["max_violation"][ok]).max() <= 1e-9
    assert viol[ok].max() <= 1e-3
    assert np.abs(X[ok, 6, :] - wl["lbx"][ok, 18:21]).max() <= 1e-3
    assert np.all(np.abs(U[ok, :, 0]) <= 1.5 + 1e-3) and np.all(np.abs(U[ok, :, 1]) <= 0.75 + 1e-3)
    assert np.all(np.isfinite(x))


def test_sqp_full_size_properties_config_D(ctx):
    """BASELINE size of config D (8192 robots per GPU, perturbed wheel base): the same size-independent checks as config A with each
    instance's own parameter in the independent numpy evaluation of the dynamics."""
    import polympc_amd as pa
    from polympc_amd import workloads
    B = 8192
    wl = workloads.robot_batch(B, perturb_d=True, first=5000)
    ss = pa.sqp_settings_default(); ss.max_iter = 10; ss.line_search_max_iter = 10
    x, lam, info = ctx.sqp_solve_batch(0, 6, 1, 0.0, 2.0, B, wl["d"], wl["lbx"], wl["ubx"], sqp_settings=ss)
    ok = info["status"] == pa.SQP_SOLVED
    assert ok.mean() > 0.5 and np.all(info["flags"] == 0) and np.all(np.isfinite(x))
    nodes, w, D = pa.chebyshev(6)
    X = x[:, :21].reshape(B, 7, 3); U = x[:, 21:].reshape(B, 7, 2)
    f = np.stack([U[..., 0] * np.cos(X[..., 2]) * np.cos(U[..., 1]), U[..., 0] * np.sin(X[..., 2]) * np.cos(U[..., 1]),
                  U[..., 0] * np.sin(U[..., 1]) / wl["d"]], axis=-1)
    viol = np.abs(np.einsum("ij,bjs->bis", D, X) - f).reshape(B, -1).max(axis=1)
    viol = np.maximum(viol, np.maximum((wl["lbx"] - x).max(axis=1), (x - wl["ubx"]).max(axis=1)))
    assert np.abs(viol[ok] - info["max_violation"][ok]).max() <= 1e-9 and viol[ok].max() <= 1e-3
    assert np.all(np.abs(U[ok, :, 0]) <= 1.5 + 1e-3) and np.all(np.abs(U[ok, :, 1]) <= 0.75 + 1e-3)


def test_sqp_full_size_properties_config_B(ctx):
    """BASELINE size of config B (16 384 CSTR instances, 110 KKT rows, two-rows-per-lane register path): every instance SOLVED, finite, inside
    the input box, initial state pinned, and the reported constraint violation equal to an independent numpy evaluation of the collocation
    defects (cstr_control_test.cpp:80-96 dynamics, t_scale = 100 / (2 * 2) = 25)."""
    import polympc_amd as pa
    from polympc_amd import workloads
    B = 16384
    wl = workloads.cstr_batch(B)
    ss = pa.sqp_settings_default(); ss.max_iter = wl["max_iter"]; ss.line_search_max_iter = wl["ls_max_iter"]
    x, lam, info = ctx.sqp_solve_batch(wl["model"], 5, 2, 0.0, 100.0, B, wl["d"], wl["lbx"], wl["ubx"], sqp_settings=ss)
    ok = info["status"] == pa.SQP_SOLVED
    assert ok.mean() >= 0.999 and np.all(info["flags"] == 0) and np.all(np.isfinite(x))
    nn = 11
    X = x[:, :4 * nn].reshape(B, nn, 4); U = x[:, 4 * nn:].reshape(B, nn, 2)
    k1 = 1.287e12 * np.exp(-9758.3 / (273.15 + X[..., 2])); k2 = k1; k3 = 9.043e09 * np.exp(-8560.0 / (273.15 + X[..., 2]))
    f = np.stack([(1 / 3600.0) * (U[..., 0] * (5.1 - X[..., 0]) - k1 * X[..., 0] - k3 * X[..., 0] ** 2),
                  (1 / 3600.0) * (-U[..., 0] * X[..., 1] + k1 * X[..., 0] - k2 * X[..., 1]),
                  (1 / 3600.0) * (U[..., 0] * (104.9 - X[..., 2]) + (4032.0 * 0.215 / (0.9342 * 3.01 * 10.0)) * (X[..., 3] - X[..., 2])
                                  - (1 / (0.9342 * 3.01)) * (k1 * X[..., 0] * 4.2 + k2 * X[..., 1] * (-11.0) + k3 * X[..., 0] * X[..., 1] * (-41.85))),
                  (1 / 3600.0) * ((1 / (5.0 * 2.0)) * (U[..., 1] + 4032.0 * 0.215 * (X[..., 2] - X[..., 3])))], axis=-1)
    nodes, w, D = pa.chebyshev(5)
    c = np.zeros((B, nn, 4))
    for seg in range(2):   # later segments overwrite the junction row (continuous_ocp.hpp:750-751)
        c[:, seg * 5:seg * 5 + 6] = np.einsum("ij,bjs->bis", D, X[:, seg * 5:seg * 5 + 6]) - 25.0 * f[:, seg * 5:seg * 5 + 6]
    viol = np.abs(c).reshape(B, -1).max(axis=1)
    viol = np.maximum(viol, np.maximum((wl["lbx"] - x).max(axis=1), (x - wl["ubx"]).max(axis=1)))
    assert np.abs(viol[ok] - info["max_violation"][ok]).max() <= 1e-6 * max(1.0, np.abs(viol[ok]).max()) + 1e-9
    assert viol[ok].max() <= 1e-3
    assert np.abs(X[ok, nn - 1, :] - wl["lbx"][ok, 4 * nn - 4:4 * nn]).max() <= 1e-3
    assert np.all(U[ok, :, 0] >= 3.0 - 1e-3) and np.all(U[ok, :, 0] <= 35.0 + 1e-3) and np.all(U[ok, :, 1] >= -9000.0 - 1e-3) and np.all(U[ok, :, 1] <= 1e-3)


def _same_bits(a, b):
    """Bit-for-bit equality (NaN payloads included)."""
    a = np.ascontiguousarray(a); b = np.ascontiguousarray(b)
    return a.shape == b.shape and a.tobytes() == b.tobytes()


@pytest.mark.gpu
@pytest.mark.parametrize("hessian_update,warm", [(0, False), (1, False), (0, True)])
def test_sqp_round_robin_execution_bit_identical(ctx, oracle, monkeypatch, hessian_update, warm):
    """PMPC_SQP_RR=1 (developer switch): batches beyond the resident wavefronts run one SQP ITERATION per work item from a ready queue, with the
    instance's state in HBM between iterations (sqp_kernel_rr, pmpc_launch.hpp). 4099 config-A instances (not a multiple of the eight queues): the
    same iterations, statuses, ADMM iteration totals, bit-identical x / lam / KKT quantities and iteration records as the one-launch kernel
    (default context) — cold and warm-started, dense and block BFGS — and as the CPU restatement on every instance."""
    import polympc_amd as pa
    from polympc_amd import workloads
    B, cap = 4099, 10
    wl = workloads.robot_batch(B)
    ss = pa.sqp_settings_default(); ss.max_iter = 10; ss.line_search_max_iter = 10; ss.hessian_update = hessian_update
    xg = lg = None
    if warm:   # warm start = the iterate of a two-iteration solve
        s2 = pa.sqp_settings_default(); s2.max_iter = 2; s2.line_search_max_iter = 10
        xg, lg, _ = ctx.sqp_solve_batch(wl["model"], wl["P"], wl["S"], wl["t0"], wl["tf"], B, wl["d"], wl["lbx"], wl["ubx"], sqp_settings=s2)
    monkeypatch.setenv("PMPC_SQP_RR", "1")
    rr_ctx = pa.Context(0)
    monkeypatch.delenv("PMPC_SQP_RR")
    res = []
    try:
        for c in (rr_ctx, ctx):
            h = c.iteration_trace_create(B, cap)
            try:
                ss.iteration_trace = h; ss.iteration_trace_capacity = cap
                x, lam, info = c.sqp_solve_batch(wl["model"], wl["P"], wl["S"], wl["t0"], wl["tf"], B, wl["d"], wl["lbx"], wl["ubx"],
                                                 x_guess=xg, lam_guess=lg, sqp_settings=ss)
                res.append((x, lam, info, c.iteration_trace_download(B, cap, h)))
            finally:
                c.iteration_trace_destroy(h)
    finally:
        rr_ctx.close()
    (x, lam, info, tr), (x1, lam1, info1, tr1) = res
    assert info["iter"].min() < info["iter"].max()   # the instances need different numbers of iterations
    assert _same_bits(info, info1) and _same_bits(x, x1) and _same_bits(lam, lam1) and _same_bits(tr, tr1)
    oss = oracle.sqp_default_settings(); oss.max_iter = 10; oss.line_search_max_iter = 10; oss.hessian_update = hessian_update
    xo, lo, io = oracle.sqp_solve_batch(wl["model"], wl["P"], wl["S"], wl["t0"], wl["tf"], B, wl["d"], wl["lbx"], wl["ubx"], x_guess=xg, lam_guess=lg,
                                        sqp_settings=oss, pivot=_gpu_order(oracle, 35, 21, 7, block_bfgs=bool(hessian_update)), threads=8)
    fin = np.isfinite(x).all(axis=1) & np.isfinite(xo).all(axis=1)   # (a warm start from an unconverged iterate can diverge: identically on both sides)
    assert np.array_equal(np.isfinite(x).all(axis=1), np.isfinite(xo).all(axis=1)) and fin.mean() > 0.99
    assert np.array_equal((info["flags"] & pa.capi.FLAG_NONFINITE) != 0, ~(np.isfinite(x).all(axis=1) & np.isfinite(lam).all(axis=1)))   # PMPC_FLAG_NONFINITE marks exactly the non-finite results
    assert np.array_equal(info["flags"], np.array([i.flags for i in io]) | (info["flags"] & pa.capi.FLAG_NONFINITE))   # (PMPC_FLAG_ILLCOND — a diverging warm start can trip the conditioning gate — exactly where the restatement sets it)
    _assert_same_solve(info[fin], [i for i, f in zip(io, fin) if f], x[fin], xo[fin], lam[fin], lo[fin])
    assert np.array_equal(info["iter"], np.array([i.iter for i in io])) and np.array_equal(info["status"], np.array([i.status for i in io]))


# ------------------------------------------------------------------------------ the DEFAULT kernels against the REFERENCE order, at full size
REFERENCE_ORDER_BOUNDS = {
    # measured on the CPU restatement in the kernel's order (oracle/cross_order.py; the GPU reproduces that run bit for bit) and asserted with
    # margin: (instances, max different trajectories, max |dx| over identical trajectories, scaled dx p99, max d violation, max rel d cost)
    "A": (4096, 0, 2e-8, 1e-10, 1e-10, 1e-10),    # 1.72e-8 (one instance above 1e-8), p99 2.1e-11, 2.4e-11, 1.9e-11
    "D": (8192, 0, 1e-7, 1e-10, 1e-9, 1e-9),      # 7.4e-8 (two instances above 1e-8), p99 1.8e-11, 1.3e-10, 5.1e-11
    "B": (2048, 0, 1e-2, 1e-8, 1e-7, 1e-8),       # 2.1e-3 on a control of magnitude 9000 = 3.1e-6 scaled; p99 2.2e-9 scaled; 2.2e-8; 4.9e-9
    "R": (2048, 0, 1e-8, 1e-9, 1e-10, 1e-10),     # 1.5e-9, p99 8.8e-11, 2.1e-12, 4.3e-12
    "C": (64, 0, 1e-10, 1e-10, 1e-10, 1e-10),     # 8.4e-12, 5.1e-12, 6.5e-14, 8.1e-13
}


@pytest.mark.parametrize("cfg", ["A", "D", "B", "R", "C"])
def test_default_kernels_against_the_reference_order(ctx, oracle, cfg):
    """north_star: "outputs match the reference Eigen CPU SQP on identical problem data to a stated fp64 tolerance". The DEFAULT kernels (whatever
    path pmpc_launch.hpp routes the size to) against the restatement AS THE REFERENCE COMPUTES — Eigen-style pivoted LDL^T (PIVOT_EIGEN) and glibc's
    sin / cos / exp — on every BASELINE configuration: A and D at full size, B on 2048 of its 16 384 instances (the full stream: CPU test
    test_config_B_cross_order_spread_is_the_conditioning_of_the_trajectory — 2 of 16 384 instances take another branch), the reference's 16-node
    robot grid at 2048, C on 64 of its 1024 (57 QP/s on the CPU). Asserted on EVERY instance, no mask: identical SQP iterations, status and total
    ADMM iterations; constraint violation and relative cost within 1e-8 (B: 1e-7 on the violation — measured 2.2e-8); and the stated tolerance on
    the solution itself, which is north_star's 1e-8 for R and C, 2e-8 / 1e-7 for the worst instance of A / D (p99 below 1e-10), and for B the
    percentiles of the difference scaled by each variable's magnitude (p99 <= 1e-8; worst instance 3e-6, i.e. 2e-3 on a control bounded by 9000).
    The record (percentiles included) is printed and is the same object bench.py emits per configuration."""
    import polympc_amd as pa
    from oracle import cross_order as tco
    nB, max_diff, tol_dx, tol_p99, tol_viol, tol_cost = REFERENCE_ORDER_BOUNDS[cfg]
    wl, _ = tco.config_workload(cfg, B=nB)
    ss = pa.sqp_settings_default(); ss.max_iter = wl["max_iter"]; ss.line_search_max_iter = wl["ls_max_iter"]
    x, lam, info = ctx.sqp_solve_batch(wl["model"], wl["P"], wl["S"], wl["t0"], wl["tf"], nB, wl["d"], wl["lbx"], wl["ubx"], sqp_settings=ss)
    xr, lr, ir = tco.oracle_run(oracle, wl, nB, oracle.PIVOT_EIGEN, True, 8)
    r = tco.cross_order_stats(cfg, wl, x, lam, info, xr, lr, ir)
    print(cfg, "route", pa.capi.ROUTE_NAMES[ctx.last_route()], r)
    assert ctx.last_route() == {"A": pa.capi.ROUTE_REG1, "D": pa.capi.ROUTE_REG1, "B": pa.capi.ROUTE_CONDREG, "R": ROUTE_OF_128_ROWS(pa), "C": pa.capi.ROUTE_HBM}[cfg]
    assert r["instances"] == nB and r["different_trajectories"] <= max_diff
    assert r["identical_trajectories_only"]["max_abs_dx"] <= tol_dx
    assert r["scaled_dx_per_instance"]["p99"] <= tol_p99
    assert r["max_abs_d_constraint_violation"] <= tol_viol and r["max_rel_d_cost"] <= tol_cost
    assert np.all(info["flags"] == 0)


@pytest.mark.parametrize("hessian_update", [0, 1])
def test_cstr_short_horizon_tight_pin_against_the_reference_order(ctx, oracle, hessian_update):
    """A TIGHT pin beside the percentile test above (config B's 20-iteration trajectories amplify last-bit differences to 3e-6 scaled on the worst
    instance): the same 1024 CSTR instances stopped after 5 SQP iterations, where nothing has been amplified yet — default kernel (two rows per lane,
    dense BFGS) and block-structured kernel (block BFGS) against the restatement as the reference computes (PIVOT_EIGEN, glibc): identical trajectories
    on every instance, every variable within 1e-9 of its magnitude (measured 8.8e-11 / 7.5e-13), multipliers within 1e-9 scaled, violation within 1e-10.
    A regression of either kernel's accuracy below ~1e-9 fails here."""
    import polympc_amd as pa
    from oracle import cross_order as tco
    nB = 1024
    wl, _ = tco.config_workload("B", B=nB); wl = dict(wl); wl["max_iter"] = 5
    ss = pa.sqp_settings_default(); ss.max_iter = 5; ss.line_search_max_iter = wl["ls_max_iter"]; ss.hessian_update = hessian_update
    x, lam, info = ctx.sqp_solve_batch(wl["model"], wl["P"], wl["S"], wl["t0"], wl["tf"], nB, wl["d"], wl["lbx"], wl["ubx"], sqp_settings=ss)
    assert ctx.last_route() == (pa.capi.ROUTE_SCHUR if hessian_update else pa.capi.ROUTE_CONDREG)
    xr, lr, ir = tco.oracle_run(oracle, wl, nB, oracle.PIVOT_EIGEN, True, 8, hessian_update=hessian_update)
    r = tco.cross_order_stats("B", wl, x, lam, info, xr, lr, ir)
    print(hessian_update, r)
    assert r["different_trajectories"] == 0
    assert r["scaled_dx_per_instance"]["max"] <= 1e-9 and r["scaled_dlam_per_instance"]["max"] <= 1e-9 and r["max_abs_d_constraint_violation"] <= 1e-10


@pytest.mark.parametrize("alpha", [1.0, 1.6])
def test_dual_residual_identity_and_its_fallback(ctx, oracle, alpha):
    """Round 6: the condensed kernels take H x of boxADMM's dual residual from the KKT identity of the last solve when alpha == 1 (x = x~, quirk Q1) and multiply by
    H, as the reference does (qp_base.hpp:240-252), for any other relaxation parameter. Both branches against the restatement that states the same rule — config B and
    the 16-node robot grid on the condensed register kernel (PIVOT_CONDSWEEP), two kite-sized instances on the large-instance kernel (PIVOT_CONDENSED) — bit for bit;
    with alpha = 1 the residuals the QPs report must also agree with the reference order's to 1e-8 although they are formed differently."""
    import polympc_amd as pa
    from polympc_amd import workloads
    from oracle import cross_order as tco
    cases = [(tco.config_workload("B", B=48)[0], 48, pa.capi.ROUTE_CONDREG, oracle.PIVOT_CONDSWEEP, 3),
             (workloads.robot_batch(32, P=5, S=3), 32, pa.capi.ROUTE_CONDREG, oracle.PIVOT_CONDSWEEP, 3),
             (workloads.kite_standin_batch(2), 2, pa.capi.ROUTE_HBM, oracle.PIVOT_CONDENSED, 2)]
    for wl, B, route, order, iters in cases:
        ss = pa.sqp_settings_default(); oss = oracle.sqp_default_settings()
        for st in (ss, oss):
            st.max_iter = iters; st.line_search_max_iter = wl["ls_max_iter"]
        qs = pa.qp_settings_sqp_default(); oqs = oracle.sqp_qp_default_settings()
        qs.alpha = alpha; oqs.alpha = alpha
        x, lam, info = ctx.sqp_solve_batch(wl["model"], wl["P"], wl["S"], wl["t0"], wl["tf"], B, wl["d"], wl["lbx"], wl["ubx"], sqp_settings=ss, qp_settings=qs)
        assert ctx.last_route() == route
        xo, lo, io = oracle.sqp_solve_batch(wl["model"], wl["P"], wl["S"], wl["t0"], wl["tf"], B, wl["d"], wl["lbx"], wl["ubx"], sqp_settings=oss, qp_settings=oqs,
                                            pivot=order, threads=8)
        _assert_same_solve(info, io, x, xo, lam, lo)
        if alpha == 1.0:   # the identity changes no count against the reference order either
            with oracle.libm():
                xr, lr, ir = oracle.sqp_solve_batch(wl["model"], wl["P"], wl["S"], wl["t0"], wl["tf"], B, wl["d"], wl["lbx"], wl["ubx"], sqp_settings=oss, qp_settings=oqs,
                                                    pivot=oracle.PIVOT_EIGEN, threads=8)
            assert [(int(a), int(b)) for a, b in zip(info["iter"], info["qp_solver_iter"])] == [(i.iter, i.qp_solver_iter) for i in ir]


@pytest.mark.parametrize("rho0", [10.0, 1e3, 1e5])
def test_condensed_register_kernel_under_a_large_penalty(ctx, oracle, rho0):
    """The condensed register kernel with the QP's penalty started at rho = 10 / 1e3 / 1e5 (rho_eq up to 1e8; RHO_MAX = 1e6, box_admm.hpp:56-59): the kernel
    equals its restatement bit for bit, the conditioning gate stays silent (the directions A leaves free are the bounded controls: cond(S) ~ 1e5 whatever
    rho is), every instance keeps the SQP and ADMM iteration counts of the run as the reference computes (pivoted LDL^T of the KKT matrix, glibc) — and
    where the two runs differ it is the REFERENCE order that left exact arithmetic: against PIVOT_EXACT (every linear solve refined in long double) the
    kernel's iterates are the closer ones (round 5 finding; tests/test_oracle_pins.py::test_kernel_orders_follow_exact_arithmetic_more_closely_than_the_reference_order
    pins it per QP). At rho0 = 1e5 neither run reproduces the exact one to 1e-8 — the computation itself is that ill-conditioned."""
    import polympc_amd as pa
    from oracle import cross_order as tco
    nB = 64
    wl, _ = tco.config_workload("B", B=nB)
    ss = pa.sqp_settings_default(); oss = oracle.sqp_default_settings()
    for st in (ss, oss):
        st.max_iter = 4; st.line_search_max_iter = wl["ls_max_iter"]
    qs = pa.qp_settings_sqp_default(); oqs = oracle.sqp_qp_default_settings()
    qs.rho = rho0; oqs.rho = rho0
    x, lam, info = ctx.sqp_solve_batch(wl["model"], wl["P"], wl["S"], wl["t0"], wl["tf"], nB, wl["d"], wl["lbx"], wl["ubx"], sqp_settings=ss, qp_settings=qs)
    assert ctx.last_route() == pa.capi.ROUTE_CONDREG
    xo, lo, io = oracle.sqp_solve_batch(wl["model"], wl["P"], wl["S"], wl["t0"], wl["tf"], nB, wl["d"], wl["lbx"], wl["ubx"], sqp_settings=oss, qp_settings=oqs,
                                        pivot=oracle.PIVOT_CONDSWEEP, threads=8)
    _assert_same_solve(info, io, x, xo, lam, lo)
    with oracle.libm():
        xr, lr, ir = oracle.sqp_solve_batch(wl["model"], wl["P"], wl["S"], wl["t0"], wl["tf"], nB, wl["d"], wl["lbx"], wl["ubx"], sqp_settings=oss, qp_settings=oqs,
                                            pivot=oracle.PIVOT_EIGEN, threads=8)
        xe, le, ie = oracle.sqp_solve_batch(wl["model"], wl["P"], wl["S"], wl["t0"], wl["tf"], nB, wl["d"], wl["lbx"], wl["ubx"], sqp_settings=oss, qp_settings=oqs,
                                            pivot=oracle.PIVOT_EXACT, threads=8)
    assert np.all(info["flags"] == 0)
    r = tco.cross_order_stats("B", wl, x, lam, info, xr, lr, ir)
    rk = tco.cross_order_stats("B", wl, x, lam, info, xe, le, ie)
    rr = tco.cross_order_stats("B", wl, xr, lr, ir, xe, le, ie)
    print(rho0, "vs reference order", r["different_trajectories"], r["scaled_dx_per_instance"]["max"], "| kernel vs exact", rk["scaled_dx_per_instance"]["max"],
          "reference order vs exact", rr["scaled_dx_per_instance"]["max"])
    assert r["different_trajectories"] == 0 and rk["different_trajectories"] == 0
    assert rk["scaled_dx_per_instance"]["max"] <= max(2 * rr["scaled_dx_per_instance"]["max"], 1e-9)
    if rho0 <= 1e3: assert r["scaled_dx_per_instance"]["max"] <= 1e-6


def _free_controls(wl, nx):
    """the workload with every bound removed except the pinned initial state: the directions the collocation Jacobian leaves free then carry
    rho_box = RHO_MIN instead of rho — the one situation in which the condensed form loses digits as rho grows (PMPC_FLAG_ILLCOND)"""
    nn = wl["P"] * wl["S"] + 1
    w = dict(wl); w["lbx"] = wl["lbx"].copy(); w["ubx"] = wl["ubx"].copy()
    w["lbx"][:, nx * nn:] = -np.inf; w["ubx"][:, nx * nn:] = np.inf
    return w


def test_block_structured_kernel_conditioning_gate_and_its_redo_launch(ctx, oracle):
    """The block-structured kernel's gate (round 5): its range-space solve x = Q (r1 - A' nu) works through S = 1/rho + A Q A' and is a difference of quantities
    ~ rho_eq times larger than x — the error of a solve grows like the conditioning estimate squared once the ADMM penalty adapts upwards (2e-9 at an estimate of 4e6,
    2e-5 at 4e7, nothing at 4e10). From max S_ii max |(S^-1)_ii| = 1e7 on the QP is given up and the redo launch solves the instance on the LDS-resident static LDL^T.
    Robot 11 / 16 nodes and CSTR with the block BFGS and the QP penalty started at 0.1 (no trip: flags clear — as on every BASELINE workload — and within 1e-8 of
    Eigen's pivoted order), 30 and 1e3: bit for bit the restatement under the same rule (PIVOT_SCHUR -> PIVOT_STATIC), flag included."""
    import polympc_amd as pa
    from polympc_amd import workloads
    B = 16
    tripped = 0
    for wl in (workloads.robot_batch(B, P=5, S=2), workloads.robot_batch(B, P=5, S=3), workloads.cstr_batch(B)):
        for rho0 in (0.1, 30.0, 1e3):
            ss = pa.sqp_settings_default(); oss = oracle.sqp_default_settings()
            for st in (ss, oss):
                st.max_iter = 4; st.line_search_max_iter = wl["ls_max_iter"]; st.hessian_update = 1
            qs = pa.qp_settings_sqp_default(); oqs = oracle.sqp_qp_default_settings()
            qs.rho = rho0; oqs.rho = rho0
            x, lam, info = ctx.sqp_solve_batch(wl["model"], wl["P"], wl["S"], wl["t0"], wl["tf"], B, wl["d"], wl["lbx"], wl["ubx"], sqp_settings=ss, qp_settings=qs)
            assert ctx.last_route() == pa.capi.ROUTE_SCHUR
            xo, lo, io = oracle.sqp_solve_batch(wl["model"], wl["P"], wl["S"], wl["t0"], wl["tf"], B, wl["d"], wl["lbx"], wl["ubx"], sqp_settings=oss, qp_settings=oqs,
                                                pivot=oracle.PIVOT_SCHUR, threads=4)
            fo = np.array([i.flags for i in io])
            print(wl["model"], wl["P"], wl["S"], rho0, "flags gpu", info["flags"].tolist(), "cpu", fo.tolist())
            assert np.array_equal(info["flags"] & pa.capi.FLAG_ILLCOND, fo & oracle.FLAG_ILLCOND)
            assert np.all(info["status"] <= pa.SQP_MAX_ITER_EXCEEDED)          # (the internal REDO status never leaves the library)
            _assert_same_solve(info, io, x, xo, lam, lo)
            tripped += int(np.count_nonzero(info["flags"] & pa.capi.FLAG_ILLCOND))
            if rho0 == 0.1:
                assert np.all(info["flags"] == 0)
                xe, _, ie = oracle.sqp_solve_batch(wl["model"], wl["P"], wl["S"], wl["t0"], wl["tf"], B, wl["d"], wl["lbx"], wl["ubx"], sqp_settings=oss, qp_settings=oqs,
                                                   pivot=oracle.PIVOT_EIGEN, threads=4)
                assert [i.qp_solver_iter for i in ie] == info["qp_solver_iter"].tolist()
                assert (np.abs(x - xe) / np.maximum(1.0, np.abs(xe).max(axis=0))).max() <= 1e-8
            if rho0 == 1e3:
                assert np.all(info["flags"] & pa.capi.FLAG_ILLCOND)
    assert tripped >= 3 * B


def test_sqp_bordered_block_structured_kernel_on_request(ctx, oracle):
    """NP = 1 on the block-structured kernel (round 5; since round 6 requested through the API, pmpc_sqp_settings::kkt_form = 2, not an environment switch;
    pmpc_schur_parking.hip): the exact Hessian of the minimal-time parking problem has the
    arrow shape — node blocks, a border row / column for the free final time, a corner — and A a dense parameter column; the QP is solved through the bordered
    range-space form. (a) minimal_time_test.cpp:146-188 as the reference configures it: every instance meets the conditioning gate once the penalty adapts upwards
    and is finished by the redo launch — SOLVED in < 20 iterations as the reference asserts, bit for bit the restatement under the same rule: which is why this form
    is not the default for that problem. (b) the same problems with the penalty held at 0.1 (adaptive_rho = 0): no trip, every QP of every iteration on the bordered kernel,
    bit-identical to PIVOT_SCHUR — the arithmetic of the border itself. (c) three iterations with the block BFGS (border row / column and corner take the same
    damped rank-2 terms, continuous_ocp.hpp:2384-2428)."""
    import os
    import polympc_amd as pa
    from test_oracle_pins import _parking_batch
    B = 12
    lbx, ubx, xg, d = _parking_batch(B)
    if True:
        for case in ("reference", "fixed_rho", "block_bfgs"):
            ss = pa.sqp_settings_default(); oss = oracle.sqp_default_settings()
            ss.kkt_form = 2
            qs = pa.qp_settings_sqp_default(); oqs = oracle.sqp_qp_default_settings()
            for st in (ss, oss):
                st.max_iter = 20; st.line_search_max_iter = 10; st.regularisation = 2
                if case == "block_bfgs": st.hessian_update = 1; st.max_iter = 3
                else: st.exact_hessian_every_iter = 1
            if case != "reference":
                qs.adaptive_rho = 0; oqs.adaptive_rho = 0
            x, lam, info = ctx.sqp_solve_batch(pa.MODEL_PARKING, 5, 2, 0.0, 1.0, B, d, lbx, ubx, x_guess=xg, sqp_settings=ss, qp_settings=qs)
            assert ctx.last_route() == pa.capi.ROUTE_SCHUR
            xo, lo, io = oracle.sqp_solve_batch(oracle.MODEL_PARKING, 5, 2, 0.0, 1.0, B, d, lbx, ubx, x_guess=xg, sqp_settings=oss, qp_settings=oqs, pivot=oracle.PIVOT_SCHUR, threads=4)
            fo = np.array([i.flags for i in io])
            print(case, "flags", info["flags"].tolist(), "iter", info["iter"].tolist(), "status", info["status"].tolist())
            assert np.array_equal(info["flags"], fo)
            _assert_same_solve(info, io, x, xo, lam, lo)
            if case == "reference":
                assert np.all(info["flags"] & pa.capi.FLAG_ILLCOND) and (info["status"] == pa.SQP_SOLVED).sum() >= B - 1 and np.all(info["iter"][info["status"] == pa.SQP_SOLVED] < 20)
            else:
                assert np.all(info["flags"] == 0)
    ss = pa.sqp_settings_default(); ss.max_iter = 2; ss.regularisation = 2; ss.exact_hessian_every_iter = 1
    ctx.sqp_solve_batch(pa.MODEL_PARKING, 5, 2, 0.0, 1.0, 1, d[:1], lbx[:1], ubx[:1], x_guess=xg[:1], sqp_settings=ss)
    assert ctx.last_route() != pa.capi.ROUTE_SCHUR   # without the request: the dense two-rows-per-lane kernel, as before


@pytest.mark.parametrize("case", ["robot_11", "robot_16", "cstr_11", "robot_7", "kite"])
def test_conditioning_gate_and_the_redo_launch(ctx, oracle, case):
    """PMPC_FLAG_ILLCOND end to end. Controls unbounded and the QP penalty started at 1e4: the kernels that eliminate the constraint block first meet
    cond(S) > 1e10 in the first QP. The one-row-per-lane kernel (robot, 7 nodes) continues that QP on the full primal-first sweep; the condensed register
    kernels (robot 11 / 16 nodes, CSTR) and the large-instance kernel (kite stand-in) give the QP up and the launcher's redo launch solves the instance
    again in the full KKT form. Bit for bit what the restatement does under the same rule, flag included; with the bounds in place the flag stays clear."""
    import polympc_amd as pa
    from polympc_amd import workloads
    B = 6
    wl, nx = {"robot_11": (workloads.robot_batch(B, P=5, S=2), 3), "robot_16": (workloads.robot_batch(B, P=5, S=3), 3), "cstr_11": (workloads.cstr_batch(B), 4),
              "robot_7": (workloads.robot_batch(B), 3), "kite": (workloads.kite_standin_batch(B), 13)}[case]
    n, m = wl["n"], wl["m"]
    order = oracle.PIVOT_SWEEP if n + m <= 64 else (oracle.PIVOT_CONDSWEEP if n + m <= 128 else oracle.PIVOT_CONDENSED)
    for free in (False, True):
        w = _free_controls(wl, nx) if free else wl
        ss = pa.sqp_settings_default(); oss = oracle.sqp_default_settings()
        for st in (ss, oss):
            st.max_iter = 3; st.line_search_max_iter = wl["ls_max_iter"]
        qs = pa.qp_settings_sqp_default(); oqs = oracle.sqp_qp_default_settings()
        qs.rho = 1e4; oqs.rho = 1e4
        x, lam, info = ctx.sqp_solve_batch(w["model"], w["P"], w["S"], w["t0"], w["tf"], B, w["d"], w["lbx"], w["ubx"], sqp_settings=ss, qp_settings=qs)
        xo, lo, io = oracle.sqp_solve_batch(w["model"], w["P"], w["S"], w["t0"], w["tf"], B, w["d"], w["lbx"], w["ubx"], sqp_settings=oss, qp_settings=oqs,
                                            pivot=order, threads=4)
        fo = np.array([i.flags for i in io])
        print(case, "free controls" if free else "bounded", "flags gpu", info["flags"].tolist(), "cpu", fo.tolist(), "status", info["status"].tolist())
        assert np.array_equal(info["flags"] & pa.capi.FLAG_ILLCOND, fo & oracle.FLAG_ILLCOND)
        assert np.all(info["status"] <= pa.SQP_MAX_ITER_EXCEEDED)          # (the internal REDO status never leaves the library)
        _assert_same_solve(info, io, x, xo, lam, lo)
        if free:
            assert np.all(info["flags"] & pa.capi.FLAG_ILLCOND)
        else:
            assert np.all(info["flags"] == 0)


def ROUTE_OF_128_ROWS(pa):
    """the kernel family pmpc_launch.hpp routes 128-row instances to (one place to change when the route changes)"""
    return pa.capi.ROUTE_CONDREG   # round 4: 80 variables, 48 constraint rows — the condensed register kernel (round 3: the two-rows-per-lane full inverse with sixteen operand tiles in LDS, still behind kkt_form = 1)


def test_last_route_reports_the_kernel_family(ctx):
    """pmpc_sqp_last_route ("no log/flag telling a caller which path served the call"): the default policy on a 7-node grid runs the
    one-row-per-lane register kernel, 11 nodes the two-rows-per-lane one, a policy hook the register kernels do not carry (OSQP-form ADMM) the
    LDS-resident kernel, 464 rows the HBM-factor kernel."""
    import polympc_amd as pa
    from polympc_amd import workloads
    def route(wl, **kw):
        ss = pa.sqp_settings_default(); ss.max_iter = 2; ss.line_search_max_iter = 4
        for k, v in kw.items():
            setattr(ss, k, v)
        B = wl["lbx"].shape[0]
        ctx.sqp_solve_batch(wl["model"], wl["P"], wl["S"], wl["t0"], wl["tf"], B, wl["d"], wl["lbx"], wl["ubx"], sqp_settings=ss)
        return ctx.last_route()
    assert route(workloads.robot_batch(4)) == pa.capi.ROUTE_REG1
    assert route(workloads.robot_batch(4, P=5, S=2)) == pa.capi.ROUTE_CONDREG                  # 55 variables, 33 constraint rows: the condensed register kernel on the one-row-per-lane tile set (round 4)
    assert route(workloads.robot_batch(4, P=5, S=2), kkt_form=1) == pa.capi.ROUTE_REG2
    assert route(workloads.cstr_batch(4)) == pa.capi.ROUTE_CONDREG                             # 66 variables, 44 constraint rows: the condensed register kernel (round 4)
    assert route(workloads.cstr_batch(4), kkt_form=1) == pa.capi.ROUTE_REG2
    assert route(workloads.robot_batch(4), qp_solver=1) == pa.capi.ROUTE_LDS
    # round 3: the Ruiz preconditioner and the filter line search on the 7- and 11-node register kernels; other grids keep the LDS / HBM kernels for them
    assert route(workloads.robot_batch(4), preconditioner=1) == pa.capi.ROUTE_REG1
    assert route(workloads.robot_batch(4), line_search=1, hessian_update=1) == pa.capi.ROUTE_REG1
    # round 4: a block-diagonal Hessian (block BFGS / exact Hessians) on a grid with a block-structured kernel
    assert route(workloads.robot_batch(4, P=5, S=2), hessian_update=1) == pa.capi.ROUTE_SCHUR
    assert route(workloads.cstr_batch(4), hessian_update=1) == pa.capi.ROUTE_SCHUR
    assert route(workloads.robot_batch(4), hessian_update=1) == pa.capi.ROUTE_REG1                # 56 KKT rows: the dense one-row-per-lane kernel is faster
    assert route(workloads.robot_batch(4, P=5, S=3), exact_hessian_every_iter=1) == pa.capi.ROUTE_SCHUR
    assert route(workloads.robot_batch(4, P=4, S=1), hessian_update=1) == pa.capi.ROUTE_REG1      # no block-structured kernel for this grid
    assert route(workloads.robot_batch(4, P=5, S=2), hessian_update=1, kkt_form=1) == pa.capi.ROUTE_REG2
    assert route(workloads.robot_batch(4, P=5, S=2), preconditioner=1, line_search=1) == pa.capi.ROUTE_CONDREG   # late round 6: the Ruiz preconditioner on the condensed hook kernels (tables and node blocks from the scaled workspace); before: the full two-rows-per-lane inverse
    assert route(workloads.robot_batch(4, P=5, S=2), preconditioner=1, line_search=1, kkt_form=1) == pa.capi.ROUTE_REG2
    assert route(workloads.robot_batch(4, P=5, S=2), line_search=1) == pa.capi.ROUTE_CONDREG            # round 5: the hook build of the small condensed kernel — then compiled WITHOUT the Ruiz calls (they were what miscompiled it, EXPERIMENTS.md)
    assert route(workloads.cstr_batch(4), line_search=1) == pa.capi.ROUTE_CONDREG
    assert route(workloads.robot_batch(4, P=4, S=2), preconditioner=1) == pa.capi.ROUTE_LDS
    assert route(workloads.robot_batch(4, P=5, S=3)) == pa.capi.ROUTE_CONDREG
    assert route(workloads.robot_batch(4, P=5, S=3), kkt_form=1) == pa.capi.ROUTE_REG2
    assert route(workloads.robot_batch(4, P=5, S=3), line_search=1) == pa.capi.ROUTE_CONDREG             # round 4: the policy hooks on the 16-node register kernels; the filter line search alone keeps the condensed QP
    assert route(workloads.robot_batch(4, P=5, S=3), preconditioner=1) == pa.capi.ROUTE_CONDREG
    assert route(workloads.robot_batch(4, P=5, S=3), preconditioner=1, kkt_form=1) == pa.capi.ROUTE_REG2
    assert route(workloads.robot_batch(4, P=5, S=4), line_search=1) == pa.capi.ROUTE_HBM
    assert route(workloads.kite_standin_batch(2)) == pa.capi.ROUTE_HBM


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["C_kite_standin_1024", "R_robot_16_nodes_2048"])
def test_sqp_full_size_properties_hbm_factor_kernel(ctx, oracle, case):
    """BASELINE size of config C (1024 instances of the 464-row stand-in, HBM-factor kernel) and 2048 instances on the reference's 16-node robot grid
    (128 rows: since round 3 the two-rows-per-lane register kernel with LDS-resident operand tiles): SOLVED fractions, finite, bounds respected, the reported constraint violation equal to what the SEPARATE collocation
    kernel (pmpc_ocp_linearise_batch) evaluates at the returned point, and a sample of instances spread over the batch bit-identical to the
    blocked-order CPU restatement run on those instances alone (the batch position must not matter)."""
    import polympc_amd as pa
    from polympc_amd import workloads
    if case.startswith("C"):
        B = 1024; wl = workloads.kite_standin_batch(B); min_solved = 0.99
    else:
        B = 2048; wl = workloads.robot_batch(B, P=5, S=3); min_solved = 0.8
    P, S = wl["P"], wl["S"]
    ss = pa.sqp_settings_default(); ss.max_iter = wl["max_iter"]; ss.line_search_max_iter = wl["ls_max_iter"]
    x, lam, info = ctx.sqp_solve_batch(wl["model"], P, S, wl["t0"], wl["tf"], B, wl["d"], wl["lbx"], wl["ubx"], sqp_settings=ss)
    ok = info["status"] == pa.SQP_SOLVED
    assert ok.mean() >= min_solved and np.all(info["flags"] == 0) and np.all(np.isfinite(x)) and np.all(np.isfinite(lam))
    viol = np.zeros(B)
    for b0 in range(0, B, 128):
        ev = ctx.ocp_linearise_batch(wl["model"], P, S, wl["t0"], wl["tf"], x[b0:b0 + 128], wl["d"][b0:b0 + 128])
        viol[b0:b0 + 128] = np.abs(ev["c"]).max(axis=1)
    viol = np.maximum(viol, np.maximum((wl["lbx"] - x).max(axis=1), (x - wl["ubx"]).max(axis=1)))
    assert np.abs(viol - info["max_violation"]).max() <= 1e-12 * max(1.0, viol.max())
    assert viol[ok].max() <= ss.eps_prim
    sample = np.array([0, 1, B // 3, B // 2 + 7, B - 2, B - 1])
    oss = oracle.sqp_default_settings(); oss.max_iter = wl["max_iter"]; oss.line_search_max_iter = wl["ls_max_iter"]
    xo, lo, io = oracle.sqp_solve_batch(wl["model"], P, S, wl["t0"], wl["tf"], len(sample), wl["d"][sample], wl["lbx"][sample], wl["ubx"][sample],
                                        sqp_settings=oss, pivot=_gpu_order(oracle, wl["n"], wl["m"], nodes=P * S + 1), threads=6)
    _assert_same_solve(info[sample], io, x[sample], xo, lam[sample], lo)


def test_large_instance_team_kernel_bit_identical_to_the_one_wavefront_kernel(oracle, monkeypatch):
    """The large-instance kernel on a workgroup of four wavefronts per instance (BigTeam, pmpc_qp_big.hpp; batches of at most one instance per CU) against
    the one-wavefront kernel (PMPC_BIG_WG4 = 0) and against the restatement: the team deals whole fma chains — tiles of the condensed build, tile-row groups of the
    blocked factorisation, 64-row slots / column quadruples of the triangular passes, chunks of the sparse products, entries of the second-order AD stage — so every
    bit must be the same; with the block BFGS too (another linearisation path into the same QP)."""
    import polympc_amd as pa
    from polympc_amd import workloads
    B = 5
    wl = workloads.kite_standin_batch(B)
    for hu in (0, 1):
        ss = pa.sqp_settings_default(); ss.max_iter = wl["max_iter"]; ss.line_search_max_iter = wl["ls_max_iter"]; ss.hessian_update = hu
        res = []
        for wg4 in ("1", "0"):
            monkeypatch.setenv("PMPC_BIG_WG4", wg4)
            c = pa.Context(0)
            try:
                res.append(c.sqp_solve_batch(wl["model"], wl["P"], wl["S"], wl["t0"], wl["tf"], B, wl["d"], wl["lbx"], wl["ubx"], sqp_settings=ss))
                assert c.last_route() == pa.capi.ROUTE_HBM
            finally:
                c.close()
        monkeypatch.delenv("PMPC_BIG_WG4")
        (x4, l4, i4), (x1, l1, i1) = res
        assert _same_bits(x4, x1) and _same_bits(l4, l1) and _same_bits(i4, i1)
        oss = oracle.sqp_default_settings(); oss.max_iter = wl["max_iter"]; oss.line_search_max_iter = wl["ls_max_iter"]; oss.hessian_update = hu
        xo, lo, io = oracle.sqp_solve_batch(wl["model"], wl["P"], wl["S"], wl["t0"], wl["tf"], B, wl["d"], wl["lbx"], wl["ubx"], sqp_settings=oss,
                                            pivot=oracle.PIVOT_CONDENSED, threads=4)
        _assert_same_solve(i4, io, x4, xo, l4, lo)
    # between one and two instances per CU (257 .. 512 on this device) a second build of the team kernel — compiled for 256 registers, two workgroups per CU —
    # serves the batch by default: the same bits as one wavefront per instance
    B = 300
    wl = workloads.kite_standin_batch(B)
    ss = pa.sqp_settings_default(); ss.max_iter = 3; ss.line_search_max_iter = wl["ls_max_iter"]
    res = []
    for wg4 in (None, "0"):
        if wg4 is not None: monkeypatch.setenv("PMPC_BIG_WG4", wg4)
        c = pa.Context(0)
        try:
            res.append(c.sqp_solve_batch(wl["model"], wl["P"], wl["S"], wl["t0"], wl["tf"], B, wl["d"], wl["lbx"], wl["ubx"], sqp_settings=ss))
        finally:
            c.close()
    monkeypatch.delenv("PMPC_BIG_WG4")
    (x4, l4, i4), (x1, l1, i1) = res
    assert _same_bits(x4, x1) and _same_bits(l4, l1) and _same_bits(i4, i1)
    # a MID-SIZE instance (round 6, ADVICE r5: the team's thresholds were measured on the kite-sized 464-row system only): the 21-node robot grid, 105 + 63 =
    # 168 KKT rows, which the launcher also hands to the team for batches of at most two instances per CU — lone instance, one and two instances per CU, against
    # the one-wavefront kernel (same-box timings: profiles/r06_ab_prologue_rule_and_midsize_team.txt — the team is faster at 1 / 64 / 256 / 512 instances)
    for B in (1, 256, 300):
        wl = workloads.robot_batch(B, P=5, S=4)
        ss = pa.sqp_settings_default(); ss.max_iter = 4; ss.line_search_max_iter = wl["ls_max_iter"]
        res = []
        for wg4 in (None, "0"):
            if wg4 is not None: monkeypatch.setenv("PMPC_BIG_WG4", wg4)
            c = pa.Context(0)
            try:
                res.append(c.sqp_solve_batch(wl["model"], wl["P"], wl["S"], wl["t0"], wl["tf"], B, wl["d"], wl["lbx"], wl["ubx"], sqp_settings=ss))
                assert c.last_route() == pa.capi.ROUTE_HBM
            finally:
                c.close()
        monkeypatch.delenv("PMPC_BIG_WG4")
        (x4, l4, i4), (x1, l1, i1) = res
        assert _same_bits(x4, x1) and _same_bits(l4, l1) and _same_bits(i4, i1), B
    oss = oracle.sqp_default_settings(); oss.max_iter = 4; oss.line_search_max_iter = wl["ls_max_iter"]
    xo, lo, io = oracle.sqp_solve_batch(wl["model"], wl["P"], wl["S"], wl["t0"], wl["tf"], 16, wl["d"][:16], wl["lbx"][:16], wl["ubx"][:16], sqp_settings=oss,
                                        pivot=oracle.PIVOT_CONDENSED, threads=4)
    _assert_same_solve(i4[:16], io, x4[:16], xo, l4[:16], lo)


def test_device_poisoning_changes_nothing(oracle):
    """pmpc_debug_set_poison (PMPC_POISON=1): signalling NaNs in the HBM workspace, the staging buffers, every CU's LDS, every SIMD's register files and the low
    scratch before each launch — a kernel that read what it never wrote would return NaN. One solve per kernel family with poisoning on: bit-identical to the same
    context's solve without it (the whole suite runs under PMPC_POISON=1 as a release check; this test keeps the switch itself alive)."""
    import polympc_amd as pa
    from polympc_amd import workloads
    c = pa.Context(0)
    try:
        cases = [(workloads.robot_batch(8), {}), (workloads.robot_batch(6, P=5, S=2), {}), (workloads.cstr_batch(6), {}), (workloads.robot_batch(6, P=5, S=3), dict(hessian_update=1)),
                 (workloads.robot_batch(6, P=4, S=1), dict(qp_solver=1)), (workloads.kite_standin_batch(2), {})]
        for wl, kw in cases:
            ss = pa.sqp_settings_default(); ss.max_iter = min(4, wl["max_iter"]); ss.line_search_max_iter = wl["ls_max_iter"]
            for k, v in kw.items(): setattr(ss, k, v)
            B = wl["lbx"].shape[0]
            run = lambda: c.sqp_solve_batch(wl["model"], wl["P"], wl["S"], wl["t0"], wl["tf"], B, wl["d"], wl["lbx"], wl["ubx"], sqp_settings=ss)
            c.set_poison(False); a = run()
            c.set_poison(True); b = run(); route = c.last_route()
            c.set_poison(False)
            assert _same_bits(a[0], b[0]) and _same_bits(a[1], b[1]) and _same_bits(a[2], b[2]), pa.capi.ROUTE_NAMES.get(route)
            assert np.isfinite(b[0]).all()
    finally:
        c.close()


def test_qp_entry_conditioning_gate_and_its_redo_launch(ctx, oracle):
    """The QP entry point's one-row-per-lane kernels gate numerically at every factorisation (max S_ii * max |(S^-1)_ii| > 1e10, read off the swept tiles): config A's
    QP stream with every bound removed and the penalty started at 1e5 — every QP gives up and the redo launch (one workgroup per 64 QPs looks at their flags) solves
    it on the LDS-resident static LDL^T: bit for bit the restatement under the same rule (PIVOT_SWEEP -> PIVOT_STATIC), flag set; with the bounds in place
    nothing trips at any penalty and the kernel is the constraint-first sweep as before."""
    import polympc_amd as pa
    from oracle import cross_order as tco
    q = tco.traced_qp_stream(oracle, "A", 200)
    free_l, free_u = np.full_like(q["xlb"], -np.inf), np.full_like(q["xub"], np.inf)
    for rho0 in (0.1, 1e3, 1e5):
        qs = pa.qp_settings_sqp_default(); qs.rho = rho0
        oqs = oracle.sqp_qp_default_settings(); oqs.rho = rho0
        for lo, hi, free in ((q["xlb"], q["xub"], False), (free_l, free_u, True)):
            x, y, info = ctx.qp_solve_batch(q["H"], q["h"], q["A"], q["Alb"], q["Aub"], lo, hi, settings=qs)
            xo, yo, io = oracle.qp_solve_batch(q["H"], q["h"], q["A"], q["Alb"], q["Aub"], lo, hi, settings=oqs, pivot=oracle.PIVOT_SWEEP, threads=8)
            fo = np.array([i.flags for i in io])
            assert np.array_equal(info["flags"] & pa.capi.FLAG_ILLCOND, fo & oracle.FLAG_ILLCOND), (rho0, free)
            assert np.array_equal(info["iter"], np.array([i.iter for i in io])) and np.array_equal(info["status"], np.array([i.status for i in io])), (rho0, free)
            assert np.array_equal(x, xo) and np.array_equal(y, yo), (rho0, free, np.abs(x - xo).max())
            if not free: assert np.all(info["flags"] == 0)
            if free and rho0 == 1e5: assert np.all(info["flags"] & pa.capi.FLAG_ILLCOND)


def test_empty_and_odd_batch_sizes(ctx, oracle):
    """Edge cases of the batch dimension through the C ABI: an empty batch is accepted and touches nothing; batches of 1, 63, 65 and 130 instances (not multiples of
    the 64 QPs a redo-launch workgroup scans, nor of anything else) give, instance by instance, exactly what the same instances give in a batch of 130 — on the QP
    entry point (one-row-per-lane kernel, with and without its conditioning gate tripping) and on the fused SQP kernels of four routes."""
    import polympc_amd as pa
    from polympc_amd import workloads
    from oracle import cross_order as tco
    q = tco.traced_qp_stream(oracle, "A", 130)
    q = {k: v[:130] for k, v in q.items() if isinstance(v, np.ndarray)}
    n, m = q["h"].shape[1], q["Alb"].shape[1]
    x, y, info = ctx.qp_solve_batch(q["H"][:0], q["h"][:0], q["A"][:0], q["Alb"][:0], q["Aub"][:0], q["xlb"][:0], q["xub"][:0])
    assert x.shape == (0, n) and y.shape == (0, n + m) and len(info["iter"]) == 0
    for free in (False, True):
        lo = np.full_like(q["xlb"], -np.inf) if free else q["xlb"]; hi = np.full_like(q["xub"], np.inf) if free else q["xub"]
        qs = pa.qp_settings_sqp_default(); qs.rho = 1e5 if free else 0.1
        xf, yf, inf_ = ctx.qp_solve_batch(q["H"], q["h"], q["A"], q["Alb"], q["Aub"], lo, hi, settings=qs)
        assert bool(np.all(inf_["flags"] & pa.capi.FLAG_ILLCOND)) == free
        for B in (1, 63, 65):
            xb, yb, ib = ctx.qp_solve_batch(q["H"][:B], q["h"][:B], q["A"][:B], q["Alb"][:B], q["Aub"][:B], lo[:B], hi[:B], settings=qs)
            assert np.array_equal(xb, xf[:B]) and np.array_equal(yb, yf[:B]) and np.array_equal(ib["iter"], inf_["iter"][:B]) and np.array_equal(ib["flags"], inf_["flags"][:B])
    for wl, kw, route in ((workloads.robot_batch(130), {}, pa.capi.ROUTE_REG1), (workloads.robot_batch(130, P=5, S=2), {}, pa.capi.ROUTE_CONDREG),
                          (workloads.robot_batch(130, P=5, S=2), dict(hessian_update=1), pa.capi.ROUTE_SCHUR), (workloads.kite_standin_batch(5), {}, pa.capi.ROUTE_HBM)):
        ss = pa.sqp_settings_default(); ss.max_iter = 3; ss.line_search_max_iter = wl["ls_max_iter"]
        for k, v in kw.items(): setattr(ss, k, v)
        nB = wl["lbx"].shape[0]
        run = lambda B: ctx.sqp_solve_batch(wl["model"], wl["P"], wl["S"], wl["t0"], wl["tf"], B, wl["d"][:B], wl["lbx"][:B], wl["ubx"][:B], sqp_settings=ss)
        xf, lf, inf_ = run(nB)
        assert ctx.last_route() == route
        x0, l0, i0 = run(0)
        assert x0.shape[0] == 0 and l0.shape[0] == 0
        for B in ((1, 63, 65) if nB >= 65 else (1, 3)):
            xb, lb, ib = run(B)
            assert np.array_equal(xb, xf[:B]) and np.array_equal(lb, lf[:B]) and np.array_equal(ib["qp_solver_iter"], inf_["qp_solver_iter"][:B])


@pytest.mark.gpu
def test_routing_table_soak_under_poison():
    """The developer soak as a test (late round 6): tests/tools_soak_routes.py in its own process with PMPC_POISON=1 — every grid of 3 .. 16 nodes of five models (robot, CSTR,
    parking with one parameter, parking and robot with a path constraint) under the default policy and the seven policy sets that change the route, every launch preceded by
    signalling NaNs in workspaces, LDS and register files: every combination bit-identical to the restatement in the order the dispatch rule names for it. This is the run that
    showed compiler hazard 3 in a second kernel the moment it appeared; the CPU suite's ISA check named the same kernel."""
    import subprocess
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, PMPC_POISON="1")
    r = subprocess.run([sys.executable, os.path.join(here, "tools_soak_routes.py"), "4"], capture_output=True, text=True, env=env, timeout=900)
    lines = [l for l in r.stdout.splitlines() if l.startswith("model ")]
    bad = [l for l in lines if "MISMATCH" in l]
    assert r.returncode == 0 and not bad and len(lines) >= 600 and r.stdout.strip().endswith("mismatches: 0"), (r.returncode, bad[:5], r.stdout[-400:], r.stderr[-400:])
