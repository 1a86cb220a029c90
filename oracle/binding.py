"""ORACLE — TEST INFRASTRUCTURE ONLY.

ctypes binding of oracle/liboracle.so (the CPU restatement of the reference algorithm) and, when present,
oracle/_ref/libref_casadi_robot.so (the reference's own CasADi-generated C fixtures, compiled by oracle/Makefile).
Import this only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg — never from polympc_amd.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "liboracle.so")
REF_PATH = os.path.join(HERE, "_ref", "libref_casadi_robot.so")

PIVOT_EIGEN, PIVOT_STATIC, PIVOT_SWEEP, PIVOT_SWEEP1, PIVOT_SWEEP2, PIVOT_BLOCKED, PIVOT_CONDENSED, PIVOT_SCHUR, PIVOT_CONDSWEEP, PIVOT_EXACT = 0, 1, 2, 3, 4, 5, 6, 7, 8, 9
FLAG_ILLCOND = 2   # qp / sqp info flags: the conditioning gate of the condensed / constraint-first orders tripped (BoxADMM::COND_GATE)
SCHUR_MAX_ROWS = 64    # PIVOT_SCHUR restates the block-structured kernel: at most 64 constraint rows (its m x m Schur complement is swept like PIVOT_SWEEP)
SWEEP2_MAX_ROWS = 128   # PIVOT_SWEEP2 restates the two-rows-per-lane register kernel (65..128 KKT rows)


def _check_sweep(pivot, rows):
    """PIVOT_SWEEP restates the register-resident kernel, which exists for KKT systems of at most 64 rows."""
    if pivot == PIVOT_SWEEP and rows > 64:
        raise ValueError(f"PIVOT_SWEEP supports at most 64 KKT rows (got {rows}); use PIVOT_STATIC or PIVOT_EIGEN")
    if pivot == PIVOT_SWEEP2 and rows > SWEEP2_MAX_ROWS:
        raise ValueError(f"PIVOT_SWEEP2 supports at most {SWEEP2_MAX_ROWS} KKT rows (got {rows}); use PIVOT_STATIC or PIVOT_EIGEN")
MODEL_ROBOT, MODEL_CSTR, MODEL_PARKING, MODEL_ROBOT_NG, MODEL_KITE_STANDIN, MODEL_PARKING_NG = 0, 1, 2, 3, 4, 5
NLP_CONSTRAINED_ROSENBROCK, NLP_ROSENBROCK, NLP_SIMPLE, NLP_HS071 = 0, 1, 2, 3
QP_SOLVED, QP_MAX_ITER_EXCEEDED, QP_UNSOLVED = 0, 1, 2
SQP_SOLVED, SQP_MAX_ITER_EXCEEDED = 0, 1


class QPSettings(C.Structure):
    _fields_ = [("eps_rel", C.c_double), ("eps_abs", C.c_double), ("max_iter", C.c_int), ("rho", C.c_double),
                ("sigma", C.c_double), ("alpha", C.c_double), ("check_termination", C.c_int),
                ("adaptive_rho", C.c_int), ("adaptive_rho_tolerance", C.c_double), ("adaptive_rho_interval", C.c_int)]


class QPInfo(C.Structure):
    _fields_ = [("status", C.c_int), ("iter", C.c_int), ("rho_updates", C.c_int), ("flags", C.c_int), ("rho_estimate", C.c_double),
                ("res_prim", C.c_double), ("res_dual", C.c_double)]


class SQPSettings(C.Structure):
    _fields_ = [("tau", C.c_double), ("eta", C.c_double), ("rho", C.c_double), ("eps_prim", C.c_double),
                ("eps_dual", C.c_double), ("max_iter", C.c_int), ("line_search_max_iter", C.c_int),
                ("regularisation", C.c_int), ("exact_hessian_every_iter", C.c_int), ("preconditioner", C.c_int), ("hessian_update", C.c_int), ("qp_solver", C.c_int),
                ("line_search", C.c_int), ("filter_max_depth", C.c_int), ("filter_beta", C.c_double), ("filter_state", C.POINTER(C.c_double)),
                ("iteration_trace", C.POINTER(C.c_double)), ("iteration_trace_capacity", C.c_int)]


FILTER_STATE_DOUBLES = 21
TRACE_DOUBLES = 8


def bind_iteration_trace(ss, trace):
    """Attach a (B, capacity, TRACE_DOUBLES) float64 array that receives the per-iteration records; None detaches it."""
    if trace is None:
        ss.iteration_trace = C.POINTER(C.c_double)(); ss.iteration_trace_capacity = 0
    else:
        assert trace.dtype == np.float64 and trace.flags.c_contiguous and trace.shape[-1] == TRACE_DOUBLES
        ss.iteration_trace = trace.ctypes.data_as(C.POINTER(C.c_double)); ss.iteration_trace_capacity = trace.shape[-2]
        ss._keep_trace = trace


def bind_filter_state(ss, state):
    """Attach a (B, FILTER_STATE_DOUBLES) float64 array as the LSFilter member that outlives solve(); None detaches it."""
    if state is None:
        ss.filter_state = C.POINTER(C.c_double)()
    else:
        assert state.dtype == np.float64 and state.flags.c_contiguous and state.shape[-1] == FILTER_STATE_DOUBLES
        ss.filter_state = state.ctypes.data_as(C.POINTER(C.c_double))
        ss._keep_filter = state


class SQPInfo(C.Structure):
    _fields_ = [("iter", C.c_int), ("qp_solver_iter", C.c_int), ("status", C.c_int), ("flags", C.c_int), ("primal_norm", C.c_double),
                ("dual_norm", C.c_double), ("max_violation", C.c_double), ("cost", C.c_double)]


def build(force=False):
    if force or not os.path.exists(LIB_PATH):
        subprocess.check_call(["make", "-C", HERE, "-s", "all"])
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(LIB_PATH)
    return _lib


def _p(a):
    if a is None:
        return None
    assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _f(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float64)


def set_libm(flag):
    """Model transcendental functions: False (default) = the IEEE-only restatement shared with the HIP kernels, True = glibc (what the
    reference binary calls). Returns the previous setting."""
    f = lib().orc_set_libm; f.argtypes = [C.c_int]; f.restype = C.c_int
    return bool(f(1 if flag else 0))


class libm:
    """with libm(): ... — evaluate the models with glibc inside the block."""
    def __enter__(self):
        self.old = set_libm(True)
    def __exit__(self, *a):
        set_libm(self.old)


def math_eval(kind, x, impl=0):
    """kind: 'sin' | 'cos' | 'exp'; impl 0 = detmath, 1 = glibc."""
    x = _f(x); y = np.zeros_like(x)
    lib().orc_math_eval({"sin": 0, "cos": 1, "exp": 2}[kind], impl, x.size, _p(x), _p(y))
    return y


def qp_default_settings():
    s = QPSettings(); lib().orc_qp_default_settings(C.byref(s)); return s


def sqp_qp_default_settings():
    s = QPSettings(); lib().orc_sqp_qp_default_settings(C.byref(s)); return s


def sqp_default_settings():
    s = SQPSettings(); lib().orc_sqp_default_settings(C.byref(s)); return s


def cheb(P):
    nodes = np.zeros(P + 1); w = np.zeros(P + 1); D = np.zeros((P + 1) * (P + 1))
    lib().orc_cheb(P, _p(nodes), _p(w), _p(D))
    return nodes, w, D.reshape(P + 1, P + 1).T.copy()  # column-major -> D[i, j]


def classify(lb, ub):
    f = lib().orc_classify; f.argtypes = [C.c_double, C.c_double]; f.restype = C.c_int
    return f(lb, ub)


def bfgs(B, s, y):
    n = len(s)
    Bc = np.ascontiguousarray(np.asarray(B, dtype=np.float64).T).ravel().copy()  # column-major
    lib().orc_bfgs(n, _p(Bc), _p(_f(s)), _p(_f(y)))
    return Bc.reshape(n, n).T.copy()


def regularise(kind, H):
    n = H.shape[0]
    Hc = np.ascontiguousarray(np.asarray(H, dtype=np.float64).T).ravel().copy()
    lib().orc_regularise(kind, n, _p(Hc))
    return Hc.reshape(n, n).T.copy()


def ldlt_solve(K, b, pivot=PIVOT_EIGEN):
    n = len(b)
    _check_sweep(pivot, n)
    Kc = np.ascontiguousarray(np.asarray(K, dtype=np.float64).T).ravel().copy()
    x = np.zeros(n)
    lib().orc_ldlt_solve(n, _p(Kc), _p(_f(b)), pivot, _p(x))
    return x


def _schur_check(structure, n, m, H=None):
    """PIVOT_SCHUR needs the collocation structure (nx, nu, nn, P) of the QP — and a Hessian that is block diagonal per node (checked here in numpy:
    the C++ side throws inside an OpenMP region otherwise)."""
    if structure is None:
        raise ValueError("PIVOT_SCHUR: pass structure=(nx, nu, nn, P) or (nx, nu, nn, P, np)")
    nx, nu, nn, P = structure[:4]
    npar = structure[4] if len(structure) > 4 else 0   # one parameter behind the node variables: the bordered form
    if npar not in (0, 1) or (nx + nu) * nn + npar != n or nx * nn != m or m > SCHUR_MAX_ROWS or P < 1 or (nn - 1) % P != 0:
        raise ValueError(f"PIVOT_SCHUR: structure {structure} does not describe a QP with n = {n}, m = {m} <= {SCHUR_MAX_ROWS}")
    if H is not None:
        node = np.concatenate([np.repeat(np.arange(nn), nx), np.repeat(np.arange(nn), nu)])
        off = node[:, None] != node[None, :]
        n0 = n - npar
        if np.any(H.reshape(-1, n, n)[:, :n0, :n0][:, off] != 0.0):
            raise ValueError("PIVOT_SCHUR: the Hessian is not block diagonal per collocation node")
    lib().orc_set_schur_structure_np(nx, nu, nn, P, npar)


COND_MAX_ROWS = 112     # PIVOT_CONDSWEEP restates the condensed register kernel: 65..112 variables (at most 64 also works: PIVOT_SWEEP's mat-vec), at most 64 constraint rows


def _cond_check(structure, n, m):
    """PIVOT_CONDSWEEP needs the collocation structure (nx, nu, nn, P) of the QP: its sparse products walk the nodes."""
    if structure is None:
        raise ValueError("PIVOT_CONDSWEEP: pass structure=(nx, nu, nn, P)")
    nx, nu, nn, P = structure[:4]
    if (nx + nu) * nn != n or m < nx * nn or (m - nx * nn) % nn != 0 or m > 64 or n > COND_MAX_ROWS:
        raise ValueError(f"PIVOT_CONDSWEEP: structure {structure} does not describe a QP with n = {n} <= {COND_MAX_ROWS}, m = {m} <= 64")
    lib().orc_set_schur_structure(nx, nu, nn, P)


def kkt_solve(K, rho_vec, rhs, pivot=PIVOT_EIGEN, structure=None):
    """One KKT solve in the order of `pivot`: K = [P A'; A -1/rho] ([n+m, n+m] array, lower triangle read), rho_vec (m), rhs (n+m)."""
    K = np.asarray(K, dtype=np.float64); m = len(rho_vec); n = K.shape[0] - m
    if pivot == PIVOT_SCHUR:
        _schur_check(structure, n, m)
    if pivot == PIVOT_CONDSWEEP:
        _cond_check(structure, n, m)
    _check_sweep(pivot, n + m)
    sol = np.zeros(n + m)
    Kc = np.ascontiguousarray(K.T).ravel().copy()
    lib().orc_kkt_solve(n, m, _p(Kc), _p(_f(rho_vec)), _p(_f(rhs)), pivot, _p(sol))
    return sol


def qp_solve_batch(H, h, A, Alb, Aub, xlb, xub, settings=None, pivot=PIVOT_EIGEN, x0=None, y0=None, threads=1, structure=None):
    """All arrays instance-major; matrices column-major per instance: H[b] is (n*n,), A[b] is (m*n,). structure = (nx, nu, nn, P): PIVOT_SCHUR only."""
    H = _f(H); h = _f(h); A = _f(A); Alb = _f(Alb); Aub = _f(Aub); xlb = _f(xlb); xub = _f(xub)
    B, n = h.shape
    m = Alb.shape[1] if Alb.ndim == 2 else 0
    if pivot == PIVOT_SCHUR:
        _schur_check(structure, n, m, H)
    if pivot == PIVOT_CONDSWEEP:
        _cond_check(structure, n, m)
    _check_sweep(pivot, n + m)
    s = settings or qp_default_settings()
    x = np.zeros((B, n)); y = np.zeros((B, n + m)); info = (QPInfo * B)()
    lib().orc_qp_solve_batch(B, n, m, _p(H), _p(h), _p(A), _p(Alb), _p(Aub), _p(xlb), _p(xub), _p(_f(x0)), _p(_f(y0)),
                             C.byref(s), pivot, threads, _p(x), _p(y), info)
    return x, y, info


def qp_solve_batch_f32(H, h, A, Alb, Aub, xlb, xub, settings=None, pivot=PIVOT_EIGEN, x0=None, y0=None, osqp_form=False):
    """boxADMM<N, M, float> (osqp_form: ADMM<N, M, float>): the same layout as qp_solve_batch with float32 arrays (pivot: PIVOT_EIGEN or PIVOT_STATIC)."""
    f32 = lambda a: None if a is None else np.ascontiguousarray(np.asarray(a, dtype=np.float32))
    pf = lambda a: None if a is None else a.ctypes.data_as(C.POINTER(C.c_float))
    H = f32(H); h = f32(h); A = f32(A); Alb = f32(Alb); Aub = f32(Aub); xlb = f32(xlb); xub = f32(xub); x0 = f32(x0); y0 = f32(y0)
    B, n = h.shape
    m = Alb.shape[1] if Alb.ndim == 2 else 0
    assert pivot in (PIVOT_EIGEN, PIVOT_STATIC)
    s = settings or qp_default_settings()
    x = np.zeros((B, n), dtype=np.float32); y = np.zeros((B, n + m), dtype=np.float32); info = (QPInfo * B)()
    f = lib().orc_qp_admm_solve_batch_f32 if osqp_form else lib().orc_qp_solve_batch_f32
    f.restype = None
    f(B, n, m, pf(H), pf(h), pf(A), pf(Alb), pf(Aub), pf(xlb), pf(xub), pf(x0), pf(y0), C.byref(s), pivot, pf(x), pf(y), info)
    return x, y, info


def qp_admm_solve_batch(H, h, A, Alb, Aub, xlb, xub, settings=None, pivot=PIVOT_EIGEN, x0=None, y0=None, threads=1):
    """The OSQP-style ADMM solver of admm.hpp; same array layout as qp_solve_batch."""
    H = _f(H); h = _f(h); A = _f(A); Alb = _f(Alb); Aub = _f(Aub); xlb = _f(xlb); xub = _f(xub)
    B, n = h.shape
    m = Alb.shape[1] if Alb.ndim == 2 else 0
    _check_sweep(pivot, 2 * n + m)
    s = settings or qp_default_settings()
    x = np.zeros((B, n)); y = np.zeros((B, n + m)); info = (QPInfo * B)()
    lib().orc_qp_admm_solve_batch(B, n, m, _p(H), _p(h), _p(A), _p(Alb), _p(Aub), _p(xlb), _p(xub), _p(_f(x0)), _p(_f(y0)),
                                  C.byref(s), pivot, threads, _p(x), _p(y), info)
    return x, y, info


def ruiz_compute_batch(H, h, A, Alb, Aub, xlb, xub):
    """In-place Ruiz equilibration of a batch (copies are made and returned): -> (H, h, A, Alb, Aub, xlb, xub, D, E, c)."""
    h = _f(h).copy(); B, n = h.shape
    Alb = _f(Alb).copy().reshape(B, -1); m = Alb.shape[1]
    H = _f(H).copy(); A = _f(A).copy(); Aub = _f(Aub).copy(); xlb = _f(xlb).copy(); xub = _f(xub).copy()
    D = np.zeros((B, n)); E = np.zeros((B, m)); c = np.zeros(B)
    lib().orc_ruiz_compute_batch(C.c_int(B), C.c_int(n), C.c_int(m), _p(H), _p(h), _p(A), _p(Alb), _p(Aub), _p(xlb), _p(xub), _p(D), _p(E), _p(c))
    return H, h, A, Alb, Aub, xlb, xub, D, E, c


def ruiz_unscale_solution_batch(D, E, c, x, y):
    x = _f(x).copy(); y = _f(y).copy(); D = _f(D); E = _f(E); c = _f(c)
    B, n = x.shape; m = y.shape[1] - n
    lib().orc_ruiz_unscale_solution_batch(C.c_int(B), C.c_int(n), C.c_int(m), _p(D), _p(E), _p(c), _p(x), _p(y))
    return x, y


def ocp_dims(model, P, S):
    v = [C.c_int() for _ in range(8)]
    lib().orc_ocp_dims(model, P, S, *[C.byref(a) for a in v])
    nx, nu, np_, nd, ng, n, me, mi = [a.value for a in v]
    return dict(nx=nx, nu=nu, np=np_, nd=nd, ng=ng, n=n, m_eq=me, m_ineq=mi, m=me + mi, nn=P * S + 1)


def ocp_time_nodes(model, P, S, t0, tf):
    tn = np.zeros(P * S + 1)
    f = lib().orc_ocp_time_nodes
    f.argtypes = [C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.POINTER(C.c_double)]
    f(model, P, S, t0, tf, _p(tn))
    return tn


def ocp_eval(model, P, S, t0, tf, var, d, lam=None, mparams=None):
    dm = ocp_dims(model, P, S)
    n, m = dm["n"], dm["m"]
    cost = C.c_double()
    c_eq = np.zeros(dm["m_eq"]); g = np.zeros(max(dm["m_ineq"], 1)); jac = np.zeros(m * n)
    cg = np.zeros(n); ch = np.zeros(n * n); lg = np.zeros(n); lh = np.zeros(n * n)
    mp = _f(mparams) if mparams is not None else None
    f = lib().orc_ocp_eval
    f.argtypes = [C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.POINTER(C.c_double), C.c_int] + \
                 [C.POINTER(C.c_double)] * 11
    f(model, P, S, t0, tf, _p(mp), 0 if mp is None else len(mp), _p(_f(var)), _p(_f(d)), _p(_f(lam)),
      C.byref(cost), _p(c_eq), _p(g), _p(jac), _p(cg), _p(ch), _p(lg), _p(lh))
    return dict(cost=cost.value, c=c_eq, g=g[:dm["m_ineq"]], jac=jac.reshape(n, m).T.copy(), cost_grad=cg,
                cost_hess=ch.reshape(n, n).T.copy(), lag_grad=lg, lag_hess=lh.reshape(n, n).T.copy())


def _sqp_schur_check(dm, P, ss):
    """PIVOT_SCHUR inside the SQP: the Hessian must stay block diagonal per node — block BFGS or exact Hessians, no parameters, no path constraints,
    at most one parameter (bordered form), default regularisation or the (diagonal) Gershgorin shift, no preconditioner, boxADMM."""
    ok = (ss.hessian_update == 1 or ss.exact_hessian_every_iter) and dm["np"] <= 1 and dm["ng"] == 0 and ss.regularisation in (0, 2) and \
         ss.preconditioner == 0 and ss.qp_solver == 0 and dm["m"] <= SCHUR_MAX_ROWS
    if not ok:
        raise ValueError("PIVOT_SCHUR restates the block-structured kernel: hessian_update = 1 or exact Hessians, NP <= 1, NG = 0, m <= 64, "
                         "regularisation 0 / 2, no preconditioner, boxADMM")


def _sqp_argtypes(extra_tail):
    dp = C.POINTER(C.c_double)
    return [C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, dp, C.c_int] + extra_tail


def sqp_solve_batch(model, P, S, t0, tf, B, d, lbx, ubx, lbg=None, ubg=None, x_guess=None, lam_guess=None,
                    sqp_settings=None, qp_settings=None, pivot=PIVOT_EIGEN, mparams=None, threads=1):
    dm = ocp_dims(model, P, S)
    n, m = dm["n"], dm["m"]
    _check_sweep(pivot, n + m)
    ss = sqp_settings or sqp_default_settings(); qs = qp_settings or sqp_qp_default_settings()
    if pivot == PIVOT_SCHUR:
        _sqp_schur_check(dm, P, ss)
    if pivot == PIVOT_CONDSWEEP and (dm["np"] > 1 or m > 64 or n > COND_MAX_ROWS):
        raise ValueError("PIVOT_CONDSWEEP restates the condensed register kernel: NP <= 1, n <= 112, m <= 64")
    x = np.zeros((B, n)); lam = np.zeros((B, m + n)); info = (SQPInfo * B)()
    mp = _f(mparams) if mparams is not None else None
    dp = C.POINTER(C.c_double)
    f = lib().orc_sqp_solve_batch
    f.argtypes = _sqp_argtypes([C.c_int] + [dp] * 7 + [C.POINTER(SQPSettings), C.POINTER(QPSettings), C.c_int, C.c_int,
                                                      dp, dp, C.POINTER(SQPInfo)])
    f(model, P, S, t0, tf, _p(mp), 0 if mp is None else len(mp), B, _p(_f(x_guess)), _p(_f(lam_guess)), _p(_f(d)),
      _p(_f(lbx)), _p(_f(ubx)), _p(_f(lbg)), _p(_f(ubg)), C.byref(ss), C.byref(qs), pivot, threads, _p(x), _p(lam), info)
    return x, lam, info


def sqp_trace_qps(model, P, S, t0, tf, d, lbx, ubx, lbg=None, ubg=None, x_guess=None, lam_guess=None,
                  sqp_settings=None, qp_settings=None, pivot=PIVOT_EIGEN, mparams=None, max_qps=32):
    dm = ocp_dims(model, P, S)
    n, m = dm["n"], dm["m"]
    _check_sweep(pivot, n + m)
    ss = sqp_settings or sqp_default_settings(); qs = qp_settings or sqp_qp_default_settings()
    if pivot == PIVOT_SCHUR:
        _sqp_schur_check(dm, P, ss)
    H = np.zeros((max_qps, n * n)); h = np.zeros((max_qps, n)); A = np.zeros((max_qps, m * n))
    al = np.zeros((max_qps, m)); au = np.zeros((max_qps, m)); lx = np.zeros((max_qps, n)); ux = np.zeros((max_qps, n))
    mp = _f(mparams) if mparams is not None else None
    dp = C.POINTER(C.c_double)
    f = lib().orc_sqp_trace_qps
    f.restype = C.c_int
    f.argtypes = _sqp_argtypes([dp] * 7 + [C.POINTER(SQPSettings), C.POINTER(QPSettings), C.c_int, C.c_int] + [dp] * 7)
    k = f(model, P, S, t0, tf, _p(mp), 0 if mp is None else len(mp), _p(_f(x_guess)), _p(_f(lam_guess)), _p(_f(d)),
          _p(_f(lbx)), _p(_f(ubx)), _p(_f(lbg)), _p(_f(ubg)), C.byref(ss), C.byref(qs), pivot, max_qps,
          _p(H), _p(h), _p(A), _p(al), _p(au), _p(lx), _p(ux))
    return dict(H=H[:k], h=h[:k], A=A[:k], al=al[:k], au=au[:k], lx=lx[:k], ux=ux[:k], n=n, m=m)


def nlp_solve(problem, x0, lbx=None, ubx=None, lbg=None, ubg=None, sqp_settings=None, qp_settings=None,
              pivot=PIVOT_EIGEN, lam0=None):
    dims = {NLP_CONSTRAINED_ROSENBROCK: (2, 1, 0), NLP_ROSENBROCK: (2, 0, 0), NLP_SIMPLE: (2, 0, 1), NLP_HS071: (4, 1, 1)}
    n, ne, ni = dims[problem]
    ss = sqp_settings or sqp_default_settings(); qs = qp_settings or sqp_qp_default_settings()
    x = np.zeros(n); lam = np.zeros(ne + ni + n); info = SQPInfo()
    lib().orc_nlp_solve(problem, _p(_f(x0)), _p(_f(lam0)), _p(_f(lbx)), _p(_f(ubx)), _p(_f(lbg)), _p(_f(ubg)),
                        C.byref(ss), C.byref(qs), pivot, _p(x), _p(lam), C.byref(info))
    return x, lam, info


# ------------------------------------------------------------------------------------------------------------------
# the reference's own compiled CasADi fixtures (oracle/_ref), P=5 S=2 mobile robot, 55 vars / 33 eq
class RefCasadiRobot:
    N, M = 55, 33

    def __init__(self):
        if not os.path.exists(REF_PATH):
            raise FileNotFoundError(REF_PATH)
        self.l = C.CDLL(REF_PATH)

    def _call(self, name, args, out_sizes):
        fn = getattr(self.l, name)
        work = getattr(self.l, name + "_work")
        sz = [C.c_longlong() for _ in range(4)]
        work(*[C.byref(a) for a in sz])
        n_arg, n_res, n_iw, n_w = [max(int(a.value), 1) for a in sz]
        argv = (C.POINTER(C.c_double) * n_arg)()
        for i, a in enumerate(args):
            argv[i] = _p(a)
        outs = [np.zeros(s) for s in out_sizes]
        resv = (C.POINTER(C.c_double) * n_res)()
        for i, o in enumerate(outs):
            resv[i] = _p(o)
        iw = (C.c_longlong * n_iw)(); w = (C.c_double * n_w)()
        fn(argv, resv, iw, w, None)
        return outs

    def _sparsity(self, name, idx=0):
        f = getattr(self.l, name + "_sparsity_out"); f.restype = C.POINTER(C.c_longlong); f.argtypes = [C.c_longlong]
        sp = f(idx)
        nrow, ncol = sp[0], sp[1]
        colind = [sp[2 + i] for i in range(ncol + 1)]
        nnz = colind[-1]
        rows = [sp[2 + ncol + 1 + i] for i in range(nnz)]
        return nrow, ncol, colind, rows

    def _dense(self, name, vals):
        nrow, ncol, colind, rows = self._sparsity(name)
        Md = np.zeros((nrow, ncol))
        for j in range(ncol):
            for k in range(colind[j], colind[j + 1]):
                Md[rows[k], j] = vals[k]
        return Md

    def cost(self, x):
        return self._call("fcost", [_f(x)], [1])[0][0]

    def constraint(self, x):
        return self._call("fconstraint", [_f(x)], [self.M])[0]

    def nnz(self, name):
        return self._sparsity(name)[2][-1]
