"""ctypes binding of include/polympc_amd.h (the drop-in C ABI). Plumbing only — every numeric operation happens in
the HIP library. Fails loudly if the library has not been built (no fallback path exists)."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_PATH = os.environ.get("PMPC_LIB") or os.path.join(HERE, "libpolympc_amd.so")   # PMPC_LIB: developer switch for A/B builds of the library

MODEL_ROBOT, MODEL_CSTR, MODEL_PARKING, MODEL_ROBOT_NG, MODEL_KITE_STANDIN, MODEL_PARKING_NG = 0, 1, 2, 3, 4, 5
QP_SOLVED, QP_MAX_ITER_EXCEEDED, QP_UNSOLVED = 0, 1, 2
SQP_SOLVED, SQP_MAX_ITER_EXCEEDED = 0, 1
FLAG_NONFINITE = 1   # pmpc_qp_info / pmpc_sqp_info flags: a non-finite value went through a QP solve
FLAG_ILLCOND = 2     # the conditioning gate of the constraint-first / condensed kernels tripped: solved in the full KKT form (information)

ABI_VERSION = 4   # PMPC_ABI_VERSION of include/polympc_amd.h that the ctypes layouts below mirror
ROUTE_NONE, ROUTE_REG1, ROUTE_REG2, ROUTE_LDS, ROUTE_HBM, ROUTE_SCHUR, ROUTE_CONDREG = 0, 1, 2, 3, 4, 5, 6   # pmpc_route
ROUTE_NAMES = {0: "none", 1: "reg1", 2: "reg2", 3: "lds", 4: "hbm", 5: "schur", 6: "condreg"}

EXPORTED_SYMBOLS = [
    "pmpc_abi_version", "pmpc_struct_size", "pmpc_sqp_last_route",
    "pmpc_version", "pmpc_status_string", "pmpc_create", "pmpc_destroy", "pmpc_synchronize", "pmpc_debug_phase_cycles", "pmpc_debug_set_poison",
    "pmpc_qp_settings_default", "pmpc_qp_settings_sqp_default", "pmpc_sqp_settings_default", "pmpc_chebyshev",
    "pmpc_qp_boxadmm_solve_batch", "pmpc_qp_boxadmm_solve_batch_dev", "pmpc_qp_boxadmm_solve_batch_f32", "pmpc_qp_boxadmm_solve_batch_f32_dev", "pmpc_qp_admm_solve_batch_f32", "pmpc_qp_admm_solve_batch_f32_dev", "pmpc_ocp_dims", "pmpc_ocp_linearise_batch",
    "pmpc_sqp_solve_batch", "pmpc_sqp_solve_batch_dev", "pmpc_sqp_solve_batch_user", "pmpc_sqp_solve_batch_multi",
    "pmpc_mpc_step_batch_dev", "pmpc_mpc_batch_create", "pmpc_mpc_batch_step", "pmpc_mpc_batch_solution", "pmpc_mpc_batch_destroy",
    "pmpc_qp_admm_solve_batch", "pmpc_qp_admm_solve_batch_dev", "pmpc_qp_ruiz_compute_batch", "pmpc_qp_ruiz_compute_batch_dev", "pmpc_qp_ruiz_unscale_batch", "pmpc_qp_ruiz_unscale_batch_dev",
    "pmpc_filter_state_create", "pmpc_filter_state_clear", "pmpc_filter_state_download", "pmpc_filter_state_destroy",
    "pmpc_iteration_trace_create", "pmpc_iteration_trace_clear", "pmpc_iteration_trace_download", "pmpc_iteration_trace_destroy",
]


class QPSettings(C.Structure):
    _fields_ = [("eps_rel", C.c_double), ("eps_abs", C.c_double), ("max_iter", C.c_int), ("rho", C.c_double),
                ("sigma", C.c_double), ("alpha", C.c_double), ("check_termination", C.c_int),
                ("adaptive_rho", C.c_int), ("adaptive_rho_tolerance", C.c_double), ("adaptive_rho_interval", C.c_int),
                ("linear_solver", C.c_int)]


class QPInfo(C.Structure):
    _fields_ = [("status", C.c_int), ("iter", C.c_int), ("rho_updates", C.c_int), ("flags", C.c_int),
                ("rho_estimate", C.c_double), ("res_prim", C.c_double), ("res_dual", C.c_double)]


class SQPSettings(C.Structure):
    _fields_ = [("tau", C.c_double), ("eta", C.c_double), ("rho", C.c_double), ("eps_prim", C.c_double),
                ("eps_dual", C.c_double), ("max_iter", C.c_int), ("line_search_max_iter", C.c_int),
                ("regularisation", C.c_int), ("exact_hessian_every_iter", C.c_int), ("preconditioner", C.c_int), ("hessian_update", C.c_int), ("qp_solver", C.c_int),
                ("line_search", C.c_int), ("filter_max_depth", C.c_int), ("filter_beta", C.c_double), ("filter_state", C.c_void_p),
                ("iteration_trace", C.c_void_p), ("iteration_trace_capacity", C.c_int), ("kkt_form", C.c_int)]


FILTER_MAX_DEPTH = 10
FILTER_STATE_DOUBLES = 1 + 2 * FILTER_MAX_DEPTH
TRACE_DOUBLES = 8   # [iter, alpha, primal_norm, dual_norm, cost, qp iterations, qp status, max violation]


class SQPInfo(C.Structure):
    _fields_ = [("iter", C.c_int), ("qp_solver_iter", C.c_int), ("status", C.c_int), ("flags", C.c_int),
                ("primal_norm", C.c_double), ("dual_norm", C.c_double), ("max_violation", C.c_double),
                ("cost", C.c_double)]


QP_INFO_DTYPE = np.dtype([("status", "i4"), ("iter", "i4"), ("rho_updates", "i4"), ("flags", "i4"),
                          ("rho_estimate", "f8"), ("res_prim", "f8"), ("res_dual", "f8")])
SQP_INFO_DTYPE = np.dtype([("iter", "i4"), ("qp_solver_iter", "i4"), ("status", "i4"), ("flags", "i4"),
                           ("primal_norm", "f8"), ("dual_norm", "f8"), ("max_violation", "f8"), ("cost", "f8")])
assert QP_INFO_DTYPE.itemsize == C.sizeof(QPInfo) == 40 and SQP_INFO_DTYPE.itemsize == C.sizeof(SQPInfo) == 48

HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC"]
BUILD_DIR = os.path.join(HERE, "_build")


def build_library(force=False, verbose=False, jobs=None):
    """Cross-compile the HIP library for gfx950 in-tree (works without a GPU): one object per translation unit
    (pmpc_api.hip + one pmpc_model_*.hip per built-in OCP), compiled in parallel, then linked."""
    from concurrent.futures import ThreadPoolExecutor
    files = sorted(os.listdir(CSRC))
    units = sorted((f for f in files if f.endswith(".hip")), key=lambda f: (not f.startswith("pmpc_model_"), not f.startswith("pmpc_grids_"), f))   # longest translation units first
    deps = [os.path.join(CSRC, f) for f in files if f.endswith(".hpp")] + [os.path.join(HERE, "..", "include", "polympc_amd.h")]
    newest_hdr = max(os.path.getmtime(d) for d in deps)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    extra = os.environ.get("PMPC_EXTRA_HIPCC_FLAGS", "").split()
    os.makedirs(BUILD_DIR, exist_ok=True)
    todo, objs = [], []
    for u in units:
        src, obj = os.path.join(CSRC, u), os.path.join(BUILD_DIR, u[:-4] + ".o")
        objs.append(obj)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(newest_hdr, os.path.getmtime(src)):
            todo.append([hipcc] + HIPCC_FLAGS + extra + ["-c", "-o", obj, src])
    if not todo and os.path.exists(LIB_PATH) and os.path.getmtime(LIB_PATH) >= max(os.path.getmtime(o) for o in objs):
        return LIB_PATH

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    with ThreadPoolExecutor(max_workers=jobs or min(8, os.cpu_count() or 1)) as ex:
        list(ex.map(run, todo))
    run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH] + objs)
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'`. "
                "polympc_amd has no CPU fallback.")
        L = C.CDLL(LIB_PATH)
        L.pmpc_version.restype = C.c_char_p
        L.pmpc_status_string.restype = C.c_char_p
        # a stale library next to new ctypes layouts (or the reverse) would have pmpc_*_settings_default() write past a struct, or hand the
        # kernels garbage in the fields it does not know: refuse it here
        if not hasattr(L, "pmpc_abi_version"):
            raise RuntimeError(f"{LIB_PATH} predates the ABI version query: rebuild it (python -c 'import __graft_entry__ as g; g.build()')")
        L.pmpc_struct_size.restype = C.c_ulong
        got = L.pmpc_abi_version()
        sizes = [int(L.pmpc_struct_size(i)) for i in range(4)]
        want = [C.sizeof(QPSettings), C.sizeof(QPInfo), C.sizeof(SQPSettings), C.sizeof(SQPInfo)]
        if (got != ABI_VERSION and not os.environ.get("PMPC_ABI_ANY")) or sizes != want:   # PMPC_ABI_ANY: developer switch for timing an older build (PMPC_LIB) whose struct sizes agree
            raise RuntimeError(f"{LIB_PATH}: ABI version {got} / struct sizes {sizes}, this binding expects version {ABI_VERSION} / {want}: "
                               "rebuild the library from the same tree")
        _lib = L
    return _lib


def _check(st):
    if st != 0:
        raise RuntimeError("polympc_amd: " + lib().pmpc_status_string(st).decode())


def qp_settings_default():
    s = QPSettings(); lib().pmpc_qp_settings_default(C.byref(s)); return s


def qp_settings_sqp_default():
    s = QPSettings(); lib().pmpc_qp_settings_sqp_default(C.byref(s)); return s


def sqp_settings_default():
    s = SQPSettings(); lib().pmpc_sqp_settings_default(C.byref(s)); return s


def chebyshev(P):
    nodes = np.zeros(P + 1); w = np.zeros(P + 1); D = np.zeros((P + 1) * (P + 1))
    dp = C.POINTER(C.c_double)
    _check(lib().pmpc_chebyshev(P, nodes.ctypes.data_as(dp), w.ctypes.data_as(dp), D.ctypes.data_as(dp)))
    return nodes, w, D.reshape(P + 1, P + 1).T.copy()


def ocp_dims(model, P, S):
    v = [C.c_int() for _ in range(8)]
    _check(lib().pmpc_ocp_dims(model, P, S, *[C.byref(a) for a in v]))
    nx, nu, np_, nd, ng, n, me, mi = [a.value for a in v]
    return dict(nx=nx, nu=nu, np=np_, nd=nd, ng=ng, n=n, m_eq=me, m_ineq=mi, m=me + mi, nn=P * S + 1)


def _h(a):
    """host array -> (keepalive, pointer)"""
    if a is None:
        return None, None
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(C.POINTER(C.c_double))


def _d(t):
    """torch CUDA tensor -> device pointer"""
    if t is None:
        return None
    assert t.is_cuda and t.is_contiguous()
    return C.cast(C.c_void_p(t.data_ptr()), C.POINTER(C.c_double))


class Context:
    """pmpc_context: one per host thread / GPU."""

    def __init__(self, device=0, stream=None):
        self._ctx = C.c_void_p()
        _check(lib().pmpc_create(int(device), C.c_void_p(stream) if stream else None, C.byref(self._ctx)))

    def close(self):
        if self._ctx:
            lib().pmpc_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def synchronize(self):
        _check(lib().pmpc_synchronize(self._ctx))

    def set_poison(self, on=True):
        """pmpc_debug_set_poison (developer harness, also PMPC_POISON=1): signalling NaNs into the HBM workspace, the staging buffers, every CU's LDS
        and every SIMD's register file before each launch of this context — an uninitialised read then returns NaN."""
        _check(lib().pmpc_debug_set_poison(self._ctx, 1 if on else 0))

    def last_route(self):
        """pmpc_sqp_last_route: the kernel family that served this context's last fused SQP call (ROUTE_* / ROUTE_NAMES)."""
        return int(lib().pmpc_sqp_last_route(self._ctx))

    def phase_cycles(self, reset=True):
        out = (C.c_ulonglong * 24)()
        _check(lib().pmpc_debug_phase_cycles(self._ctx, out, 1 if reset else 0))
        return list(out)

    # ------------------------------------------------------------------ QP, host buffers
    def qp_admm_solve_batch(self, H, h, A, Alb, Aub, xlb, xub, settings=None, x0=None, y0=None):
        """The OSQP-style ADMM solver (admm.hpp); same layout as qp_solve_batch."""
        return self.qp_solve_batch(H, h, A, Alb, Aub, xlb, xub, settings=settings, x0=x0, y0=y0, _entry="pmpc_qp_admm_solve_batch")

    def qp_solve_batch(self, H, h, A, Alb, Aub, xlb, xub, settings=None, x0=None, y0=None, _entry="pmpc_qp_boxadmm_solve_batch"):
        hk, hp = _h(h); B, n = hk.shape
        Hk, Hp = _h(H); Ak, Ap = _h(A); albk, albp = _h(Alb); aubk, aubp = _h(Aub); xlk, xlp = _h(xlb); xuk, xup = _h(xub)
        m = albk.shape[1] if albk.ndim == 2 else 0
        x0k, x0p = _h(x0); y0k, y0p = _h(y0)
        s = settings or qp_settings_default()
        x = np.zeros((B, n)); y = np.zeros((B, n + m)); info = np.zeros(B, dtype=QP_INFO_DTYPE)
        _check(getattr(lib(), _entry)(self._ctx, B, n, m, Hp, hp, Ap, albp, aubp, xlp, xup, x0p, y0p, C.byref(s),
                                                 x.ctypes.data_as(C.POINTER(C.c_double)),
                                                 y.ctypes.data_as(C.POINTER(C.c_double)), C.c_void_p(info.ctypes.data)))
        return x, y, info

    def qp_solve_batch_f32(self, H, h, A, Alb, Aub, xlb, xub, settings=None, x0=None, y0=None, osqp_form=False):
        """boxADMM<N, M, float> (pmpc_qp_boxadmm_solve_batch_f32; osqp_form: ADMM<N, M, float>, pmpc_qp_admm_solve_batch_f32): float32 host arrays in
        the layout of qp_solve_batch."""
        f32 = lambda a: None if a is None else np.ascontiguousarray(np.asarray(a, dtype=np.float32))
        pf = lambda a: None if a is None else a.ctypes.data_as(C.POINTER(C.c_float))
        H, h, A, Alb, Aub, xlb, xub, x0, y0 = (f32(a) for a in (H, h, A, Alb, Aub, xlb, xub, x0, y0))
        B, n = h.shape
        m = Alb.shape[1] if Alb.ndim == 2 else 0
        s = settings or qp_settings_default()
        x = np.zeros((B, n), dtype=np.float32); y = np.zeros((B, n + m), dtype=np.float32); info = np.zeros(B, dtype=QP_INFO_DTYPE)
        f = lib().pmpc_qp_admm_solve_batch_f32 if osqp_form else lib().pmpc_qp_boxadmm_solve_batch_f32
        f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int] + [C.POINTER(C.c_float)] * 9 + [C.POINTER(QPSettings)] + [C.POINTER(C.c_float)] * 2 + [C.c_void_p]
        _check(f(self._ctx, B, n, m, pf(H), pf(h), pf(A), pf(Alb), pf(Aub), pf(xlb), pf(xub), pf(x0), pf(y0), C.byref(s), pf(x), pf(y),
                 C.c_void_p(info.ctypes.data)))
        return x, y, info

    # ------------------------------------------------------------------ QP, device buffers (torch tensors), asynchronous
    def qp_ruiz_compute_batch(self, H, h, A, Alb, Aub, xlb, xub):
        """RuizEquilibration::compute on copies of the host arrays -> (H, h, A, Alb, Aub, xlb, xub, D, E, c)."""
        P_ = C.POINTER(C.c_double)
        h = np.array(h, dtype=np.float64, order="C"); B, n = h.shape
        Alb = np.array(Alb, dtype=np.float64, order="C").reshape(B, -1); m = Alb.shape[1]
        H, A, Aub, xlb, xub = (np.array(a, dtype=np.float64, order="C") for a in (H, A, Aub, xlb, xub))
        D = np.zeros((B, n)); E = np.zeros((B, max(m, 1))); c = np.zeros(B)
        _check(lib().pmpc_qp_ruiz_compute_batch(self._ctx, B, n, m, *[a.ctypes.data_as(P_) for a in (H, h, A, Alb, Aub, xlb, xub, D, E, c)]))
        return H, h, A, Alb, Aub, xlb, xub, D, E[:, :m], c

    def qp_ruiz_unscale_batch(self, D, E, c, x, y):
        P_ = C.POINTER(C.c_double)
        x = np.array(x, dtype=np.float64, order="C"); y = np.array(y, dtype=np.float64, order="C")
        D, E, c = (np.ascontiguousarray(a, dtype=np.float64) for a in (D, E, c))
        B, n = x.shape; m = y.shape[1] - n
        Ep = np.ascontiguousarray(E if m > 0 else np.zeros((B, 1)))
        _check(lib().pmpc_qp_ruiz_unscale_batch(self._ctx, B, n, m, *[a.ctypes.data_as(P_) for a in (D, Ep, c, x, y)]))
        return x, y

    def qp_solve_batch_dev(self, B, n, m, H, h, A, Alb, Aub, xlb, xub, x, y, info, settings, x0=None, y0=None):
        _check(lib().pmpc_qp_boxadmm_solve_batch_dev(self._ctx, B, n, m, _d(H), _d(h), _d(A), _d(Alb), _d(Aub), _d(xlb), _d(xub),
                                                     _d(x0), _d(y0), C.byref(settings), _d(x), _d(y), C.c_void_p(info.data_ptr())))

    # ------------------------------------------------------------------ collocation assembly (host buffers)
    def ocp_linearise_batch(self, model, P, S, t0, tf, var, d, lam=None, mparams=None):
        dm = ocp_dims(model, P, S); n, m = dm["n"], dm["m"]
        vk, vp = _h(var); B = vk.shape[0]
        dk, dp_ = _h(d); lk, lp = _h(lam); mk, mp = _h(mparams)
        cost = np.zeros((B, 2)); c = np.zeros((B, m)); jac = np.zeros((B, m * n)); cg = np.zeros((B, n)); lg = np.zeros((B, n))
        lh = np.zeros((B, n * n))
        P_ = C.POINTER(C.c_double)
        f = lib().pmpc_ocp_linearise_batch
        f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, P_, C.c_int, C.c_int] + [P_] * 9
        _check(f(self._ctx, model, P, S, t0, tf, mp, 0 if mk is None else len(mk), B, vp, dp_, lp,
                 *[a.ctypes.data_as(P_) for a in (cost, c, jac, cg, lg, lh)]))
        return dict(cost=cost[:, 0], cost_values_only=cost[:, 1], c=c, jac=jac.reshape(B, n, m).transpose(0, 2, 1).copy(),
                    cost_grad=cg, lag_grad=lg, lag_hess=lh.reshape(B, n, n).transpose(0, 2, 1).copy())

    # ------------------------------------------------------------------ SQP, host buffers
    def sqp_solve_batch(self, model, P, S, t0, tf, B, d, lbx, ubx, lbg=None, ubg=None, x_guess=None, lam_guess=None,
                        sqp_settings=None, qp_settings=None, mparams=None):
        dm = ocp_dims(model, P, S); n, m = dm["n"], dm["m"]
        ss = sqp_settings or sqp_settings_default(); qs = qp_settings or qp_settings_sqp_default()
        x = np.zeros((B, n)); lam = np.zeros((B, m + n)); info = np.zeros(B, dtype=SQP_INFO_DTYPE)
        keep = [_h(a) for a in (mparams, x_guess, lam_guess, d, lbx, ubx, lbg, ubg)]
        P_ = C.POINTER(C.c_double)
        f = lib().pmpc_sqp_solve_batch
        f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, P_, C.c_int, C.c_int] + [P_] * 7 + \
                     [C.POINTER(SQPSettings), C.POINTER(QPSettings), P_, P_, C.c_void_p]
        _check(f(self._ctx, model, P, S, t0, tf, keep[0][1], 0 if keep[0][0] is None else len(keep[0][0]), B,
                 *[k[1] for k in keep[1:]], C.byref(ss), C.byref(qs), x.ctypes.data_as(P_), lam.ctypes.data_as(P_),
                 C.c_void_p(info.ctypes.data)))
        return x, lam, info

    @staticmethod
    def sqp_solve_batch_multi(ctxs, model, P, S, t0, tf, B, d, lbx, ubx, lbg=None, ubg=None, x_guess=None, lam_guess=None,
                              sqp_settings=None, qp_settings=None, mparams=None):
        """pmpc_sqp_solve_batch_multi: the batch in contiguous shards over the given contexts, one host thread per context."""
        dm = ocp_dims(model, P, S); n, m = dm["n"], dm["m"]
        ss = sqp_settings or sqp_settings_default(); qs = qp_settings or qp_settings_sqp_default()
        x = np.zeros((B, n)); lam = np.zeros((B, m + n)); info = np.zeros(B, dtype=SQP_INFO_DTYPE)
        keep = [_h(a) for a in (mparams, x_guess, lam_guess, d, lbx, ubx, lbg, ubg)]
        P_ = C.POINTER(C.c_double)
        arr = (C.c_void_p * len(ctxs))(*[c._ctx for c in ctxs])
        f = lib().pmpc_sqp_solve_batch_multi
        f.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, P_, C.c_int, C.c_int] + [P_] * 7 + \
                     [C.POINTER(SQPSettings), C.POINTER(QPSettings), P_, P_, C.c_void_p]
        _check(f(arr, len(ctxs), model, P, S, t0, tf, keep[0][1], 0 if keep[0][0] is None else len(keep[0][0]), B,
                 *[k[1] for k in keep[1:]], C.byref(ss), C.byref(qs), x.ctypes.data_as(P_), lam.ctypes.data_as(P_),
                 C.c_void_p(info.ctypes.data)))
        return x, lam, info

    # ------------------------------------------------------------------ LSFilter state of B solver objects (line_search = 1)
    def filter_state_create(self, B):
        """Device buffer of B empty filters; put the returned handle into SQPSettings.filter_state to carry the filter between solves."""
        out = C.c_void_p()
        f = lib().pmpc_filter_state_create
        f.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]
        _check(f(self._ctx, B, C.byref(out)))
        return out.value

    def filter_state_clear(self, B, handle):
        f = lib().pmpc_filter_state_clear
        f.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        _check(f(self._ctx, B, handle))

    def filter_state_download(self, B, handle):
        out = np.zeros((B, FILTER_STATE_DOUBLES))
        f = lib().pmpc_filter_state_download
        f.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_double)]
        _check(f(self._ctx, B, handle, out.ctypes.data_as(C.POINTER(C.c_double))))
        return out

    def filter_state_destroy(self, handle):
        f = lib().pmpc_filter_state_destroy
        f.argtypes = [C.c_void_p, C.c_void_p]
        _check(f(self._ctx, handle))

    # ------------------------------------------------------------------ per-iteration records (the reference's iteration_callback, recorded)
    def iteration_trace_create(self, B, capacity):
        """Device buffer of B x capacity zeroed records; put the handle into SQPSettings.iteration_trace (and the capacity next to it)."""
        out = C.c_void_p()
        f = lib().pmpc_iteration_trace_create
        f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        _check(f(self._ctx, B, capacity, C.byref(out)))
        return out.value

    def iteration_trace_clear(self, B, capacity, handle):
        f = lib().pmpc_iteration_trace_clear
        f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        _check(f(self._ctx, B, capacity, handle))

    def iteration_trace_download(self, B, capacity, handle):
        out = np.zeros((B, capacity, TRACE_DOUBLES))
        f = lib().pmpc_iteration_trace_download
        f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_double)]
        _check(f(self._ctx, B, capacity, handle, out.ctypes.data_as(C.POINTER(C.c_double))))
        return out

    def iteration_trace_destroy(self, handle):
        f = lib().pmpc_iteration_trace_destroy
        f.argtypes = [C.c_void_p, C.c_void_p]
        _check(f(self._ctx, handle))

    # ------------------------------------------------------------------ MPC step, device buffers (torch tensors), asynchronous
    def mpc_step_batch_dev(self, model, P, S, t0, tf, B, x0, d, lbx, ubx, x, lam, info, sqp_settings, qp_settings, u0=None, lbg=None,
                           ubg=None, mparams=None):
        mk, mp = _h(mparams)
        P_ = C.POINTER(C.c_double)
        f = lib().pmpc_mpc_step_batch_dev
        f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, P_, C.c_int, C.c_int] + [P_] * 6 + \
                     [C.POINTER(SQPSettings), C.POINTER(QPSettings), P_, P_, C.c_void_p, P_]
        _check(f(self._ctx, model, P, S, t0, tf, mp, 0 if mk is None else len(mk), B, _d(x0), _d(d), _d(lbx), _d(ubx), _d(lbg), _d(ubg),
                 C.byref(sqp_settings), C.byref(qp_settings), _d(x), _d(lam), C.c_void_p(info.data_ptr()), _d(u0)))

    # ------------------------------------------------------------------ SQP, device buffers (torch tensors), asynchronous
    def sqp_solve_batch_dev(self, model, P, S, t0, tf, B, d, lbx, ubx, x, lam, info, sqp_settings, qp_settings, lbg=None,
                            ubg=None, x_guess=None, lam_guess=None, mparams=None):
        mk, mp = _h(mparams)
        P_ = C.POINTER(C.c_double)
        f = lib().pmpc_sqp_solve_batch_dev
        f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, P_, C.c_int, C.c_int] + [P_] * 7 + \
                     [C.POINTER(SQPSettings), C.POINTER(QPSettings), P_, P_, C.c_void_p]
        _check(f(self._ctx, model, P, S, t0, tf, mp, 0 if mk is None else len(mk), B, _d(x_guess), _d(lam_guess), _d(d),
                 _d(lbx), _d(ubx), _d(lbg), _d(ubg), C.byref(sqp_settings), C.byref(qp_settings), _d(x), _d(lam),
                 C.c_void_p(info.data_ptr())))
