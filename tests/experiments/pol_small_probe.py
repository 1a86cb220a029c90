"""Developer probe (GPU): the hook build (POL = true) of the small condensed kernel — robot (5, 2), 55 + 33 — against its CPU restatement
(PIVOT_CONDSWEEP). Run with PMPC_LIB pointing at a library built with -DPMPC_EXPERIMENT_SMALL_POL=2 (the hook build under the default policies):
prints, per run, which instances / entries differ, so that a pattern (lanes, slots, first iteration) can be read off; PMPC_POISON=1 turns an
uninitialised read into NaN."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import polympc_amd as pa
from polympc_amd import workloads
from oracle import binding as ob

B = int(os.environ.get("B", 8)); MAXIT = int(os.environ.get("MAXIT", 1))
P, S = int(os.environ.get("P", 5)), int(os.environ.get("S", 2))
wl = workloads.robot_batch(B, P=P, S=S)
ctx = pa.Context(0)
ss = pa.sqp_settings_default(); ss.max_iter = MAXIT; ss.line_search_max_iter = wl["ls_max_iter"]
for k in ("line_search", "hessian_update", "preconditioner", "kkt_form"):
    if k.upper() in os.environ: setattr(ss, k, int(os.environ[k.upper()]))
oss = ob.sqp_default_settings(); oss.max_iter = MAXIT; oss.line_search_max_iter = wl["ls_max_iter"]
for k in ("line_search", "hessian_update"):
    if k.upper() in os.environ: setattr(oss, k, int(os.environ[k.upper()]))
otrace = np.zeros((B, MAXIT, 8)); ob.bind_iteration_trace(oss, otrace)
xo, lo, io = ob.sqp_solve_batch(0, P, S, wl["t0"], wl["tf"], B, wl["d"], wl["lbx"], wl["ubx"], sqp_settings=oss, pivot=ob.PIVOT_CONDSWEEP)
prev = None
th = ctx.iteration_trace_create(B, MAXIT); ss.iteration_trace = th; ss.iteration_trace_capacity = MAXIT
for run in range(int(os.environ.get("RUNS", 3))):
    x, lam, info = ctx.sqp_solve_batch(wl["model"], P, S, wl["t0"], wl["tf"], B, wl["d"], wl["lbx"], wl["ubx"], sqp_settings=ss)
    route = pa.capi.ROUTE_NAMES.get(ctx.last_route())
    dx = np.abs(x - xo); dl = np.abs(lam - lo)
    bad = np.argwhere(~(dx == 0))
    print(f"run {run}: route {route}  x bit-identical {np.array_equal(x, xo)}  lam bit-identical {np.array_equal(lam, lo)}  nan x {int(np.isnan(x).sum())} nan lam {int(np.isnan(lam).sum())}  "
          f"max|dx| {np.nanmax(dx):.3e}  qp iters gpu {info['qp_solver_iter'].tolist()} cpu {[i.qp_solver_iter for i in io]}  flags {info['flags'].tolist()}", flush=True)
    if bad.size:
        inst = sorted(set(bad[:, 0].tolist())); cols = sorted(set(bad[:, 1].tolist()))
        print(f"   instances with differences {inst[:16]}  columns {cols}", flush=True)
        b0 = inst[0]
        print("   x gpu ", np.array2string(x[b0], precision=4, max_line_width=250))
        print("   x cpu ", np.array2string(xo[b0], precision=4, max_line_width=250))
        badl = np.argwhere(~(dl[b0] == 0)).ravel().tolist()
        print("   lam entries differing on that instance:", badl)
    tr = ctx.iteration_trace_download(B, MAXIT, th)
    print("   trace gpu [iter, alpha, primal_norm, dual_norm, cost, qp it, qp status, max viol]:", np.array2string(np.asarray(tr)[0, 0], precision=6))
    print("   trace cpu                                                                     :", np.array2string(otrace[0, 0], precision=6))
    ctx.iteration_trace_clear(B, MAXIT, th)
    if prev is not None:
        print("   same bits as the previous run:", np.array_equal(x, prev, equal_nan=True))
    prev = x.copy()
ctx.close()
