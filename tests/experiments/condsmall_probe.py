"""Developer probe (GPU; PMPC_LIB = a library built with -DPMPC_EXPERIMENT_COND_SMALL=1): config A on the condensed register QP (35 instead of 56 rows) — bit-identity
against PIVOT_CONDSWEEP, route, and timing beside the shipped route."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import polympc_amd as pa
from polympc_amd import workloads
from oracle import binding as ob
B = 256
wl = workloads.robot_batch(B)
ctx = pa.Context(0)
ss = pa.sqp_settings_default(); ss.max_iter = wl["max_iter"]; ss.line_search_max_iter = wl["ls_max_iter"]
x, lam, info = ctx.sqp_solve_batch(wl["model"], wl["P"], wl["S"], wl["t0"], wl["tf"], B, wl["d"], wl["lbx"], wl["ubx"], sqp_settings=ss)
print("route", pa.capi.ROUTE_NAMES.get(ctx.last_route()))
oss = ob.sqp_default_settings(); oss.max_iter = wl["max_iter"]; oss.line_search_max_iter = wl["ls_max_iter"]
for name, piv in (("CONDSWEEP", ob.PIVOT_CONDSWEEP), ("SWEEP", ob.PIVOT_SWEEP)):
    xo, lo, io = ob.sqp_solve_batch(ob.MODEL_ROBOT, wl["P"], wl["S"], wl["t0"], wl["tf"], B, wl["d"], wl["lbx"], wl["ubx"], sqp_settings=oss, pivot=piv, threads=8)
    same = [(int(a), int(b), int(c)) for a, b, c in zip(info["iter"], info["status"], info["qp_solver_iter"])] == [(i.iter, i.status, i.qp_solver_iter) for i in io]
    print(name, "bit-identical x", np.array_equal(x, xo), "lam", np.array_equal(lam, lo), "same counts", same, "max|dx| %.3e" % np.abs(x - xo).max(), "nan", int(np.isnan(x).sum()))
ctx.close()
