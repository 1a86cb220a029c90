"""Developer tool (not a test): larger-batch parity of the LDS-resident SQP kernels against the static-order CPU restatement on the
reference's own grids (11 and 16 nodes) and on config B. Run on a GPU box:  python tests/tools_soak_lds.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import polympc_amd as pa
from polympc_amd import workloads
from oracle import binding as o
ctx = pa.Context(0)
for name, wl, B in (("robot 5x2", workloads.robot_batch(768, P=5, S=2), 768), ("robot 5x3", workloads.robot_batch(384, P=5, S=3), 384), ("cstr", workloads.cstr_batch(384), 384)):
    ss = pa.sqp_settings_default(); ss.max_iter = wl["max_iter"]; ss.line_search_max_iter = wl["ls_max_iter"]
    x, lam, info = ctx.sqp_solve_batch(wl["model"], wl["P"], wl["S"], wl["t0"], wl["tf"], B, wl["d"], wl["lbx"], wl["ubx"], sqp_settings=ss)
    oss = o.sqp_default_settings(); oss.max_iter = wl["max_iter"]; oss.line_search_max_iter = wl["ls_max_iter"]
    xo, lo, io = o.sqp_solve_batch(wl["model"], wl["P"], wl["S"], wl["t0"], wl["tf"], B, wl["d"], wl["lbx"], wl["ubx"], sqp_settings=oss, pivot=o.PIVOT_STATIC, threads=64)
    same = (info["iter"] == np.array([i.iter for i in io])) & (info["qp_solver_iter"] == np.array([i.qp_solver_iter for i in io]))
    sc = np.maximum(1.0, np.abs(xo))
    print(name, "B", B, "same SQP+ADMM iteration counts", same.mean(), "max rel dx on same", (np.abs(x - xo) / sc)[same].max())
