"""Developer tool (not a test): per-phase shader-clock breakdown of the fused SQP kernel on config A.
Run on a GPU box:  PMPC_PHASE_PROFILE=1 python tests/tools_phase_profile.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import polympc_amd as pa
from polympc_amd import workloads

B = int(os.environ.get("B", 4096))
P_, S_ = int(os.environ.get("P", 6)), int(os.environ.get("S", 1))   # P=5 S=2 (88 KKT rows) or PMPC_FORCE_LDS_PATH=1: the LDS-resident kernel
_cfg = os.environ.get("CFG")   # CFG=B: config B (CSTR, 110 KKT rows); CFG=C: the kite stand-in (464 KKT rows)
wl = workloads.cstr_batch(B) if _cfg == "B" else (workloads.kite_standin_batch(B) if _cfg == "C" else workloads.robot_batch(B, P=P_, S=S_))
ctx = pa.Context(0)
ss = pa.sqp_settings_default(); ss.max_iter = wl["max_iter"]; ss.line_search_max_iter = wl["ls_max_iter"]; ss.hessian_update = int(os.environ.get("HESSIAN_UPDATE", 0))
for rep in range(2):
    t = time.perf_counter()
    x, lam, info = ctx.sqp_solve_batch(wl["model"], wl["P"], wl["S"], wl["t0"], wl["tf"], B, wl["d"], wl["lbx"], wl["ubx"], sqp_settings=ss)
    t = time.perf_counter() - t
cyc = ctx.phase_cycles()
names = ["linearise(+update)", "QP", "line search", "termination", "total loop", "BFGS", "KKT build+factor", "QP residuals",
         "ls node evaluation", "ls scalar sums", "first-order staging", "second-order staging", "first-order assembly",
         "Hessian assembly", "Lagrangian gradient", "QP: ADMM updates (one-row-per-lane kernels)", "inv: row loads + staging | LDS path: KKT build", "inv: panel moves | LDS path: substitutions", "inv: sweeps", "inv: MFMA updates",
         "inv: final conversion", "ls prologue", "ls acceptance", "QP: prologue (one-row-per-lane kernels)"]
qps = info["iter"].sum()
print(f"host wall {t*1e3:.2f} ms (incl. copies), {qps} QPs, {info['qp_solver_iter'].sum()/qps:.2f} ADMM it/QP")
for nme, c in zip(names, cyc):
    if nme != "-":
        print(f"{nme:>20s}: {c/2/qps:12.0f} cycles per SQP iteration  ({100*c/max(cyc[4],1):5.1f}%)")
