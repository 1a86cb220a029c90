"""Developer tool (GPU): how identical are the HIP path and the CPU restatement, configuration by configuration?

    python tests/tools_parity_probe.py [--big]

Prints, per configuration: fraction of instances with identical (iter, status, qp_solver_iter), whether x / lam are BIT-identical,
and the max abs differences of x, lam and the reported KKT quantities. This is what decides the tolerances written in
tests/test_gpu_parity.py (the product is never compared against anything looser than what is measured here)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import polympc_amd as pa                     # noqa: E402
from polympc_amd import workloads            # noqa: E402
from oracle import binding as ob             # noqa: E402


def order_for(n, m, nodes, kw):
    if kw.get("qp_solver", 0):
        return ob.PIVOT_STATIC
    if kw.get("preconditioner", 0) or kw.get("line_search", 0):
        return (ob.PIVOT_BLOCKED if kw.get("preconditioner", 0) else ob.PIVOT_CONDENSED) if n + m >= 96 else ob.PIVOT_STATIC
    reg = (3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16)
    if n + m <= 64 and nodes in ((5, 7) if kw.get("hessian_update", 0) else reg):
        return ob.PIVOT_SWEEP
    if 64 < n + m <= 128 and nodes in reg:
        return ob.PIVOT_SWEEP2
    return ob.PIVOT_CONDENSED if n + m >= 96 else ob.PIVOT_STATIC


def probe(ctx, name, wl, B, **kw):
    ss = pa.sqp_settings_default(); ss.max_iter = wl["max_iter"]; ss.line_search_max_iter = wl["ls_max_iter"]
    oss = ob.sqp_default_settings(); oss.max_iter = wl["max_iter"]; oss.line_search_max_iter = wl["ls_max_iter"]
    for k, v in kw.items():
        setattr(ss, k, v); setattr(oss, k, v)
    t0 = time.time()
    x, lam, info = ctx.sqp_solve_batch(wl["model"], wl["P"], wl["S"], wl["t0"], wl["tf"], B, wl["d"], wl["lbx"], wl["ubx"], sqp_settings=ss)
    t1 = time.time()
    dm = ob.ocp_dims(wl["model"], wl["P"], wl["S"])
    order = order_for(dm["n"], dm["m"], wl["P"] * wl["S"] + 1, kw)
    xo, lo, io = ob.sqp_solve_batch(wl["model"], wl["P"], wl["S"], wl["t0"], wl["tf"], B, wl["d"], wl["lbx"], wl["ubx"],
                                    sqp_settings=oss, pivot=order, threads=os.cpu_count())
    t2 = time.time()
    it = np.array([i.iter for i in io]); st = np.array([i.status for i in io]); qi = np.array([i.qp_solver_iter for i in io])
    same = (info["iter"] == it) & (info["status"] == st) & (info["qp_solver_iter"] == qi)
    f = lambda a, key: float(np.abs(info[key] - np.array([getattr(i, key) for i in io])).max())
    print(f"{name:34s} B={B:5d} identical-trajectory {same.mean():.4f}  x bit-equal {np.array_equal(x, xo)}  lam bit-equal {np.array_equal(lam, lo)}  "
          f"max|dx| {np.abs(x - xo).max():.2e} max|dlam| {np.abs(lam - lo).max():.2e} d(primal_norm) {f(0, 'primal_norm'):.1e} d(dual_norm) {f(0, 'dual_norm'):.1e} "
          f"d(viol) {f(0, 'max_violation'):.1e} d(cost) {f(0, 'cost'):.1e}  solved {np.mean(info['status'] == 0):.2f}  mean iter {info['iter'].mean():.2f}  gpu {t1 - t0:.2f}s cpu {t2 - t1:.2f}s",
          flush=True)


def main():
    big = "--big" in sys.argv
    ctx = pa.Context(0)
    if "--hunt" in sys.argv:   # localise the first SQP iteration at which a policy stops being bit-identical
        for mi in (1, 2, 3):
            for name, kw in (("block BFGS", dict(hessian_update=1)), ("ruiz", dict(preconditioner=1))):
                for P, S in ((6, 1), (5, 2)):
                    wl = workloads.robot_batch(64, P=P, S=S); wl["max_iter"] = mi
                    probe(ctx, f"{name} P{P}S{S} max_iter={mi}", wl, 64, **kw)
        ctx.close()
        return
    probe(ctx, "A robot P6S1", workloads.robot_batch(4096 if big else 512), 4096 if big else 512)
    probe(ctx, "A block BFGS", workloads.robot_batch(512), 512, hessian_update=1)
    probe(ctx, "D robot perturbed d", workloads.robot_batch(1024, perturb_d=True, first=5000), 1024)
    probe(ctx, "robot P4S1 (5 nodes, reg)", workloads.robot_batch(256, P=4, S=1), 256)
    probe(ctx, "A' robot P5S2 (88 rows)", workloads.robot_batch(256, P=5, S=2), 256)
    probe(ctx, "A' robot P5S3 (128 rows)", workloads.robot_batch(128, P=5, S=3), 128)
    probe(ctx, "A ruiz", workloads.robot_batch(256), 256, preconditioner=1)
    probe(ctx, "A admm", workloads.robot_batch(128), 128, qp_solver=1)
    probe(ctx, "A filter ls", workloads.robot_batch(128), 128, line_search=1)
    probe(ctx, "B cstr P5S2 (110 rows)", workloads.cstr_batch(1024 if big else 256), 1024 if big else 256)
    probe(ctx, "C kite stand-in (464 rows)", workloads.kite_standin_batch(16), 16)
    ctx.close()


if __name__ == "__main__":
    main()
