"""Developer tool (CPU only): the cross-order table of DESIGN.md §5 — every BASELINE configuration solved by the oracle twice, in the order of the
kernel that serves it (with the IEEE-only sin / cos / exp the kernels share) and as the reference computes (Eigen-style pivoted LDL^T, glibc) — and the
differences between the two runs: trajectories (SQP iterations, status, total ADMM iterations), reported KKT quantities, and the solutions in absolute
terms and scaled per variable by its box / steady-state magnitude. `python oracle/cross_order.py [A D B C R] [--full]` prints one JSON object.
The same function (cross_order_stats) is what tests/test_oracle_pins.py, tests/test_gpu_parity.py and bench.py use to compare a solution set with the
reference-order run, so the numbers in the bench line, the tests and the table come from one piece of code."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))   # repo root (this file lives in oracle/)
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


from polympc_amd.parity_stats import cross_order_stats, qp_level_stats, variable_scales  # noqa: E402,F401


def config_workload(cfg, B=None, full=False):
    from polympc_amd import workloads
    if cfg == "A":
        return workloads.robot_batch(B or 4096), "PIVOT_SWEEP"
    if cfg == "D":
        return workloads.robot_batch(B or 8192, perturb_d=True, first=5000), "PIVOT_SWEEP"
    if cfg == "B":
        return workloads.cstr_batch(B or (16384 if full else 2048)), "PIVOT_SWEEP2"
    if cfg == "C":
        return workloads.kite_standin_batch(B or (1024 if full else 64)), "PIVOT_CONDENSED"
    if cfg == "R":
        return workloads.robot_batch(B or 2048, P=5, S=3), None   # order decided by the route table (see kernel_order_R)
    raise ValueError(cfg)


def oracle_run(ob, wl, B, pivot, glibc, threads, hessian_update=0):
    oss = ob.sqp_default_settings(); oss.max_iter = wl["max_iter"]; oss.line_search_max_iter = wl["ls_max_iter"]; oss.hessian_update = hessian_update
    prev = ob.set_libm(glibc)
    try:
        return ob.sqp_solve_batch(wl["model"], wl["P"], wl["S"], wl["t0"], wl["tf"], B, wl["d"][:B], wl["lbx"][:B], wl["ubx"][:B],
                                  sqp_settings=oss, pivot=pivot, threads=threads)
    finally:
        ob.set_libm(prev)


def traced_qp_stream(ob, cfg, min_qps, B=None, hessian_update=0):
    """The QPs the reference-order SQP (Eigen-style pivoted LDL^T, glibc) emits for the first instances of configuration `cfg` — true collocation
    structure and conditioning, SURVEY 8d's "QP-only microbenchmark" — stacked until at least `min_qps` of them: dict of H [Q, n n], h, A [Q, m n],
    Alb, Aub, xlb, xub (column-major matrices, as the C ABI takes them)."""
    wl, _ = config_workload(cfg, B=B or 4096)
    oss = ob.sqp_default_settings(); oss.max_iter = wl["max_iter"]; oss.line_search_max_iter = wl["ls_max_iter"]; oss.hessian_update = hessian_update
    keys = ("H", "h", "A", "al", "au", "lx", "ux")
    parts = {k: [] for k in keys}
    got = 0
    prev = ob.set_libm(True)
    try:
        for b in range(wl["lbx"].shape[0]):
            t = ob.sqp_trace_qps(wl["model"], wl["P"], wl["S"], wl["t0"], wl["tf"], wl["d"][b:b + 1], wl["lbx"][b:b + 1], wl["ubx"][b:b + 1],
                                 sqp_settings=oss, pivot=ob.PIVOT_EIGEN, max_qps=wl["max_iter"])
            for k in keys:
                parts[k].append(t[k])
            got += t["H"].shape[0]
            if got >= min_qps:
                break
    finally:
        ob.set_libm(prev)
    q = {k: np.concatenate(v) for k, v in parts.items()}
    dm = ob.ocp_dims(wl["model"], wl["P"], wl["S"])
    return dict(H=q["H"], h=q["h"], A=q["A"], Alb=q["al"], Aub=q["au"], xlb=q["lx"], xub=q["ux"], n=wl["n"], m=wl["m"], instances=b + 1,
                structure=(dm["nx"], dm["nu"], dm["nn"], wl["P"]))


def reference_qp_solve(ob, q, threads=1):
    """Every QP of the stream solved once as the reference's boxADMM computes (PIVOT_EIGEN; the SQP constructor's QP settings, sqp_base.hpp:83-90)."""
    return ob.qp_solve_batch(q["H"], q["h"], q["A"], q["Alb"], q["Aub"], q["xlb"], q["xub"], settings=ob.sqp_qp_default_settings(),
                             pivot=ob.PIVOT_EIGEN, threads=threads)


def kernel_order(ob, cfg, wl):
    """The restatement order of the kernel that serves the configuration's size (the dispatch rule of pmpc_launch.hpp)."""
    rows = wl["n"] + wl["m"]
    if rows <= 64:
        return ob.PIVOT_SWEEP
    if rows <= ob.SWEEP2_MAX_ROWS:
        if wl["n"] <= 112 and wl["m"] <= 64:
            return ob.PIVOT_CONDSWEEP   # condensed register kernel (pmpc_qp_cond.hpp) since round 4
        return ob.PIVOT_SWEEP2
    return ob.PIVOT_CONDENSED   # the large-instance kernel inside the fused SQP kernel: condensed form since round 3


def main():
    from oracle import binding as ob
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    full = "--full" in sys.argv
    threads = len(os.sched_getaffinity(0))
    out = {}
    for cfg in (args or ["A", "D", "B", "C", "R"]):
        wl, _ = config_workload(cfg, full=full)
        B = wl["lbx"].shape[0]
        t0 = time.perf_counter()
        xk, lk, ik = oracle_run(ob, wl, B, kernel_order(ob, cfg, wl), False, threads)
        xr, lr, ir = oracle_run(ob, wl, B, ob.PIVOT_EIGEN, True, threads)
        rec = cross_order_stats(cfg, wl, xk, lk, ik, xr, lr, ir)
        rec["kernel_order"] = int(kernel_order(ob, cfg, wl)); rec["seconds"] = time.perf_counter() - t0
        rec["qp_solves"] = int(sum(i.iter for i in ir))
        out[cfg] = rec
        print(cfg, json.dumps(rec), flush=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
