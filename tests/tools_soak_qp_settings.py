"""Developer tool (GPU): the condensed register route (and, with kkt_form = 1, the full inverse on the same grids) against its CPU restatement under QP
settings that move the control flow of boxADMM — no adaptive rho, a residual check every iteration / every 25, a co-prime adaptation interval, a tiny
iteration cap, other rho / alpha / sigma / tolerances. Prints one line per combination; exits non-zero when anything is not bit-identical.

    python tests/tools_soak_qp_settings.py [B]"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
sys.path.insert(0, HERE)
import polympc_amd as pa                       # noqa: E402
from polympc_amd import workloads              # noqa: E402
from oracle import binding as ob               # noqa: E402
import test_gpu_parity as T                    # noqa: E402

VARIANTS = [dict(), dict(adaptive_rho=0), dict(check_termination=1), dict(check_termination=25, adaptive_rho_interval=7), dict(max_iter=7),
            dict(rho=1.0), dict(alpha=1.6), dict(sigma=1e-3), dict(eps_abs=1e-6, eps_rel=1e-6), dict(adaptive_rho_tolerance=1.5, adaptive_rho_interval=10),
            dict(check_termination=0, max_iter=40)]


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    ctx = pa.Context(0)
    bad = 0
    cases = [("cstr 11", workloads.cstr_batch(B)), ("robot 16", workloads.robot_batch(B, P=5, S=3)), ("robot 11", workloads.robot_batch(B, P=5, S=2)),
             ("robot 13", workloads.robot_batch(B, P=6, S=2))]
    for name, wl in cases:
        dm = ob.ocp_dims(wl["model"], wl["P"], wl["S"])
        for kkt_form in (0, 1):
            for kw in VARIANTS:
                ss = pa.sqp_settings_default(); oss = ob.sqp_default_settings()
                for st in (ss, oss):
                    st.max_iter = 5; st.line_search_max_iter = wl["ls_max_iter"]; st.kkt_form = kkt_form
                qs = pa.qp_settings_sqp_default(); oqs = ob.sqp_qp_default_settings()
                for k, v in kw.items():
                    setattr(qs, k, v); setattr(oqs, k, v)
                x, lam, info = ctx.sqp_solve_batch(wl["model"], wl["P"], wl["S"], wl["t0"], wl["tf"], B, wl["d"], wl["lbx"], wl["ubx"], sqp_settings=ss, qp_settings=qs)
                order = T._gpu_order(ob, dm["n"], dm["m"], wl["P"] * wl["S"] + 1, kkt_form=kkt_form, ng=dm["ng"])
                xo, lo, io = ob.sqp_solve_batch(wl["model"], wl["P"], wl["S"], wl["t0"], wl["tf"], B, wl["d"], wl["lbx"], wl["ubx"], sqp_settings=oss, qp_settings=oqs,
                                                pivot=order, threads=8)
                try:
                    T._assert_same_solve(info, io, x, xo, lam, lo)
                    res = "ok"
                except AssertionError as e:
                    res = "MISMATCH " + str(e).split("\n")[0][:90]; bad += 1
                print(f"{name:9s} kkt_form={kkt_form} route={pa.capi.ROUTE_NAMES[ctx.last_route()]:8s} {str(kw):62s} {res}", flush=True)
    ctx.close()
    print("mismatches:", bad)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
