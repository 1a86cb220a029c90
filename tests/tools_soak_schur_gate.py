"""Developer tool (GPU): the block-structured route — conditioning gate, redo launch, filter line search, the bordered NP = 1 form — against its CPU restatement
over start penalties, QP settings that move boxADMM's control flow, Hessian policies and line searches. Prints one line per combination (with the number of
instances the gate sent to the redo launch); exits non-zero when anything is not bit-identical or a flag differs.

    python tests/tools_soak_schur_gate.py [B]"""
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
from test_oracle_pins import _parking_batch    # noqa: E402

QP_VARIANTS = [dict(), dict(adaptive_rho=0), dict(check_termination=1), dict(adaptive_rho_interval=7), dict(max_iter=7), dict(alpha=1.6), dict(sigma=1e-3),
               dict(adaptive_rho_tolerance=1.5, adaptive_rho_interval=10)]
SQP_VARIANTS = [dict(hessian_update=1), dict(hessian_update=1, regularisation=2), dict(exact_hessian_every_iter=1, regularisation=2), dict(hessian_update=1, line_search=1)]


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    ctx = pa.Context(0)
    bad = 0
    lbx, ubx, xg, d = _parking_batch(B)
    cases = [("cstr 11", workloads.cstr_batch(B), None), ("robot 16", workloads.robot_batch(B, P=5, S=3), None), ("robot 11", workloads.robot_batch(B, P=5, S=2), None),
             ("parking 11", dict(model=pa.MODEL_PARKING, P=5, S=2, t0=0.0, tf=1.0, d=d, lbx=lbx, ubx=ubx, ls_max_iter=10), xg)]
    for name, wl, guess in cases:
        for skw in SQP_VARIANTS:
            if name.startswith("parking") and skw.get("line_search"): continue   # (no hook build of the bordered form)
            for rho0 in (0.1, 3.0, 30.0, 300.0):
                for qkw in (QP_VARIANTS if rho0 == 0.1 else QP_VARIANTS[:2]):
                    ss = pa.sqp_settings_default(); oss = ob.sqp_default_settings()
                    for st in (ss, oss):
                        st.max_iter = 5; st.line_search_max_iter = wl["ls_max_iter"]
                        if name.startswith("parking") and skw.get("hessian_update"): st.max_iter = 3   # (quasi-Newton updates diverge on the minimal-time problem from the fourth iteration on — NaN on both sides; the reference solves it with exact Hessians)
                        for k, v in skw.items(): setattr(st, k, v)
                    if name.startswith("parking"): ss.kkt_form = 2   # the bordered NP = 1 form is served on request (pmpc_sqp_settings::kkt_form = 2)
                    qs = pa.qp_settings_sqp_default(); oqs = ob.sqp_qp_default_settings()
                    for k, v in dict(qkw, rho=rho0).items():
                        setattr(qs, k, v); setattr(oqs, k, v)
                    x, lam, info = ctx.sqp_solve_batch(wl["model"], wl["P"], wl["S"], wl["t0"], wl["tf"], B, wl["d"], wl["lbx"], wl["ubx"], x_guess=guess, sqp_settings=ss, qp_settings=qs)
                    route = pa.capi.ROUTE_NAMES[ctx.last_route()]
                    xo, lo, io = ob.sqp_solve_batch(wl["model"], wl["P"], wl["S"], wl["t0"], wl["tf"], B, wl["d"], wl["lbx"], wl["ubx"], x_guess=guess, sqp_settings=oss, qp_settings=oqs,
                                                    pivot=ob.PIVOT_SCHUR, threads=8)
                    fo = np.array([i.flags for i in io])
                    try:
                        assert route == "schur", route
                        assert np.array_equal(info["flags"] & pa.capi.FLAG_ILLCOND, fo & ob.FLAG_ILLCOND), "conditioning flags differ"
                        nonfinite = ~(np.isfinite(x).all(axis=1) & np.isfinite(lam).all(axis=1))
                        assert np.array_equal((info["flags"] & pa.capi.FLAG_NONFINITE) != 0, nonfinite) or np.all(((info["flags"] & pa.capi.FLAG_NONFINITE) != 0) >= nonfinite), "non-finite flag"
                        if np.isfinite(xo).all() and np.isfinite(lo).all():
                            T._assert_same_solve(info, io, x, xo, lam, lo)
                            res = "ok"
                        else:   # (a divergent setup — quasi-Newton updates on the unregularised minimal-time problem: the same NaNs in the same places, the same counts)
                            assert np.array_equal(x, xo, equal_nan=True) and np.array_equal(lam, lo, equal_nan=True), "non-finite results differ"
                            assert info["iter"].tolist() == [i.iter for i in io] and info["qp_solver_iter"].tolist() == [i.qp_solver_iter for i in io], "counts differ"
                            res = "ok (non-finite on both sides, identical)"
                    except AssertionError as e:
                        res = "MISMATCH " + str(e).split("\n")[0][:90]; bad += 1
                    print(f"{name:10s} {str(skw):62s} rho0={rho0:<6g} {str(qkw):58s} redone {int(np.count_nonzero(info['flags'] & pa.capi.FLAG_ILLCOND)):3d}/{B}  {res}", flush=True)
    ctx.close()
    print("mismatches:", bad)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
