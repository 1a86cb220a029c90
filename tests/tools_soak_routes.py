"""Developer tool (GPU): GPU vs CPU restatement over the routing table of the fused SQP kernel — every grid of 3..16 nodes of the robot, CSTR and (NP = 1) parking models
(register paths, LDS-resident kernel, HBM-factor kernel) under the default policy and the policies that change the route (block BFGS, Ruiz, filter line
search, OSQP-form ADMM). Prints one line per combination; exits non-zero when anything is not bit-identical.

    python tests/tools_soak_routes.py [B]"""
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


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    ctx = pa.Context(0)
    bad = 0
    grids = [(P, S) for P in range(2, 8) for S in range(1, 4) if 3 <= P * S + 1 <= 16]
    policies = [dict(), dict(hessian_update=1), dict(preconditioner=1), dict(line_search=1), dict(qp_solver=1), dict(preconditioner=1, line_search=1, hessian_update=1), dict(kkt_form=1),
                dict(regularisation=1, exact_hessian_every_iter=1)]   # (round 6: eigenvalue mirroring — the hook builds of the register kernels on 7 / 11 / 16 nodes, the LDS / HBM-resident kernels elsewhere)
    from test_oracle_pins import _parking_batch
    for model in (0, 1, pa.MODEL_PARKING, pa.MODEL_PARKING_NG, pa.MODEL_ROBOT_NG):
        for P, S in grids:
            dm = ob.ocp_dims(model, P, S)
            if model == 0:
                wl = workloads.robot_batch(B, P=P, S=S); wl["max_iter"] = 6
            elif model in (pa.MODEL_PARKING, pa.MODEL_PARKING_NG):   # (round 6) NP = 1 (and NG = 1: the path constraint of nonlinear_constraints_test.cpp, binding): the reference's minimal-time problem, perturbed per instance
                lbx, ubx, xg, d = _parking_batch(B, P * S + 1)
                wl = dict(model=model, P=P, S=S, t0=0.0, tf=1.0, d=d, lbx=lbx, ubx=ubx, x_guess=xg, max_iter=6, ls_max_iter=10, settings=dict(regularisation=2, exact_hessian_every_iter=1))
                if model == pa.MODEL_PARKING_NG: wl["lbg"] = np.full((B, P * S + 1), -10.0); wl["ubg"] = np.full((B, P * S + 1), 1.2)
            elif model == pa.MODEL_ROBOT_NG:
                wl = workloads.robot_batch(B, P=P, S=S); wl["max_iter"] = 6; wl["model"] = model
                wl["lbg"] = np.full((B, P * S + 1), -1e3); wl["ubg"] = np.full((B, P * S + 1), 3.0)
            else:
                lbx, ubx = T._cstr_grid(B, P, S)
                wl = dict(model=1, P=P, S=S, t0=0.0, tf=100.0, d=np.zeros((B, 1)), lbx=lbx, ubx=ubx, max_iter=6, ls_max_iter=20)
            for kw in policies:
                if kw.get("qp_solver") and 2 * dm["n"] + dm["m"] > 180: continue    # the stacked system lives in LDS only
                if kw.get("regularisation") == 1 and dm["n"] > 80: continue          # the Jacobi workspace is 16 n^2 bytes of LDS
                try:
                    (x, lam, info), (xo, lo, io) = T._sqp_both(ctx, ob, wl, B, **kw)
                    T._assert_same_solve(info, io, x, xo, lam, lo)
                    res = "ok"
                except AssertionError as e:
                    res = "MISMATCH " + str(e).split("\n")[0][:90]; bad += 1
                except RuntimeError as e:
                    res = "error " + str(e)[:60]
                print(f"model {model} P={P} S={S} nodes={P * S + 1:2d} rows={dm['n'] + dm['m']:3d} {str(kw):28s} {res}", flush=True)
    ctx.close()
    print("mismatches:", bad)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
