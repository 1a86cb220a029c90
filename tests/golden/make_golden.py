"""Generate golden vectors from the REFERENCE's own compiled artefacts (oracle/_ref/libref_casadi_robot.so, built by
oracle/Makefile from /root/reference/tests/solvers/sqp/casadi_codegen/*.cpp — the reference's CasADi-generated C for
the 11-node (P=5, S=2) mobile-robot NLP: 55 variables, 33 equality constraints, t in [0,1], wheel base 1, Q=R=I,
Mayer x'x).  Run in the build container (needs /root/reference):  python tests/golden/make_golden.py
Output: tests/golden/casadi_robot_P5S2.npz  (inputs + expected outputs only; no reference source)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import binding as ob  # noqa: E402

ob.build(force=True)
r = ob.RefCasadiRobot()
rng = np.random.default_rng(20260929)
pts = [np.array([0.1 * (i % 7) + 0.05 * i / 55 for i in range(55)])]  # SURVEY.md Appendix A test point
pts += [rng.uniform(-1.0, 1.0, 55) for _ in range(5)]
pts += [np.zeros(55)]
lams = [rng.uniform(-1.0, 1.0, 33) for _ in pts]
lams[-1] = np.zeros(33)
out = dict(x=np.array(pts), lam=np.array(lams), cost=[], c=[], jac=[], cost_grad=[], cost_hess=[], lag=[], lag_grad=[],
           lag_hess=[])
for x, lam in zip(pts, lams):
    out["cost"].append(r.cost(x))
    out["c"].append(r.constraint(x))
    out["jac"].append(r._dense("fconstraints_jacobian", r._call("fconstraints_jacobian", [x], [r.nnz("fconstraints_jacobian")])[0]))
    out["cost_grad"].append(r._call("fcost_gradient", [x], [55])[0])
    out["cost_hess"].append(r._dense("fcost_hessian", r._call("fcost_hessian", [x], [r.nnz("fcost_hessian")])[0]))
    out["lag"].append(r._call("flagrangian", [x, lam], [1])[0][0])
    out["lag_grad"].append(r._call("flagrangian_gradient", [x, lam], [55])[0])
    out["lag_hess"].append(r._dense("flagrangian_hessian", r._call("flagrangian_hessian", [x, lam], [r.nnz("flagrangian_hessian")])[0]))
np.savez_compressed(os.path.join(os.path.dirname(__file__), "casadi_robot_P5S2.npz"), **{k: np.array(v) for k, v in out.items()})
print("wrote casadi_robot_P5S2.npz with", len(pts), "points")
