"""Developer probe (CPU only): the block-structured order (PIVOT_SCHUR) with its refinement step only above a conditioning estimate — how many solves keep the step, what happens to the
trajectories (iteration counts against the shipped order and the reference order, distance to exact arithmetic) on the block-BFGS streams of configs B / R / A."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import binding as ob
from polympc_amd import workloads
import test_gpu_parity as T
L = ob.lib(); L.orc_set_schur_refine_gate.restype = C.c_double; L.orc_set_schur_refine_gate.argtypes = [C.c_double]
def counts(reset=True):
    a = (C.c_longlong * 2)(); L.orc_schur_refine_counts(a, int(reset)); return a[0], a[1]
which = sys.argv[1] if len(sys.argv) > 1 else "B"
Bn = int(sys.argv[2]) if len(sys.argv) > 2 else 256
if which == "B": wl = workloads.cstr_batch(Bn)
elif which == "R": wl = workloads.robot_batch(Bn, P=5, S=3)
else: wl = workloads.robot_batch(Bn)
ss = ob.sqp_default_settings(); ss.max_iter = wl["max_iter"]; ss.line_search_max_iter = wl["ls_max_iter"]; ss.hessian_update = 1
run = lambda piv: ob.sqp_solve_batch(wl["model"], wl["P"], wl["S"], wl["t0"], wl["tf"], Bn, wl["d"], wl["lbx"], wl["ubx"], sqp_settings=ss, pivot=piv, threads=8)
key = lambda info: [(i.iter, i.status, i.qp_solver_iter) for i in info]
xe, le, ie = run(ob.PIVOT_EIGEN); xx, lx, ix = run(ob.PIVOT_EXACT)
L.orc_set_schur_refine_gate(0.0); counts()
xs, ls, is_ = run(ob.PIVOT_SCHUR); c0 = counts()
print(f"{which} {Bn} instances, block BFGS. shipped order: solves with / without the step {c0}; same trajectory as eigen {sum(a == b for a, b in zip(key(is_), key(ie)))}, as exact {sum(a == b for a, b in zip(key(is_), key(ix)))}; |x - exact| schur {np.abs(xs - xx).max():.1e} eigen {np.abs(xe - xx).max():.1e}; flags {sum(i.flags != 0 for i in is_)}")
for g in [float(v) for v in os.environ.get("GATES", "1e2,1e3,1e4,1e5,1e6,1e30").split(",")]:
    L.orc_set_schur_refine_gate(g); counts()
    x, l, inf = run(ob.PIVOT_SCHUR); c = counts()
    same_s = sum(a == b for a, b in zip(key(inf), key(is_))); same_e = sum(a == b for a, b in zip(key(inf), key(ie))); same_x = sum(a == b for a, b in zip(key(inf), key(ix)))
    sel = np.array([a == b for a, b in zip(key(inf), key(ix))])
    print(f"  gate {g:7.0e}: with / without the step {c} ({c[1] / max(1, sum(c)):.2f} skipped); same trajectory as shipped {same_s}, eigen {same_e}, exact {same_x}; |x - exact| on those {np.abs(x - xx)[sel].max():.1e}; all {np.abs(x - xx).max():.1e}")
L.orc_set_schur_refine_gate(0.0)
