"""Developer probe (library built with PMPC_EXTRA_HIPCC_FLAGS=-DPMPC_RR_PROFILE; sets PMPC_SQP_RR=1): per-item completion stamps of sqp_kernel_rr."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["PMPC_SQP_RR"] = "1"
import polympc_amd as pa
from polympc_amd import workloads
B = int(os.environ.get("B", "4096")); cap = 10
wl = workloads.robot_batch(B)
ctx = pa.Context(0)
ss = pa.sqp_settings_default(); ss.max_iter = 10; ss.line_search_max_iter = 10
args = (wl["model"], wl["P"], wl["S"], wl["t0"], wl["tf"], B, wl["d"], wl["lbx"], wl["ubx"])
ctx.sqp_solve_batch(*args, sqp_settings=ss)
h = ctx.iteration_trace_create(B, cap)
ss.iteration_trace = h; ss.iteration_trace_capacity = cap
x, lam, info = ctx.sqp_solve_batch(*args, sqp_settings=ss)
tr = ctx.iteration_trace_download(B, cap, h)
done = tr[:, :, 1].copy(); it = info["iter"]
stamped = done > 1e6   # an item that ran several iterations (an instance that kept its wavefront) stamps its LAST record only; the others hold alpha
done[~stamped] = 0.0
t0 = done[done > 0].min()
us = np.where(stamped, (done - t0) / 100.0, np.nan)   # 100 MHz
fin = np.array([us[b, it[b] - 1] for b in range(B)])
print("makespan us", fin.max(), "finish percentiles 50/90/99/100:", np.percentile(fin, [50, 90, 99, 100]))
for p in range(cap):
    a = (it > p) & stamped[:, p]
    if not a.any():
        continue
    d = us[a, p]
    solve = tr[a, p, 3]; pre = tr[a, p, 2]
    print(f"pass {p}: items ending here {a.sum():5d} (active {(it > p).sum():5d}) done at us p1 {np.percentile(d,1):7.1f} p50 {np.percentile(d,50):7.1f} p99 {np.percentile(d,99):7.1f} max {d.max():7.1f} | solve cyc p50 {np.percentile(solve,50):8.0f} p99 {np.percentile(solve,99):8.0f} max {solve.max():8.0f} | pop+load cyc p50 {np.percentile(pre,50):6.0f} p99 {np.percentile(pre,99):7.0f}")
late = np.argsort(-fin)[:8]
for b in late:
    print("late instance", b, "iters", it[b], "xcc", int(tr[b, 0, 4]), "done us", np.round(us[b, :it[b]], 0), "item kcyc (an item = one or, for an instance that kept its wavefront, several iterations)", np.round(np.where(stamped[b, :it[b]], tr[b, :it[b], 3], 0) / 1e3, 0))
xc = tr[:, 0, 4].astype(int)
for q in range(8):
    m = xc == q
    print("xcc", q, "instances", m.sum(), "b%8 values", np.unique(np.arange(B)[m] % 8), "last finish", fin[m].max() if m.any() else None)
