"""Developer probe (library built with PMPC_EXTRA_HIPCC_FLAGS=-DPMPC_RR_PROFILE; sets PMPC_SQP_RR=1, PMPC_PHASE_PROFILE=1): where the wavefronts of sqp_kernel_rr spend their cycles."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["PMPC_PHASE_PROFILE"] = "1"
os.environ["PMPC_SQP_RR"] = "1"
import polympc_amd as pa
from polympc_amd import workloads
B = int(os.environ.get("B", "4096"))
wl = workloads.robot_batch(B)
ctx = pa.Context(0)
ss = pa.sqp_settings_default(); ss.max_iter = 10; ss.line_search_max_iter = 10
args = (wl["model"], wl["P"], wl["S"], wl["t0"], wl["tf"], B, wl["d"], wl["lbx"], wl["ubx"])
for rep in range(3):
    ctx.phase_cycles(reset=True)
    t = time.perf_counter(); x, lam, info = ctx.sqp_solve_batch(*args, sqp_settings=ss); dt = time.perf_counter() - t
    c = np.array(ctx.phase_cycles(reset=True), dtype=np.float64)
    waves, items = c[6], c[5]
    print(f"B={B} waves={waves:.0f} items={items:.0f} (sum iters {info['iter'].sum()}) life/wave={c[0]/max(waves,1):.0f} cyc; per item: solve={c[1]/max(items,1):.0f} wait={c[2]/max(items,1):.0f} load={c[3]/max(items,1):.0f} store+push={c[4]/max(items,1):.0f}; "
          f"sum(solve+wait+load+store)/life={(c[1]+c[2]+c[3]+c[4])/max(c[0],1):.3f} hist={np.bincount(info['iter'])}")
