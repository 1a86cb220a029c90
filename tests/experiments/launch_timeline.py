"""Developer probe (GPU; PMPC_LIB = a library whose robot translation unit was built with -DPMPC_EXPERIMENT_WG_STAMPS): the timeline of ONE launch of the headline
kernel — per 100 us of the launch: resident wavefronts, wavefronts that start / end, SQP iterations completed (the work the chip gets done) — for the BASELINE batch
(4096: two rounds of 2048 resident wavefronts) and for config D's 8192. VERDICT r5 item 1d: where the 21 % between 0.278 us (A) and 0.230 us (D) per instance go."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import polympc_amd as pa
from polympc_amd import workloads
cap = 10
ctx = pa.Context(0)
for B, perturb in ((4096, False), (8192, True)):
    wl = workloads.robot_batch(B, perturb_d=perturb, first=5000 if perturb else 0)
    ss = pa.sqp_settings_default(); ss.max_iter = 10; ss.line_search_max_iter = 10
    args = (wl["model"], wl["P"], wl["S"], wl["t0"], wl["tf"], B, wl["d"], wl["lbx"], wl["ubx"])
    ctx.sqp_solve_batch(*args, sqp_settings=ss)
    h = ctx.iteration_trace_create(B, cap)
    ss.iteration_trace = h; ss.iteration_trace_capacity = cap
    x, lam, info = ctx.sqp_solve_batch(*args, sqp_settings=ss)
    tr = np.asarray(ctx.iteration_trace_download(B, cap, h))
    it = info["iter"]
    start = tr[:, 0, 2]; t0 = start.min()
    ends = np.full((B, cap), np.nan)
    for p in range(cap):
        a = it > p
        ends[a, p] = (tr[a, p, 1] - t0) / 100.0       # us
    start_us = (start - t0) / 100.0
    fin = np.array([ends[b, it[b] - 1] for b in range(B)])
    dur = fin - start_us
    print(f"== batch {B}: makespan {fin.max():.0f} us = {fin.max() / B:.3f} us per instance; SQP iterations {int(it.sum())}; instance life: mean {dur.mean():.0f} us, p50 {np.percentile(dur, 50):.0f}, p99 {np.percentile(dur, 99):.0f}, max {dur.max():.0f}; "
          f"iterations hist {np.bincount(it)[1:].tolist()}")
    first_round = start_us < 20.0
    print(f"   wavefronts that started in the first 20 us: {int(first_round.sum())} (the resident set); the last start at {start_us.max():.0f} us; the instance that ends last started at {start_us[np.argmax(fin)]:.0f} us and ran {it[np.argmax(fin)]} iterations")
    edges = np.arange(0.0, fin.max() + 100.0, 100.0)
    print("   t [us]   resident(mid-bin)  started  ended  iterations completed  mean us per iteration of those")
    itdur = np.diff(np.concatenate([start_us[:, None], ends], axis=1), axis=1)    # duration of every iteration
    for lo, hi in zip(edges[:-1], edges[1:]):
        mid = 0.5 * (lo + hi)
        resident = int(((start_us <= mid) & (fin > mid)).sum())
        sel = (ends >= lo) & (ends < hi)
        print(f"   {lo:6.0f}   {resident:8d}          {int(((start_us >= lo) & (start_us < hi)).sum()):6d} {int(((fin >= lo) & (fin < hi)).sum()):6d}  {int(sel.sum()):8d}              {np.nanmean(itdur[sel]) if sel.any() else float('nan'):8.1f}")
    ctx.iteration_trace_destroy(h) if hasattr(ctx, "iteration_trace_destroy") else None
ctx.close()
