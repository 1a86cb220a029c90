#!/usr/bin/env python
"""bench.py — box-ADMM QP subproblem solves/s of the fused SQP hot path on MI355X.

One "step" = one pass of the hot path over one batch: pmpc_sqp_solve_batch_dev on B = 4096 mobile-robot OCPs per GPU
(BASELINE.json configs[1]: nx=3 nu=2, Chebyshev N=6 -> 7 nodes, n=35, m=21, KKT 56x56, fp64, randomised x0, zero
guesses, SQP max_iter=10 / line search 10, QP settings = SQPBase constructor defaults). Every SQP iteration solves
one box-ADMM QP subproblem (plus its linearisation, Hessian update and line search, all inside the same kernel), so
value = (sum over instances of SQP iterations) * steps / wall time. Inputs are resident in HBM before the timed region.
Multi-GPU: one process per GPU, the batch shards embarrassingly (each rank solves its own 4096 instances — weak
scaling, no data-path collective); torch.distributed (RCCL) is used only for the barrier and the max-over-ranks time.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_HBM_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)
PEAK_FP64_TFLOPS = 78.6      # fp64 vector peak, 256 CU x 4 SIMD x 16 lanes x 2 flop x 2.4 GHz


def qp_algorithmic_bytes(n, m):
    """read the QP data once, write the solution once."""
    return 8 * (n * n + m * n + 3 * n + 2 * m) + 8 * (2 * n + m) + 32


def qp_algorithmic_flops(n, m, it, f):
    N = n + m
    chk = it // 10
    return f * N ** 3 / 3.0 + it * (2.0 * N * N + 12.0 * N) + chk * 2.0 * (n * n + 2 * m * n)


def measured_traffic():
    """HBM bytes per launch of the bench kernel from the committed PMC summary of this round (rocprofv3 --pmc FETCH_SIZE /
    WRITE_SIZE in separate passes, corrected by the calibration kernel; tests/tools_pmc.sh + tests/tools_pmc_summary.py).
    bench.py cannot run rocprofv3 around itself, so it reports the number measured on the same command line."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary.json"))):
        try:
            t = json.load(open(f)).get("traffic")
        except Exception:
            t = None
        if t:
            best = (t["bytes_per_launch"], os.path.basename(f))
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=4096, help="OCP instances per GPU")
    ap.add_argument("--cpu-sample", type=int, default=4096, help="instances per pass of the CPU-baseline sample (0 = skip)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="minimum wall time of the CPU-baseline sample")
    ap.add_argument("--streams", type=int, default=1, help="1 (default, the contract's configuration): every step is one launch on one "
                    "stream. S > 1: consecutive steps alternate over S contexts (own stream, workspace and output buffers), so that a "
                    "step's tail overlaps the next step's start — a pipelined-server figure, reported in DESIGN.md, not the bench line")
    args = ap.parse_args()

    import numpy as np
    import torch
    import polympc_amd as pa
    from polympc_amd import workloads, sharding

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: polympc_amd has no CPU fallback")
    # developer switches for exercising the N > 1 code path on a box with ONE GPU (every rank on device 0, gloo for the barrier and
    # the two reductions); the driver's multi-GPU runs use neither: one rank per GPU, nccl (= RCCL)
    backend = os.environ.get("PMPC_BENCH_BACKEND", "nccl")
    if os.environ.get("PMPC_BENCH_SINGLE_DEVICE") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    dev = torch.device("cuda", local_rank)
    red_dev = dev if backend == "nccl" else torch.device("cpu")   # where the statistics tensors of the reductions live

    B = args.batch
    wl = workloads.robot_batch(B, first=sharding.shard_first_instance(rank, B))   # each rank owns a contiguous shard of the instance stream
    n, m = wl["n"], wl["m"]
    S_ = max(1, args.streams)
    streams = [torch.cuda.Stream(dev) for _ in range(S_)]   # real (non-null) HIP streams: the kernels and the timing events share them
    stream = streams[0]
    torch.cuda.set_stream(stream)
    ctxs = [pa.Context(local_rank, stream=st.cuda_stream) for st in streams]
    ctx = ctxs[0]
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    d_d, d_lbx, d_ubx = t(wl["d"]), t(wl["lbx"]), t(wl["ubx"])
    outs = [(torch.zeros(B, n, dtype=torch.float64, device=dev), torch.zeros(B, m + n, dtype=torch.float64, device=dev),
             torch.zeros(B, 48, dtype=torch.uint8, device=dev)) for _ in range(S_)]
    d_x, d_lam, d_info = outs[0]
    ss = pa.sqp_settings_default(); ss.max_iter = wl["max_iter"]; ss.line_search_max_iter = wl["ls_max_iter"]
    qs = pa.qp_settings_sqp_default()
    torch.cuda.synchronize(dev)
    counter = [0]

    def step():
        k = counter[0] % S_; counter[0] += 1
        ox, ol, oi = outs[k]
        ctxs[k].sqp_solve_batch_dev(wl["model"], wl["P"], wl["S"], wl["t0"], wl["tf"], B, d_d, d_lbx, d_ubx, ox, ol, oi, ss, qs)
        return streams[k]

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize(dev)
    if dist:
        dist.barrier()
    torch.cuda.synchronize(dev)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for e0, e1 in evs:
        st = streams[counter[0] % S_]
        e0.record(st)
        step()
        e1.record(st)
    torch.cuda.synchronize(dev)
    if dist:
        dist.barrier()
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    kernel_ms = float(np.mean([e0.elapsed_time(e1) for e0, e1 in evs]))

    info = np.frombuffer(d_info.cpu().numpy().tobytes(), dtype=pa.capi.SQP_INFO_DTYPE)
    qp_solves = int(info["iter"].sum())
    admm_iters = int(info["qp_solver_iter"].sum())
    solved = int((info["status"] == pa.SQP_SOLVED).sum())
    (qp_all, admm_all, solved_all), elapsed = sharding.combine_stats(dist, red_dev, [qp_solves, admm_iters, solved], elapsed)

    if rank == 0:
        value = qp_all * args.steps / elapsed
        it_per_qp = admm_iters / max(qp_solves, 1)
        # roofline of the dominant kernel (sqp_kernel<RobotOCP>), per launch on this rank
        bytes_launch = qp_algorithmic_bytes(n, m) * qp_solves
        flops_launch = qp_algorithmic_flops(n, m, it_per_qp, 1.0) * qp_solves
        ach_gbs = bytes_launch / (kernel_ms * 1e-3) / 1e9
        ach_tf = flops_launch / (kernel_ms * 1e-3) / 1e12
        tr = measured_traffic() if B == 4096 else None
        out = {
            "metric": "box-ADMM QP subproblem solves/s (fused SQP hot path)", "value": value, "unit": "QP solves/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "mobile_robot OCP nx=3 nu=2 N=6 (P=6,S=1; n=35 m=21 KKT 56), batch=%d per GPU, randomised x0, fp64, "
                                   "SQP max_iter=10 ls=10, QP = SQPBase defaults" % B,
                       "global_batch": B * world, "parallelism": "batch-shard x%d (no collectives)" % world,
                       **({"streams": S_, "note": "steps pipelined over %d streams: NOT the contract's configuration" % S_} if S_ > 1 else {})},
            "sqp_solves_per_s": B * world * args.steps / elapsed, "qp_solves_per_step": qp_all, "admm_iters_per_qp": admm_all / max(qp_all, 1),
            "sqp_solved_fraction": solved_all / (B * world),
            "roofline": {"bound": "hbm", "achieved": ach_gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": ach_gbs / PEAK_HBM_GBS,
                         "traffic": (tr[0] if tr else None), "traffic_source": (("profiles/" + tr[1]) if tr else None),
                         "algorithmic_bytes_per_launch": bytes_launch, "kernel": "sqp_kernel<RobotOCP,35,21>", "kernel_ms": kernel_ms,
                         "note": "algorithmic bytes = 17616 B per QP subproblem x QPs per launch (SURVEY 8d); the path is bound by fp64 issue / "
                                 "dependent-chain latency, not by HBM: see roofline_fp64 and DESIGN.md"},
            "roofline_fp64": {"bound": "fp64-valu", "achieved": ach_tf, "peak": PEAK_FP64_TFLOPS, "unit": "TFLOP/s", "frac": ach_tf / PEAK_FP64_TFLOPS},
        }
        if world == 1:
            # the same batch with the Hessian update the reference's own mobile-robot MPC test plugs in (mpc_wrapper_test.cpp:100-105:
            # ContinuousOCP's block BFGS). Reported beside the bench line, which stays on SQPBase's defaults (dense damped BFGS).
            ss2 = pa.sqp_settings_default(); ss2.max_iter = wl["max_iter"]; ss2.line_search_max_iter = wl["ls_max_iter"]; ss2.hessian_update = 1
            vx, vl, vi = torch.zeros_like(d_x), torch.zeros_like(d_lam), torch.zeros_like(d_info)
            vstep = lambda: ctx.sqp_solve_batch_dev(wl["model"], wl["P"], wl["S"], wl["t0"], wl["tf"], B, d_d, d_lbx, d_ubx, vx, vl, vi, ss2, qs)
            for _ in range(args.warmup):
                vstep()
            torch.cuda.synchronize(dev)
            tv = time.perf_counter()
            for _ in range(args.steps):
                vstep()
            torch.cuda.synchronize(dev)
            tv = (time.perf_counter() - tv) / args.steps
            vinfo = np.frombuffer(vi.cpu().numpy().tobytes(), dtype=pa.capi.SQP_INFO_DTYPE)
            out["variant_block_bfgs"] = {"settings": "hessian_update = 1 (continuous_ocp.hpp:2304-2431), everything else as the bench line",
                                         "ms_per_step": tv * 1e3, "qp_solves_per_s": int(vinfo["iter"].sum()) / tv,
                                         "sqp_solves_per_s": B / tv, "sqp_solved_fraction": float((vinfo["status"] == pa.SQP_SOLVED).mean())}
        if args.cpu_sample > 0 and world == 1:
            from oracle import binding as ob   # CPU restatement of the reference algorithm: baseline only, never the product path
            Bc = min(args.cpu_sample, B)
            cores = os.cpu_count() or 1
            oss = ob.sqp_default_settings(); oss.max_iter = wl["max_iter"]; oss.line_search_max_iter = wl["ls_max_iter"]
            tc, cpu_qps, passes = 0.0, 0, 0
            while tc < args.cpu_seconds and passes < 2000:   # bounded sample: repeated passes over the same instances
                t1 = time.perf_counter()
                xo, lo, io = ob.sqp_solve_batch(ob.MODEL_ROBOT, wl["P"], wl["S"], wl["t0"], wl["tf"], Bc, wl["d"][:Bc], wl["lbx"][:Bc], wl["ubx"][:Bc],
                                                sqp_settings=oss, pivot=ob.PIVOT_EIGEN, threads=cores)
                tc += time.perf_counter() - t1
                cpu_qps += sum(i.iter for i in io)
                passes += 1
            out["cpu_baseline"] = {"value": cpu_qps / tc, "unit": "QP solves/s", "cores": cores, "kind": "port",
                                   "sample": "%d passes over the first %d instances of the same batch, CPU restatement of the reference SQP+boxADMM "
                                             "(Eigen-like pivoted LDLT, gcc -O2 AVX2), OpenMP over instances on all host threads, %.1f s" % (passes, Bc, tc)}
            xg = d_x.cpu().numpy()[:Bc]
            same = np.array([i.iter for i in io]) == info["iter"][:Bc]
            # the same instances through the restatement in the KERNEL's own elimination order (swept inverse): isolates kernel
            # errors from the last-bit effects of a different, equally valid, order of the linear algebra
            xs, ls_, is_ = ob.sqp_solve_batch(ob.MODEL_ROBOT, wl["P"], wl["S"], wl["t0"], wl["tf"], Bc, wl["d"][:Bc], wl["lbx"][:Bc], wl["ubx"][:Bc],
                                              sqp_settings=oss, pivot=ob.PIVOT_SWEEP, threads=cores)
            same_s = np.array([i.iter for i in is_]) == info["iter"][:Bc]
            kk = lambda f, ii, msk: float(np.abs(info[f][:Bc] - np.array([getattr(i, f) for i in ii]))[msk].max()) if msk.any() else None
            out["parity_vs_cpu_same_order"] = {"same_iteration_count_fraction": float(same_s.mean()),
                                               "max_abs_dx_on_matching": float(np.abs(d_x.cpu().numpy()[:Bc] - xs)[same_s].max()) if same_s.any() else None,
                                               "median_abs_dx_per_instance": float(np.median(np.abs(d_x.cpu().numpy()[:Bc] - xs).max(axis=1))),
                                               "p99_abs_dx_per_instance": float(np.percentile(np.abs(d_x.cpu().numpy()[:Bc] - xs).max(axis=1), 99)),
                                               "note": "identical linear algebra on both sides; the residual difference is sin/cos (device library vs "
                                                       "glibc, last bit) carried through up to 10 SQP iterations",
                                               "max_abs_d_primal_norm": kk("primal_norm", is_, same_s), "max_abs_d_dual_norm": kk("dual_norm", is_, same_s),
                                               "max_abs_d_constraint_violation": kk("max_violation", is_, same_s)}
            kkt = lambda f: float(np.abs(info[f][:Bc] - np.array([getattr(i, f) for i in io]))[same].max()) if same.any() else None
            out["parity_vs_cpu_sample"] = {"same_iteration_count_fraction": float(same.mean()),
                                           "max_abs_dx_on_matching": float(np.abs(xg - xo)[same].max()) if same.any() else None,
                                           "median_abs_dx_per_instance": float(np.median(np.abs(xg - xo).max(axis=1))),
                                           "p99_abs_dx_per_instance": float(np.percentile(np.abs(xg - xo).max(axis=1), 99)),
                                           # the KKT quantities of the termination test (primal / dual step norm, constraint violation)
                                           "max_abs_d_primal_norm": kkt("primal_norm"), "max_abs_d_dual_norm": kkt("dual_norm"),
                                           "max_abs_d_constraint_violation": kkt("max_violation")}
        print(json.dumps(out))
    for c in ctxs:
        c.close()
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
