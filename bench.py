#!/usr/bin/env python
"""bench.py — box-ADMM QP subproblem solves/s of the fused SQP hot path on MI355X.

One "step" = one pass of the hot path over one batch: pmpc_sqp_solve_batch_dev on B = 4096 mobile-robot OCPs per GPU
(BASELINE.json configs[1]: nx=3 nu=2, Chebyshev N=6 -> 7 nodes, n=35, m=21, KKT 56x56, fp64, randomised x0, zero
guesses, SQP max_iter=10 / line search 10, QP settings = SQPBase constructor defaults). Every SQP iteration solves
one box-ADMM QP subproblem (plus its linearisation, Hessian update and line search, all inside the same kernel), so
value = (sum over instances of SQP iterations) * steps / wall time. Inputs are resident in HBM before the timed region.

Multi-GPU: one process per GPU, the batch shards embarrassingly (each rank solves its own 4096 instances — weak
scaling, no data-path collective); torch.distributed (RCCL) is used only for the barrier and the max-over-ranks time.
`python bench.py --gpus N` on its own starts the N ranks itself (torch.distributed.run on 127.0.0.1); launched by a
driver through torch.distributed.run it finds WORLD_SIZE in the environment and must be told the same N.

Beside the headline the JSON line carries (rank 0, N = 1 only): per-step min / median / max, `configs` (BASELINE.json
configs[2..4]: CSTR 16 384, kite stand-in 1024, scenario 8192 per GPU — a few steps each), `qp_replay` (a flat batch
of QPs with the collocation structure through the QP entry point alone), `cpu_baseline` (CPU restatement of the reference,
single core and all usable cores) and two parity objects over the whole batch.

Output (round 5): the DETAILED record goes to a file (`--detail-out`, default gpurun_out/bench_detail.json) and to stderr on one line prefixed
`DETAIL `; stdout carries exactly ONE line, printed last: the compact contract line built by `contract_line()` (< 4 KB — a CPU test pins the
size), which is what the driver parses. The timed region is a BLOCK of exactly `--steps` steps bracketed by barrier + synchronize on both
sides; the block is repeated until at least `--min-seconds` of timed work have run and the MEDIAN block (max over ranks per block) is reported,
so that a 20-step block of 23 ms is not the only evidence of a 48 s run.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_HBM_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)
PEAK_FP64_TFLOPS = 78.6      # fp64 vector = matrix peak: 256 CU x 4 SIMD x 16 lanes x 2 flop x 2.4 GHz


def qp_algorithmic_bytes(n, m):
    """read the QP data once, write the solution once."""
    return 8 * (n * n + m * n + 3 * n + 2 * m) + 8 * (2 * n + m) + 32


def qp_algorithmic_flops(n, m, it, f):
    N = n + m
    chk = it // 10
    return f * N ** 3 / 3.0 + it * (2.0 * N * N + 12.0 * N) + chk * 2.0 * (n * n + 2 * m * n)


def library_build_id():
    """sha256 of the loaded product library: ties a committed PMC summary to the build it was measured on."""
    import hashlib
    import polympc_amd as pa
    h = hashlib.sha256()
    with open(pa.LIB_PATH, "rb") as f:
        for chunk in iter(lambda: f.read(1 << 20), b""):
            h.update(chunk)
    return h.hexdigest()[:16]


def measured_traffic(build_id):
    """HBM bytes per launch of the bench kernel from the committed PMC summary of this round (rocprofv3 --pmc FETCH_SIZE /
    WRITE_SIZE in separate passes, corrected by the calibration kernel; tests/tools_pmc.sh + tests/tools_pmc_summary.py).
    bench.py cannot run rocprofv3 around itself, so it reports the number measured on the same command line — and says whether that
    summary was recorded for the library build that is loaded now (`traffic_build_matches`)."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary.json"))):
        try:
            j = json.load(open(f))
            t = j.get("traffic")
        except Exception:
            t = None
        if t and "RobotOCP,35,21" in str(t.get("kernel", "")).replace(" ", ""):   # the bench kernel's summary only (other configurations have their own)
            best = (t["bytes_per_launch"], os.path.basename(f), j.get("library_build_id"))
    if not best:
        return None
    return {"bytes": best[0], "source": "profiles/" + best[1], "build_matches": (best[2] == build_id) if best[2] else None}


def _r(v, sig=6):
    """Round a float to `sig` significant digits for the compact line (None / bool / int pass through)."""
    if isinstance(v, bool) or v is None or isinstance(v, int):
        return v
    try:
        return float("%.*g" % (sig, float(v)))
    except (TypeError, ValueError):
        return v


def contract_line(out, detail_path=None):
    """The ONE stdout line the driver parses, built from the detailed record `out`: the contract's keys, `roofline`, `cpu_baseline`, two
    one-number parity summaries and {ms, route, frac} per configuration. Everything else lives in the detail file. Must stay < 4096 bytes
    (tests/test_bench_line_cpu.py)."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")
    line = {k: _r(out[k], 9) if isinstance(out.get(k), float) else out.get(k) for k in keep}
    cfg = out.get("config", {})
    line["config"] = {k: cfg[k] for k in ("workload", "global_batch", "parallelism") if k in cfg}
    rf = out.get("roofline")
    if rf:
        line["roofline"] = {k: (_r(rf[k]) if isinstance(rf.get(k), float) else rf.get(k)) for k in
                            ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "traffic_build_matches", "kernel", "kernel_ms") if k in rf}
    if out.get("roofline_fp64"):
        line["roofline_fp64"] = {"frac": _r(out["roofline_fp64"]["frac"]), "peak": out["roofline_fp64"]["peak"], "unit": "TFLOP/s"}
    cb = out.get("cpu_baseline")
    if cb:
        line["cpu_baseline"] = {"value": _r(cb["value"]), "unit": cb["unit"], "cores": cb["cores"], "kind": cb["kind"], "sample": str(cb.get("sample", ""))[:200]}
        if "single_core" in cb:
            line["cpu_baseline"]["single_core"] = {"value": _r(cb["single_core"]["value"])}
    tb = out.get("timed_blocks")
    if tb:
        line["timed_blocks"] = {k: _r(tb[k]) for k in ("blocks", "timed_s", "min_ms_per_step", "max_ms_per_step") if k in tb}
    par = {}
    so = out.get("parity_vs_cpu_same_order")
    if so:
        par["same_order_bit_identical"] = bool(so.get("bit_identical_x") and so.get("bit_identical_lam")); par["same_order_max_abs_dx"] = _r(so.get("max_abs_dx"), 3)
    pr = out.get("parity_vs_cpu_reference")
    if pr:
        par["vs_reference_order_max_abs_dx"] = _r(pr.get("max_abs_dx"), 3); par["vs_reference_order_identical_trajectories"] = _r(pr.get("identical_trajectory_fraction"), 6)
    qp = (out.get("qp_replay") or {}).get("parity_vs_cpu_reference", {}).get("configs")
    if qp:
        par["qp_level_within_1e-8"] = {k: bool(v.get("within_1e-8")) for k, v in qp.items()}
        par["qp_level_max_abs_d_res"] = _r(max(max(v.get("max_abs_d_res_prim", 0.0), v.get("max_abs_d_res_dual", 0.0)) for v in qp.values()), 3)
    if par:
        line["parity"] = par
    if out.get("variant_block_bfgs"):
        line["variant_block_bfgs_ms"] = _r(out["variant_block_bfgs"]["ms_per_step"])
    if out.get("small_batches"):   # config A at B = 1 / 64 / 512: [GPU ms, one host core's ms for the same instances]
        line["small_batches_ms"] = {k: [_r(v["ms_per_batch"]["median"], 4), _r(v.get("cpu_single_core_ms"), 4)] for k, v in out["small_batches"].items()}
    if out.get("gpu_over_cpu_all_cores"):
        line["gpu_over_cpu_all_cores"] = _r(out["gpu_over_cpu_all_cores"]["value"], 4)
    if out.get("qp_replay"):
        q = out["qp_replay"]
        line["qp_replay"] = {"ms": _r(q["ms_per_batch"]["median"]), "qp_solves_per_s": _r(q["qp_solves_per_s"]), "frac": _r(q["roofline_hbm_frac"], 4)}
    cfgs = {}
    for key, c in (out.get("configs") or {}).items():
        e = {"ms": _r(c["ms_per_batch"]["median"]), "route": c.get("route"), "frac": _r(c["roofline_hbm_frac"], 4)}
        vb = c.get("variant_block_bfgs")
        if vb:
            e["block_bfgs"] = {"ms": _r(vb["ms_per_batch"]["median"]), "route": vb.get("route")}
        if "parity_vs_cpu_same_order" in c:
            e["bit_identical"] = bool(c["parity_vs_cpu_same_order"].get("bit_identical_x"))
        if "small_batches" in c:
            e["small_batches_ms"] = {k: _r(v["ms_per_batch"]["median"]) for k, v in c["small_batches"].items()}
        cfgs[key.split("_")[0]] = e
    if cfgs:
        line["configs"] = cfgs
    if out.get("reference_tests"):   # the reference's two NP = 1 control tests as batches of 4096: [ms per batch, lone instance ms, route, bit-identical to the restatement]
        line["reference_tests"] = {k: [_r(v["ms_per_batch"]["median"]), _r(v["lone_instance_ms"]), v.get("route"), (v.get("parity") or {}).get("bit_identical_x")] for k, v in out["reference_tests"].items()}
    line["library_build_id"] = out.get("library_build_id")
    if detail_path:
        line["detail"] = detail_path
    return line


def usable_cpus():
    """Host threads this process may really use: the affinity mask, capped by the cgroup CPU quota (a container can report 256
    CPUs and be allowed 16 of them)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = max(1, min(n, int(float(q) / float(p) + 0.5)))
    except Exception:
        pass
    return n


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks here (one process per GPU, rendezvous on 127.0.0.1)."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ); env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    raise SystemExit(subprocess.call(cmd, env=env))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=4096, help="OCP instances per GPU")
    ap.add_argument("--cpu-sample", type=int, default=4096, help="instances per pass of the CPU-baseline sample (0 = skip)")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="minimum wall time of the all-core CPU-baseline sample (single core: half)")
    ap.add_argument("--configs", default="B,C,D,R,P", help="sub-records beside the headline (N = 1 only): any of B, C, D (BASELINE.json configs[2..4]) R (the reference's own 16-node robot grid, 128 KKT rows) and P (the problems of the reference's two NP = 1 control tests as batches); '' = none")
    ap.add_argument("--no-replay", action="store_true", help="skip the QP-only replay record")
    ap.add_argument("--min-seconds", type=float, default=2.0, help="repeat the timed block of --steps steps until this much timed work has run; the median block is reported (0 = one block)")
    ap.add_argument("--max-blocks", type=int, default=400)
    ap.add_argument("--detail-out", default=os.path.join("gpurun_out", "bench_detail.json"), help="file for the detailed record (rank 0); '' = stderr only")
    ap.add_argument("--streams", type=int, default=1, help="1 (default, the contract's configuration): every step is one launch on one "
                    "stream. S > 1: consecutive steps alternate over S contexts (own stream, workspace and output buffers), so that a "
                    "step's tail overlaps the next step's start — a pipelined-server figure, reported in DESIGN.md, not the bench line")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        spawn_ranks(args)

    import numpy as np
    import torch
    import polympc_amd as pa
    from polympc_amd import workloads, sharding

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: polympc_amd has no CPU fallback")
    # developer switches for exercising the N > 1 code path on a box with ONE GPU (every rank on device 0, gloo for the barrier and
    # the two reductions); the driver's multi-GPU runs use neither: one rank per GPU, nccl (= RCCL)
    backend = os.environ.get("PMPC_BENCH_BACKEND", "nccl")
    if os.environ.get("PMPC_BENCH_SINGLE_DEVICE") == "1":
        local_rank = 0
    elif local_rank >= torch.cuda.device_count():
        raise SystemExit(f"bench.py: rank {rank} needs GPU {local_rank} but only {torch.cuda.device_count()} are visible")
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
    def timed_block():
        """EXACTLY --steps steps, bracketed by barrier + synchronize on both sides; HIP events around every step on its launch stream."""
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
        el = time.perf_counter() - t0
        return el, [e0.elapsed_time(e1) for e0, e1 in evs]

    el0, ms0 = timed_block()
    # how many more blocks: decided from the first block's MAX over ranks, so that every rank runs the same number (the barriers must pair up)
    el0_max = sharding.combine_stats(dist, red_dev, [0], el0)[1]
    more = 0 if args.min_seconds <= 0 else max(0, min(args.max_blocks - 1, int(np.ceil(args.min_seconds / max(el0_max, 1e-6))) - 1))
    block_el, block_ms = [el0], [ms0]
    for _ in range(more):
        el, ms_ = timed_block()
        block_el.append(el); block_ms.append(ms_)
    block_el = sharding.max_over_ranks(dist, red_dev, block_el)      # per block: the slowest rank's wall time
    order = np.argsort(block_el)
    med = int(order[(len(order) - 1) // 2])                          # the median block (lower median: a block that was really run)
    elapsed = float(block_el[med])
    step_ms = np.array(block_ms[med])                                 # HIP events on the launch stream: one kernel launch per step
    kernel_ms = float(np.mean(step_ms))                               # average launch duration over the MEDIAN block's steps: the dominant kernel cannot exceed the step it is part of (kernel_ms <= ms_per_step)
    kernel_ms_all_blocks = float(np.mean([np.mean(b) for b in block_ms]))   # the same average over every timed block (what rocprofv3 --stats averages; carries the slow blocks of a noisy box)

    info = np.frombuffer(d_info.cpu().numpy().tobytes(), dtype=pa.capi.SQP_INFO_DTYPE)
    qp_solves = int(info["iter"].sum())
    admm_iters = int(info["qp_solver_iter"].sum())
    solved = int((info["status"] == pa.SQP_SOLVED).sum())
    (qp_all, admm_all, solved_all), _ = sharding.combine_stats(dist, red_dev, [qp_solves, admm_iters, solved], elapsed)

    sol = [None]

    def sqp_record(cwl, Bc, steps, warmup, kernel_name, **settings):
        """A few launches of another configuration on this rank's context: ms per batch (median of HIP-event times), QP/s, rooflines."""
        cn, cm = cwl["n"], cwl["m"]
        css = pa.sqp_settings_default(); css.max_iter = cwl["max_iter"]; css.line_search_max_iter = cwl["ls_max_iter"]
        for k, v in {**cwl.get("settings", {}), **settings}.items():   # (a workload's own solver configuration, e.g. the reference tests' exact Hessians + Gershgorin shift)
            setattr(css, k, v)
        cd, cl, cu = t(cwl["d"]), t(cwl["lbx"]), t(cwl["ubx"])
        extra = {k: t(cwl[k]) for k in ("x_guess", "lbg", "ubg") if k in cwl}
        cqs = qs
        if "qp_max_iter" in cwl:
            cqs = pa.qp_settings_sqp_default(); cqs.max_iter = cwl["qp_max_iter"]
        cx = torch.zeros(Bc, cn, dtype=torch.float64, device=dev); clam = torch.zeros(Bc, cm + cn, dtype=torch.float64, device=dev)
        ci = torch.zeros(Bc, 48, dtype=torch.uint8, device=dev)
        run = lambda: ctx.sqp_solve_batch_dev(cwl["model"], cwl["P"], cwl["S"], cwl["t0"], cwl["tf"], Bc, cd, cl, cu, cx, clam, ci, css, cqs, **extra)
        for _ in range(warmup):
            run()
        torch.cuda.synchronize(dev)
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
        for a, b in ev:
            a.record(stream); run(); b.record(stream)
        torch.cuda.synchronize(dev)
        ms = np.array([a.elapsed_time(b) for a, b in ev])
        inf = np.frombuffer(ci.cpu().numpy().tobytes(), dtype=pa.capi.SQP_INFO_DTYPE)
        qps = int(inf["iter"].sum()); its = int(inf["qp_solver_iter"].sum())
        med = float(np.median(ms)) * 1e-3
        sol[0] = (cx.cpu().numpy(), clam.cpu().numpy(), inf)   # for the parity objects of the CPU leg
        return {"batch": Bc, "n": cn, "m": cm, "kkt_rows": cn + cm, "steps": steps, "kernel": kernel_name,
                "ms_per_batch": {"min": float(ms.min()), "median": float(np.median(ms)), "max": float(ms.max())},
                "qp_solves_per_s": qps / med, "sqp_solves_per_s": Bc / med, "qp_solves_per_batch": qps, "admm_iters_per_qp": its / max(qps, 1),
                "sqp_solved_fraction": float((inf["status"] == pa.SQP_SOLVED).mean()), "route": pa.capi.ROUTE_NAMES.get(ctx.last_route(), "?"),
                "roofline_hbm_frac": qp_algorithmic_bytes(cn, cm) * qps / med / 1e9 / PEAK_HBM_GBS,
                "roofline_fp64_frac": qp_algorithmic_flops(cn, cm, its / max(qps, 1), 1.0) * qps / med / 1e12 / PEAK_FP64_TFLOPS}

    if rank == 0:
        build_id = library_build_id()
        value = qp_all * args.steps / elapsed
        it_per_qp = admm_iters / max(qp_solves, 1)
        # roofline of the dominant kernel (sqp_kernel<RobotOCP>), per launch on this rank
        bytes_launch = qp_algorithmic_bytes(n, m) * qp_solves
        flops_launch = qp_algorithmic_flops(n, m, it_per_qp, 1.0) * qp_solves
        ach_gbs = bytes_launch / (kernel_ms * 1e-3) / 1e9
        ach_tf = flops_launch / (kernel_ms * 1e-3) / 1e12
        tr = measured_traffic(build_id) if B == 4096 else None
        out = {
            "metric": "box-ADMM QP subproblem solves/s (fused SQP hot path)", "value": value, "unit": "QP solves/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "mobile_robot OCP nx=3 nu=2 N=6 (P=6,S=1; n=35 m=21 KKT 56), batch=%d per GPU, randomised x0, fp64, "
                                   "SQP max_iter=10 ls=10, QP = SQPBase defaults" % B,
                       "global_batch": B * world, "parallelism": "batch-shard x%d (no collectives)" % world,
                       **({"streams": S_, "note": "steps pipelined over %d streams: NOT the contract's configuration" % S_} if S_ > 1 else {})},
            "timed_blocks": {"blocks": len(block_el), "timed_s": float(np.sum(block_el)), "min_ms_per_step": float(np.min(block_el)) / args.steps * 1e3,
                             "max_ms_per_step": float(np.max(block_el)) / args.steps * 1e3, "median_block": med,
                             "note": "each block = exactly --steps steps between barrier + synchronize; value / ms_per_step are the MEDIAN block's (max over ranks per block)"},
            "step_ms": {"min": float(step_ms.min()), "median": float(np.median(step_ms)), "max": float(step_ms.max()), "mean": kernel_ms,
                        "mean_all_blocks": kernel_ms_all_blocks,
                        "note": "HIP events around each step's launches on the launch stream (rank 0), the median block's steps; mean_all_blocks: every timed block"},
            "sqp_solves_per_s": B * world * args.steps / elapsed, "qp_solves_per_step": qp_all, "admm_iters_per_qp": admm_all / max(qp_all, 1),
            "sqp_solved_fraction": solved_all / (B * world), "library_build_id": build_id,
            "roofline": {"bound": "hbm", "achieved": ach_gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": ach_gbs / PEAK_HBM_GBS,
                         "traffic": (tr["bytes"] if tr else None), "traffic_source": (tr["source"] if tr else None),
                         "traffic_build_matches": (tr["build_matches"] if tr else None),
                         "algorithmic_bytes_per_launch": bytes_launch, "kernel": "sqp_kernel<RobotOCP,35,21>", "kernel_ms": kernel_ms,
                         "note": "algorithmic bytes = 17616 B per QP subproblem x QPs per launch (SURVEY 8d); the path is bound by fp64 issue / "
                                 "dependent-chain latency, not by HBM: see roofline_fp64 and DESIGN.md"},
            "roofline_fp64": {"bound": "fp64-valu", "achieved": ach_tf, "peak": PEAK_FP64_TFLOPS, "unit": "TFLOP/s", "frac": ach_tf / PEAK_FP64_TFLOPS},
        }
        if world == 1:
            # the same batch with the Hessian update the reference's own mobile-robot MPC test plugs in (mpc_wrapper_test.cpp:100-105:
            # ContinuousOCP's block BFGS). Reported beside the bench line, which stays on SQPBase's defaults (dense damped BFGS).
            v = sqp_record(wl, B, max(3, args.steps // 2), args.warmup, "sqp_kernel<RobotOCP,35,21,HU=1>", hessian_update=1)
            out["variant_block_bfgs"] = {"settings": "hessian_update = 1 (continuous_ocp.hpp:2304-2431), everything else as the bench line",
                                         "ms_per_step": v["ms_per_batch"]["median"], "qp_solves_per_s": v["qp_solves_per_s"],
                                         "sqp_solves_per_s": v["sqp_solves_per_s"], "sqp_solved_fraction": v["sqp_solved_fraction"]}
            # ---- BASELINE.json configs[0] is ONE Solver<OCP>::solve(): small batches of the headline workload (a lone instance is what a receding-horizon
            # controller waits for); the single-core CPU time of the same instances is added by the CPU leg below (the honest crossover)
            sbA = {}
            for Bs in (1, 64, 512):
                r_ = sqp_record(workloads.robot_batch(Bs), Bs, 20, 3, "sqp_kernel<RobotOCP,35,21>")
                sbA[str(Bs)] = {"ms_per_batch": r_["ms_per_batch"], "qp_solves_per_batch": r_["qp_solves_per_batch"], "qp_solves_per_s": r_["qp_solves_per_s"],
                                "sqp_solved_fraction": r_["sqp_solved_fraction"], "route": r_["route"]}
            out["small_batches"] = sbA
            # ---- BASELINE.json configs[2..4] (a few launches each; the headline above stays configs[1])
            want = [c for c in args.configs.split(",") if c]
            cfg = {}
            cfg_runs = {}   # key -> (letter, workload, batch, GPU solution) for the CPU leg below
            cfg_ref_runs = {}   # key -> (workload, GPU solution) of the reference-test problems
            def add(key, letter, cwl, Bc, steps, warmup, kernel_name, workload):
                cfg[key] = sqp_record(cwl, Bc, steps, warmup, kernel_name)
                cfg[key]["workload"] = workload
                base_sol = sol[0]
                # the same batch with the Hessian update every control test of the reference selects (cstr_control_test.cpp:128-132,
                # mpc_wrapper_test.cpp:100-105, minimal_time_test.cpp:84-88, valet_parking_mpc_test.cpp:161-165 -> continuous_ocp.hpp:2304-2431)
                vb = sqp_record(cwl, Bc, max(3, steps // 2), 1, kernel_name + ", hessian_update = 1", hessian_update=1)
                cfg[key]["variant_block_bfgs"] = {k_: vb[k_] for k_ in ("steps", "ms_per_batch", "qp_solves_per_s", "sqp_solves_per_s", "qp_solves_per_batch",
                                                                          "admm_iters_per_qp", "sqp_solved_fraction", "route")}
                cfg_runs[key] = (letter, cwl, Bc, base_sol, sol[0])
            if "D" in want:
                add("D_scenario_8192_per_gpu", "D", workloads.robot_batch(8192, perturb_d=True, first=5000), 8192, 10, 2, "sqp_kernel<RobotOCP,35,21>",
                    "mobile robot, perturbed wheel base d = 2(1+0.1U), 8192 instances per GPU (65 536 over 8 GPUs)")
            if "B" in want:
                add("B_cstr_16384", "B", workloads.cstr_batch(16384), 16384, 10, 1, "sqp_kernel<CstrOCP,66,44> (110 KKT rows)",
                    "CSTR nx=4 nu=2, P=5 S=2 (11 nodes), t in [0,100], SQP max_iter=20 ls=20")
            if "C" in want:
                add("C_kite_standin_1024", "C", workloads.kite_standin_batch(1024), 1024, 8, 1, "sqp_kernel<KiteStandInOCP> (464 KKT rows)",
                    "SYNTHETIC 13-state / 3-input stand-in (the reference tree has no kite model), P=5 S=3 (16 nodes), SQP max_iter=5")
            if "C" in want:
                # small batches of the large instance: a lone instance is what a receding-horizon controller waits for (the four-wavefront team kernel, BigTeam)
                sb = {}
                for Bs in (1, 256):
                    r_ = sqp_record(workloads.kite_standin_batch(Bs), Bs, 8, 1, "sqp_kernel<KiteStandInOCP, WG4> (one workgroup of four wavefronts per instance)")
                    sb[str(Bs)] = {"ms_per_batch": r_["ms_per_batch"], "qp_solves_per_s": r_["qp_solves_per_s"], "sqp_solved_fraction": r_["sqp_solved_fraction"], "route": r_["route"]}
                cfg["C_kite_standin_1024"]["small_batches"] = sb
            if "R" in want:
                add("R_robot_16_nodes_2048", "R", workloads.robot_batch(2048, P=5, S=3), 2048, 10, 1, "sqp_kernel<RobotOCP> (128 KKT rows)",
                    "mobile robot on the reference's mpc_wrapper_test grid, P=5 S=3 (16 nodes, n=80, m=48), 2048 instances (not a BASELINE.json configuration: the mid-size path)")
            if cfg:
                out["configs"] = cfg
            if "P" in want:
                # ---- the reference's control tests that are no BASELINE.json configuration, as batches: minimal_time_test.cpp:146-188 and nonlinear_constraints_test.cpp:159-184 (NP = 1)
                # and the policy set of valet_parking_mpc_test.cpp (Ruiz + filter + block BFGS), configured as those tests configure the solver; a lone instance beside each (what the
                # reference's test solves). (cstr_control_test / mpc_wrapper_test: configs B / R with the block BFGS, above.) Checked against the restatement in the CPU leg.
                rt = {}
                for key, pc in (("minimal_time_parking_np1", False), ("nonlinear_constraints_parking_np1_ng1", True), ("valet_parking_policy_set_robot_11_nodes", None)):
                    mk = (lambda Bq: workloads.valet_parking_policy_batch(Bq)) if pc is None else (lambda Bq: workloads.parking_reference_tests_batch(Bq, path_constraint=pc))
                    rwl = mk(4096)
                    r_ = sqp_record(rwl, 4096, 6, 1, "sqp_kernel<...,CND>")
                    gsol_ = sol[0]
                    l_ = sqp_record(mk(1), 1, 10, 2, "")
                    rt[key] = {"batch": 4096, "n": rwl["n"], "m": rwl["m"], "ms_per_batch": r_["ms_per_batch"], "qp_solves_per_s": r_["qp_solves_per_s"], "qp_solves_per_batch": r_["qp_solves_per_batch"],
                               "admm_iters_per_qp": r_["admm_iters_per_qp"], "sqp_solved_fraction": r_["sqp_solved_fraction"], "route": r_["route"],
                               "lone_instance_ms": l_["ms_per_batch"]["median"], "lone_instance_route": l_["route"],
                               "settings": ("Ruiz preconditioner + filter line search + block BFGS, QP cap 1000 (valet_parking_mpc_test.cpp:161-165,183-241) on randomised robot OCPs of that test's grid" if pc is None else
                                            "max_iter 20, ls 10, exact Hessians every iteration, Gershgorin shift (as the reference's tests configure the solver); perturbed start states and wheel bases")}
                    cfg_ref_runs[key] = (rwl, gsol_)
                out["reference_tests"] = rt
            if not args.no_replay:
                # ---- QP-only replay: a flat batch of QPs with the collocation structure, through pmpc_qp_boxadmm_solve_batch_dev.
                # The QPs are built by the PRODUCT (pmpc_ocp_linearise_batch at seeded random points, lam = 0): H = cost Hessian, A = collocation
                # Jacobian, h = cost gradient, equality bounds -c, input boxes shifted by the point.
                Bq = 16384
                rng = np.random.default_rng(workloads.SEED)
                wq = workloads.robot_batch(Bq)
                pts = rng.uniform(-0.5, 0.5, size=(Bq, n))
                lin = ctx.ocp_linearise_batch(wq["model"], wq["P"], wq["S"], wq["t0"], wq["tf"], pts, wq["d"])
                Hq = t(lin["lag_hess"].transpose(0, 2, 1).reshape(Bq, n * n)); Aq = t(lin["jac"].transpose(0, 2, 1).reshape(Bq, m * n))
                hq = t(lin["cost_grad"]); alq = t(-lin["c"]); auq = t(-lin["c"])
                lxq = t(wq["lbx"] - pts); uxq = t(wq["ubx"] - pts)
                xq = torch.zeros(Bq, n, dtype=torch.float64, device=dev); yq = torch.zeros(Bq, n + m, dtype=torch.float64, device=dev)
                iq = torch.zeros(Bq, 40, dtype=torch.uint8, device=dev)
                runq = lambda: ctx.qp_solve_batch_dev(Bq, n, m, Hq, hq, Aq, alq, auq, lxq, uxq, xq, yq, iq, qs)
                for _ in range(2):
                    runq()
                torch.cuda.synchronize(dev)
                ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
                for a, b in ev:
                    a.record(stream); runq(); b.record(stream)
                torch.cuda.synchronize(dev)
                ms = np.array([a.elapsed_time(b) for a, b in ev]); med = float(np.median(ms)) * 1e-3
                qinf = np.frombuffer(iq.cpu().numpy().tobytes(), dtype=pa.capi.QP_INFO_DTYPE)
                itq = float(qinf["iter"].mean()); fq = float(qinf["rho_updates"].mean())
                out["qp_replay"] = {"what": "pmpc_qp_boxadmm_solve_batch_dev on %d QPs (n=35, m=21) linearised from the config-A OCP at seeded random points; "
                                            "H, A, bounds read from HBM, solution written back" % Bq, "kernel": "qp_boxadmm_reg_kernel<35,21>",
                                    "ms_per_batch": {"min": float(ms.min()), "median": float(np.median(ms)), "max": float(ms.max())},
                                    "qp_solves_per_s": Bq / med, "admm_iters_per_qp": itq, "factorisations_per_qp": fq,
                                    "solved_fraction": float((qinf["status"] == pa.QP_SOLVED).mean()),
                                    "roofline_hbm_frac": qp_algorithmic_bytes(n, m) * Bq / med / 1e9 / PEAK_HBM_GBS,
                                    "roofline_fp64_frac": qp_algorithmic_flops(n, m, itq, fq) * Bq / med / 1e12 / PEAK_FP64_TFLOPS}
        if args.cpu_sample > 0 and world == 1:
            from oracle import binding as ob   # CPU restatement of the reference algorithm: baseline / checker only, never the product path
            Bc = min(args.cpu_sample, B)
            cores = usable_cpus()
            oss = ob.sqp_default_settings(); oss.max_iter = wl["max_iter"]; oss.line_search_max_iter = wl["ls_max_iter"]

            def cpu_run(count, threads, pivot, libm):
                with (ob.libm() if libm else _null()):
                    return ob.sqp_solve_batch(ob.MODEL_ROBOT, wl["P"], wl["S"], wl["t0"], wl["tf"], count, wl["d"][:count], wl["lbx"][:count],
                                              wl["ubx"][:count], sqp_settings=oss, pivot=pivot, threads=threads)

            def cpu_rate(count, threads, seconds):
                rates, tc, qps, passes = [], 0.0, 0, 0
                while tc < seconds and passes < 2000:   # bounded sample: repeated passes over the same instances
                    t1 = time.perf_counter()
                    _, _, io_ = cpu_run(count, threads, ob.PIVOT_EIGEN, True)
                    dt = time.perf_counter() - t1
                    q = sum(i.iter for i in io_)
                    rates.append(q / dt); tc += dt; qps += q; passes += 1
                return float(np.median(rates)), passes, tc

            r1, p1, t1_ = cpu_rate(min(256, Bc), 1, args.cpu_seconds / 2)
            rN, pN, tN_ = cpu_rate(Bc, cores, args.cpu_seconds)
            out["cpu_baseline"] = {"value": rN, "unit": "QP solves/s", "cores": cores, "kind": "port",
                                   "single_core": {"value": r1, "unit": "QP solves/s", "cores": 1,
                                                   "sample": "median of %d passes over the first %d instances, %.1f s" % (p1, min(256, Bc), t1_)},
                                   "parallel_efficiency": rN / (r1 * cores),
                                   "host": {"os_cpu_count": os.cpu_count(), "usable_threads": cores},
                                   "sample": "median of %d passes over the first %d instances of the same batch, CPU restatement of the reference SQP+boxADMM "
                                             "(Eigen-like pivoted LDLT, glibc sin/cos, g++ -O3 AVX2+FMA), OpenMP (dynamic chunks of 4 instances) on %d threads "
                                             "(affinity mask capped by the cgroup quota), %.1f s" % (pN, Bc, cores, tN_)}
            # baseline only — a large GPU / CPU ratio says nothing about kernel quality (the roofline fraction does)
            out["gpu_over_cpu_all_cores"] = {"value": value / rN, "cores": cores, "note": "baseline only: headline QP solves/s over the CPU restatement on all usable host threads of this box"}
            for Bs_, rec_ in (out.get("small_batches") or {}).items():   # the same instances on ONE host core: where a lone solve is better off on the CPU
                nb = int(Bs_); reps = max(1, min(50, 2048 // nb))
                t1 = time.perf_counter()
                for _ in range(reps):
                    cpu_run(nb, 1, ob.PIVOT_EIGEN, True)
                rec_["cpu_single_core_ms"] = (time.perf_counter() - t1) / reps * 1e3
                rec_["gpu_over_cpu_single_core"] = rec_["cpu_single_core_ms"] / rec_["ms_per_batch"]["median"]
            xg = d_x.cpu().numpy()[:Bc]; lg = d_lam.cpu().numpy()[:Bc]

            def parity(xo, lo, io_, note):
                it_o = np.array([i.iter for i in io_]); st_o = np.array([i.status for i in io_]); qi_o = np.array([i.qp_solver_iter for i in io_])
                same = (it_o == info["iter"][:Bc]) & (st_o == info["status"][:Bc]) & (qi_o == info["qp_solver_iter"][:Bc])
                dxi = np.abs(xg - xo).max(axis=1)
                kk = lambda f: float(np.abs(info[f][:Bc] - np.array([getattr(i, f) for i in io_])).max())
                return {"instances": Bc, "identical_trajectory_fraction": float(same.mean()),   # SQP iterations, status and total ADMM iterations
                        "bit_identical_x": bool(np.array_equal(xg, xo)), "bit_identical_lam": bool(np.array_equal(lg, lo)),
                        "max_abs_dx": float(dxi.max()), "median_abs_dx_per_instance": float(np.median(dxi)), "p99_abs_dx_per_instance": float(np.percentile(dxi, 99)),
                        "max_abs_dlam": float(np.abs(lg - lo).max()), "max_abs_d_primal_norm": kk("primal_norm"), "max_abs_d_dual_norm": kk("dual_norm"),
                        "max_abs_d_constraint_violation": kk("max_violation"), "max_abs_d_cost": kk("cost"), "note": note}

            # ---- the same two objects + a CPU baseline beside every sub-record ("the reference CPU path timed next to it"): bounded samples
            from polympc_amd.parity_stats import cross_order_stats
            SAMPLE = {"D": (4096, 256), "B": (2048, 64), "C": (64, 8), "R": (1024, 128)}   # instances of the (all-core, single-core) CPU samples / parity objects
            for key, (letter, cwl, Bfull, gsol, vsol) in (cfg_runs.items() if world == 1 and "configs" in out else []):
                n_all, n_one = SAMPLE[letter]
                coss = ob.sqp_default_settings(); coss.max_iter = cwl["max_iter"]; coss.line_search_max_iter = cwl["ls_max_iter"]
                rows = cwl["n"] + cwl["m"]
                korder = ob.PIVOT_SWEEP if rows <= 64 else (ob.PIVOT_SWEEP2 if rows <= ob.SWEEP2_MAX_ROWS else ob.PIVOT_CONDENSED)
                if cfg[key].get("route") == "condreg":   # the kernel that served it decides the restated order (condensed register kernel: PIVOT_CONDSWEEP)
                    korder = ob.PIVOT_CONDSWEEP
                korder_default = korder

                def crun(count, threads, pivot, glibc, hu=0):
                    coss.hessian_update = hu
                    with (ob.libm() if glibc else _null()):
                        return ob.sqp_solve_batch(cwl["model"], cwl["P"], cwl["S"], cwl["t0"], cwl["tf"], count, cwl["d"][:count], cwl["lbx"][:count],
                                                  cwl["ubx"][:count], sqp_settings=coss, pivot=pivot, threads=threads)

                def crate(count, threads, seconds):
                    rates, tc, passes = [], 0.0, 0
                    while tc < seconds and passes < 200:
                        t1 = time.perf_counter(); _, _, io_ = crun(count, threads, ob.PIVOT_EIGEN, True); dt = time.perf_counter() - t1
                        rates.append(sum(i.iter for i in io_) / dt); tc += dt; passes += 1
                    return float(np.median(rates)), passes, tc

                r1c, p1c, t1c = crate(n_one, 1, 3.0)
                rNc, pNc, tNc = crate(n_all, cores, 3.0)
                cfg[key]["cpu_baseline"] = {"value": rNc, "unit": "QP solves/s", "cores": cores, "kind": "port",
                                            "single_core": {"value": r1c, "unit": "QP solves/s", "cores": 1, "sample": "median of %d passes over the first %d instances, %.1f s" % (p1c, n_one, t1c)},
                                            "sample": "median of %d passes over the first %d instances, Eigen-style pivoted LDLT + glibc, OpenMP on %d threads, %.1f s" % (pNc, n_all, cores, tNc)}
                gx, gl, gi = gsol
                xr, lr, ir = crun(n_all, cores, ob.PIVOT_EIGEN, True)
                pr = cross_order_stats(letter, cwl, gx[:n_all], gl[:n_all], gi[:n_all], xr, lr, ir)
                pr["note"] = "GPU (default kernels) vs the CPU restatement as the reference computes (Eigen-style pivoted LDLT, glibc), first %d instances, no mask" % n_all
                cfg[key]["parity_vs_cpu_reference"] = pr
                xk, lk, ik = crun(n_all, cores, korder, False)
                it_k = np.array([i.iter for i in ik]); qi_k = np.array([i.qp_solver_iter for i in ik])
                # the block-BFGS variant: the kernel that served it decides the restated order (block-structured kernel: PIVOT_SCHUR)
                vx, vl, vi = vsol
                nv = max(32, n_all // 2)
                vroute = cfg[key]["variant_block_bfgs"]["route"]
                vorder = ob.PIVOT_SCHUR if vroute == "schur" else (ob.PIVOT_CONDSWEEP if vroute == "condreg" else
                                                                    (ob.PIVOT_SWEEP if rows <= 64 else (ob.PIVOT_SWEEP2 if rows <= ob.SWEEP2_MAX_ROWS else ob.PIVOT_CONDENSED)))
                xk2, lk2, ik2 = crun(nv, cores, vorder, False, hu=1)
                xr2, lr2, ir2 = crun(nv, cores, ob.PIVOT_EIGEN, True, hu=1)
                it2 = np.array([i.iter for i in ik2]); qi2 = np.array([i.qp_solver_iter for i in ik2])
                cfg[key]["variant_block_bfgs"]["parity_vs_cpu_same_order"] = {
                    "instances": nv, "order": int(vorder), "identical_trajectory_fraction": float(((it2 == vi["iter"][:nv]) & (qi2 == vi["qp_solver_iter"][:nv])).mean()),
                    "bit_identical_x": bool(np.array_equal(vx[:nv], xk2)), "bit_identical_lam": bool(np.array_equal(vl[:nv], lk2)), "max_abs_dx": float(np.abs(vx[:nv] - xk2).max())}
                cfg[key]["variant_block_bfgs"]["parity_vs_cpu_reference"] = cross_order_stats(letter, cwl, vx[:nv], vl[:nv], vi[:nv], xr2, lr2, ir2)
                coss.hessian_update = 0
                cfg[key]["parity_vs_cpu_same_order"] = {"instances": n_all, "order": int(korder),
                                                        "identical_trajectory_fraction": float(((it_k == gi["iter"][:n_all]) & (qi_k == gi["qp_solver_iter"][:n_all])).mean()),
                                                        "bit_identical_x": bool(np.array_equal(gx[:n_all], xk)), "bit_identical_lam": bool(np.array_equal(gl[:n_all], lk)),
                                                        "max_abs_dx": float(np.abs(gx[:n_all] - xk).max())}
            for key, (rwl, (gx, gl, gi)) in (cfg_ref_runs.items() if world == 1 and "reference_tests" in out else []):
                # the reference-test problems: the first 32 instances against the restatement in the kernel's order (bit for bit) and as the reference computes
                nr = 32
                ross = ob.sqp_default_settings(); ross.max_iter = rwl["max_iter"]; ross.line_search_max_iter = rwl["ls_max_iter"]
                for k_, v_ in rwl["settings"].items(): setattr(ross, k_, v_)
                kw_ = {k_: rwl[k_][:nr] for k_ in ("x_guess", "lbg", "ubg") if k_ in rwl}
                if "qp_max_iter" in rwl:
                    kw_["qp_settings"] = ob.sqp_qp_default_settings(); kw_["qp_settings"].max_iter = rwl["qp_max_iter"]
                rrun = lambda piv: ob.sqp_solve_batch(rwl["model"], rwl["P"], rwl["S"], rwl["t0"], rwl["tf"], nr, rwl["d"][:nr], rwl["lbx"][:nr], rwl["ubx"][:nr], sqp_settings=ross, pivot=piv, threads=cores, **kw_)
                rorder = ob.PIVOT_CONDSWEEP if out["reference_tests"][key]["route"] == "condreg" else ob.PIVOT_SWEEP2
                xk, lk, ik = rrun(rorder)
                with ob.libm():
                    xr, lr, ir = rrun(ob.PIVOT_EIGEN)
                same = lambda io_: float(np.mean([(a.iter == b_ and a.qp_solver_iter == c_) for a, b_, c_ in zip(io_, gi["iter"][:nr], gi["qp_solver_iter"][:nr])]))
                out["reference_tests"][key]["parity"] = {"instances": nr, "order": int(rorder), "bit_identical_x": bool(np.array_equal(gx[:nr], xk)), "bit_identical_lam": bool(np.array_equal(gl[:nr], lk)),
                                                         "identical_trajectory_fraction_vs_reference_order": same(ir), "max_abs_dx_vs_reference_order": float(np.abs(gx[:nr] - xr).max())}
            if "qp_replay" in out:
                # ---- north_star's criterion on its own unit (one box-ADMM solve; SURVEY 8d: "max |D| of (x, y, res_prim, res_dual) GPU-vs-CPU"): the QPs the
                # reference-order SQP emits for every configuration, through pmpc_qp_boxadmm_solve_batch (default kernels) against PIVOT_EIGEN. Not timed.
                from oracle import cross_order as tco
                qpar = {}
                for letter, nq in (("A", 4096), ("D", 1024), ("B", 2048), ("R", 1024), ("C", 128)):
                    if letter != "A" and letter not in [c for c in args.configs.split(",") if c]:
                        continue
                    qq = tco.traced_qp_stream(ob, letter, nq)
                    gx_, gy_, gi_ = ctx.qp_solve_batch(qq["H"], qq["h"], qq["A"], qq["Alb"], qq["Aub"], qq["xlb"], qq["xub"], settings=qs)
                    xr_, yr_, ir_ = tco.reference_qp_solve(ob, qq, threads=cores)
                    rec_ = tco.qp_level_stats(gx_, gy_, gi_, xr_, yr_, ir_)
                    rec_["n"], rec_["m"], rec_["instances_traced"] = qq["n"], qq["m"], qq["instances"]
                    rec_["within_1e-8"] = bool(rec_["different_iter"] == 0 and rec_["different_status"] == 0 and rec_["max_abs_d_res_prim"] <= 1e-8 and rec_["max_abs_d_res_dual"] <= 1e-8)
                    qpar[letter] = rec_
                out["qp_replay"]["parity_vs_cpu_reference"] = {
                    "what": "QPs of the reference-order SQP trajectories (Eigen-style pivoted LDLT, glibc), each solved once by the default GPU kernel of its "
                            "size at the QP entry point and once by the restatement as the reference computes (PIVOT_EIGEN): every QP, no mask; res_prim / "
                            "res_dual as defined at qp_base.hpp:240-252, box_admm.hpp:398-431", "configs": qpar}
            xs, ls_, is_ = cpu_run(Bc, cores, ob.PIVOT_SWEEP, False)
            out["parity_vs_cpu_same_order"] = parity(xs, ls_, is_, "CPU restatement in the kernel's own elimination order and with the shared IEEE-only sin/cos "
                                                                   "(pmpc_math.hpp): identical arithmetic on both sides, every instance, no mask")
            xo, lo, io = cpu_run(Bc, cores, ob.PIVOT_EIGEN, True)
            out["parity_vs_cpu_reference"] = parity(xo, lo, io, "CPU restatement as the reference computes: Eigen-style pivoted LDLT and glibc sin/cos — a different, "
                                                                "equally valid, order of the linear algebra and last-bit differences in sin/cos; every instance, no mask")
            out["parity_vs_cpu_reference"]["record"] = cross_order_stats("A", wl, xg, lg, info[:Bc], xo, lo, io)   # the object the sub-records and the tests share
        detail_path = None
        if args.detail_out:
            try:
                os.makedirs(os.path.dirname(os.path.abspath(args.detail_out)), exist_ok=True)
                with open(args.detail_out, "w") as f:
                    json.dump(out, f)
                detail_path = args.detail_out
            except OSError:
                detail_path = None
        print("DETAIL " + json.dumps(out), file=sys.stderr, flush=True)
        sys.stdout.flush()
        print(json.dumps(contract_line(out, detail_path)), flush=True)   # the LAST stdout line, and the only one: what the driver parses
    for c in ctxs:
        c.close()
    if dist:
        dist.destroy_process_group()


class _null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


if __name__ == "__main__":
    main()
