"""Developer tool (not a test, not the bench line): device-resident timing of the fused SQP kernel on the other
BASELINE.json configurations (B: CSTR batch 16384, C: kite stand-in batch 1024, D: perturbed robots 8192 per GPU).
Run on a GPU box:  python tests/tools_config_bench.py [A|B|C|D ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import polympc_amd as pa
from polympc_amd import workloads

CONFIGS = {
    "A": lambda: workloads.robot_batch(int(os.environ.get("BA", 4096))),
    "R": lambda: workloads.robot_batch(int(os.environ.get("BA", 4096)), P=int(os.environ.get("P", 5)), S=int(os.environ.get("S", 3))),   # any robot grid: P=5 S=3 -> 128 KKT rows (the reference's mpc_wrapper_test size)
    "B": lambda: workloads.cstr_batch(int(os.environ.get("BB", 16384))),
    "C": lambda: workloads.kite_standin_batch(int(os.environ.get("BC", 1024))),
    "D": lambda: workloads.robot_batch(8192, perturb_d=True),
}
dev = torch.device("cuda", 0)
stream = torch.cuda.Stream(dev)
torch.cuda.set_stream(stream)
ctx = pa.Context(0, stream=stream.cuda_stream)
for name in (sys.argv[1:] or ["A", "B", "C", "D"]):
    wl = CONFIGS[name]()
    B = wl["lbx"].shape[0]; n, m = wl["n"], wl["m"]
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    d_d, d_lbx, d_ubx = t(wl["d"]), t(wl["lbx"]), t(wl["ubx"])
    d_x = torch.zeros(B, n, dtype=torch.float64, device=dev)
    d_lam = torch.zeros(B, m + n, dtype=torch.float64, device=dev)
    d_info = torch.zeros(B, 48, dtype=torch.uint8, device=dev)
    ss = pa.sqp_settings_default(); ss.max_iter = wl["max_iter"]; ss.line_search_max_iter = wl["ls_max_iter"]
    for fld in ("hessian_update", "preconditioner", "qp_solver", "line_search"):   # policy flags: HESSIAN_UPDATE=1 python tests/tools_config_bench.py A
        if fld.upper() in os.environ: setattr(ss, fld, int(os.environ[fld.upper()]))
    qs = pa.qp_settings_sqp_default()
    step = lambda: ctx.sqp_solve_batch_dev(wl["model"], wl["P"], wl["S"], wl["t0"], wl["tf"], B, d_d, d_lbx, d_ubx, d_x, d_lam, d_info, ss, qs)
    step(); torch.cuda.synchronize(dev)
    reps = int(os.environ.get("REPS", 3))
    t0 = time.perf_counter()
    for _ in range(reps):
        step()
    torch.cuda.synchronize(dev)
    ms = (time.perf_counter() - t0) / reps * 1e3
    info = np.frombuffer(d_info.cpu().numpy().tobytes(), dtype=pa.capi.SQP_INFO_DTYPE)
    qps = int(info["iter"].sum())
    print(f"config {name}: B={B} n={n} m={m}  {ms:9.2f} ms/step  {qps / ms * 1e3:12.0f} QP/s  {info['qp_solver_iter'].sum() / max(qps, 1):6.1f} ADMM it/QP  "
          f"solved {np.mean(info['status'] == pa.SQP_SOLVED):.3f}  iters hist {np.bincount(info['iter'])}", flush=True)
ctx.close()
