import os, sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import polympc_amd as pa
from polympc_amd import workloads
dev = torch.device("cuda", 0); stream = torch.cuda.Stream(dev); torch.cuda.set_stream(stream)
ctx = pa.Context(0, stream=stream.cuda_stream)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
for P, S, B in ((6, 1, 1), (6, 1, 2048), (5, 2, 1), (5, 3, 1)):
    wl = workloads.robot_batch(B, P=P, S=S); n, m = wl["n"], wl["m"]
    x = torch.zeros(B, n, dtype=torch.float64, device=dev); lam = torch.zeros(B, n + m, dtype=torch.float64, device=dev); info = torch.zeros(B, 48, dtype=torch.uint8, device=dev)
    ss = pa.sqp_settings_default(); ss.max_iter = 10; ss.line_search_max_iter = 10; ss.regularisation = 1; ss.exact_hessian_every_iter = 1
    qs = pa.qp_settings_sqp_default()
    dd, dl, du = t(wl["d"]), t(wl["lbx"]), t(wl["ubx"])
    step = lambda: ctx.sqp_solve_batch_dev(pa.MODEL_ROBOT, P, S, 0.0, 2.0, B, dd, dl, du, x, lam, info, ss, qs)
    step(); torch.cuda.synchronize(dev)
    t0 = time.perf_counter(); step(); torch.cuda.synchronize(dev); ms = (time.perf_counter() - t0) * 1e3
    inf = np.frombuffer(info.cpu().numpy().tobytes(), dtype=pa.capi.SQP_INFO_DTYPE)
    print(f"mirroring: robot P={P} S={S} n={n} B={B}: route {pa.capi.ROUTE_NAMES.get(ctx.last_route())} {ms:9.2f} ms, {int(inf['iter'].sum())} SQP iterations -> {ms / max(1, inf['iter'].max()):.2f} ms per iteration of the longest instance")
ctx.close()
