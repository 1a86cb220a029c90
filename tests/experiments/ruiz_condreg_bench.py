"""Developer probe (GPU): the policy set of valet_parking_mpc_test.cpp (Ruiz preconditioner + filter line search + block BFGS, QP max_iter 1000) on batches of robot OCPs of the
reference's grids — the condensed hook kernel (default since late round 6) against the full two-rows-per-lane inverse (PMPC_NO_CONDREG_RUIZ=1, or an older library via PMPC_LIB)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import polympc_amd as pa
from polympc_amd import workloads
dev = torch.device("cuda", 0); stream = torch.cuda.Stream(dev); torch.cuda.set_stream(stream)
ctx = pa.Context(0, stream=stream.cuda_stream)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
for P, S, B in ((5, 2, 4096), (5, 2, 1), (5, 3, 2048)):
    for pol in (dict(preconditioner=1, line_search=1, hessian_update=1), dict(preconditioner=1), dict(line_search=1), dict(regularisation=1, exact_hessian_every_iter=1)):
        wl = workloads.robot_batch(B, P=P, S=S)
        n, m = wl["n"], wl["m"]
        x = torch.zeros(B, n, dtype=torch.float64, device=dev); lam = torch.zeros(B, n + m, dtype=torch.float64, device=dev); info = torch.zeros(B, 48, dtype=torch.uint8, device=dev)
        ss = pa.sqp_settings_default(); ss.max_iter = 10; ss.line_search_max_iter = 10
        for k, v in pol.items(): setattr(ss, k, v)
        qs = pa.qp_settings_sqp_default(); qs.max_iter = 1000
        dd, dl, du = t(wl["d"]), t(wl["lbx"]), t(wl["ubx"])
        step = lambda: ctx.sqp_solve_batch_dev(pa.MODEL_ROBOT, P, S, 0.0, 2.0, B, dd, dl, du, x, lam, info, ss, qs)
        step(); torch.cuda.synchronize(dev)
        t0 = time.perf_counter(); R = 5
        for _ in range(R): step()
        torch.cuda.synchronize(dev)
        ms = (time.perf_counter() - t0) / R * 1e3
        inf = np.frombuffer(info.cpu().numpy().tobytes(), dtype=pa.capi.SQP_INFO_DTYPE)
        print(f"robot P={P} S={S} B={B} {pol}: route {pa.capi.ROUTE_NAMES.get(ctx.last_route())}  {ms:8.3f} ms/step  QPs {int(inf['iter'].sum())}  {inf['qp_solver_iter'].sum() / max(1, inf['iter'].sum()):.1f} ADMM it/QP  solved {np.mean(inf['status'] == pa.SQP_SOLVED):.3f}")
ctx.close()
