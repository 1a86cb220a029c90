"""Developer probe (GPU): the reference's nonlinear_constraints_test.cpp problem (NP = 1, NG = 1; exact Hessians + Gershgorin) as a batch of perturbed instances —
the condensed register kernel with path-constraint rows (default since round 6) against the full two-rows-per-lane inverse (PMPC_NO_CONDREG=1)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import polympc_amd as pa
from test_oracle_pins import _parking_batch
B = int(os.environ.get("B", 4096)); UBG = float(os.environ.get("UBG", 1.2))
lbx, ubx, xg, d = _parking_batch(B)
nn = 11; n, m = 56, 44
dev = torch.device("cuda", 0); stream = torch.cuda.Stream(dev); torch.cuda.set_stream(stream)
ctx = pa.Context(0, stream=stream.cuda_stream)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
dd, dl, du, dg = t(d), t(lbx), t(ubx), t(xg)
lg, ug = t(np.full((B, nn), -10.0)), t(np.full((B, nn), UBG))
x = torch.zeros(B, n, dtype=torch.float64, device=dev); lam = torch.zeros(B, n + m, dtype=torch.float64, device=dev); info = torch.zeros(B, 48, dtype=torch.uint8, device=dev)
ss = pa.sqp_settings_default(); ss.max_iter = 20; ss.line_search_max_iter = 10; ss.regularisation = 2; ss.exact_hessian_every_iter = 1
qs = pa.qp_settings_sqp_default()
step = lambda: ctx.sqp_solve_batch_dev(pa.MODEL_PARKING_NG, 5, 2, 0.0, 1.0, B, dd, dl, du, x, lam, info, ss, qs, x_guess=dg, lbg=lg, ubg=ug)
step(); torch.cuda.synchronize(dev)
t0 = time.perf_counter(); R = 5
for _ in range(R): step()
torch.cuda.synchronize(dev)
ms = (time.perf_counter() - t0) / R * 1e3
inf = np.frombuffer(info.cpu().numpy().tobytes(), dtype=pa.capi.SQP_INFO_DTYPE)
print(f"parking NP=1 NG=1 ubg={UBG} B={B}: route {pa.capi.ROUTE_NAMES.get(ctx.last_route())}  {ms:8.2f} ms/step  QPs {int(inf['iter'].sum())}  {inf['qp_solver_iter'].sum() / inf['iter'].sum():.1f} ADMM it/QP  solved {np.mean(inf['status'] == pa.SQP_SOLVED):.3f}")
ctx.close()
