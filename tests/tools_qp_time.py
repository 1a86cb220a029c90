"""Developer tool: time the LDS-resident QP kernel on random QPs (n=66, m=44: config B's size) with and without residual checks."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import polympc_amd as pa
from polympc_amd import workloads
B, n, m = 4096, int(os.environ.get("N", 66)), int(os.environ.get("M", 44))
q = workloads.random_qp_batch(B, n, m, seed=2)
dev = torch.device("cuda", 0)
stream = torch.cuda.Stream(dev); torch.cuda.set_stream(stream)
ctx = pa.Context(0, stream=stream.cuda_stream)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
d = {k: t(v) for k, v in q.items()}
x = torch.zeros(B, n, dtype=torch.float64, device=dev); y = torch.zeros(B, n + m, dtype=torch.float64, device=dev)
info = torch.zeros(B, 40, dtype=torch.uint8, device=dev)
for name, it, chk, ar in (("1 iteration (factor + 1 solve)", 1, 0, 0), ("51 iterations, no checks", 51, 0, 0), ("51 iterations, check every 10", 51, 10, 0), ("51 iterations, check every 1", 51, 1, 0)):
    s = pa.qp_settings_default(); s.max_iter = it; s.check_termination = chk; s.adaptive_rho = ar; s.eps_abs = 0.0; s.eps_rel = 0.0
    for rep in range(2):
        torch.cuda.synchronize(dev); t0 = time.perf_counter()
        ctx.qp_solve_batch_dev(B, n, m, d["H"], d["h"], d["A"], d["Alb"], d["Aub"], d["xlb"], d["xub"], x, y, info, s)
        torch.cuda.synchronize(dev); dt = time.perf_counter() - t0
    print(f"{name:40s} {dt*1e3:9.2f} ms for {B} QPs")
ctx.close()
