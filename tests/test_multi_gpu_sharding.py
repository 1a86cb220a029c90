"""N > 1 path on CPU: world_size-2 gloo. Checks the batch sharding used by bench.py (disjoint, covering, identical
problem data to the single-process stream), the statistics reduction, and shard-invariance of the results (solving the
two shards separately == solving the global batch), with the oracle standing in for the GPU solve on this CPU-only box."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

B = 6


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from polympc_amd import workloads, sharding
    from oracle import binding as ob
    wl = workloads.robot_batch(B, first=sharding.shard_first_instance(rank, B))
    ss = ob.sqp_default_settings(); ss.max_iter = 10; ss.line_search_max_iter = 10
    x, lam, info = ob.sqp_solve_batch(ob.MODEL_ROBOT, 6, 1, 0.0, 2.0, B, wl["d"], wl["lbx"], wl["ubx"], sqp_settings=ss, pivot=ob.PIVOT_STATIC)
    qp = sum(i.iter for i in info)
    dist.barrier()
    (qp_all, n_all), t_max = sharding.combine_stats(dist, torch.device("cpu"), [qp, B], elapsed=1.0 + rank)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), x=x, lbx=wl["lbx"], qp=qp, qp_all=qp_all, n_all=n_all, t_max=t_max)
    dist.destroy_process_group()


def test_two_rank_sharding_gloo(tmp_path, oracle):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r = [np.load(tmp_path / f"rank{k}.npz") for k in range(world)]
    from polympc_amd import workloads
    g = workloads.robot_batch(world * B)
    # shards are disjoint, cover the stream in order, and carry exactly the single-process problem data
    assert np.array_equal(np.concatenate([r[0]["lbx"], r[1]["lbx"]]), g["lbx"])
    # statistics: SUM over ranks of counts, MAX over ranks of time, identical on every rank
    for k in range(world):
        assert r[k]["qp_all"] == r[0]["qp"] + r[1]["qp"] and r[k]["n_all"] == world * B and r[k]["t_max"] == 2.0
    # shard-invariance: no cross-instance coupling anywhere on the path
    ss = oracle.sqp_default_settings(); ss.max_iter = 10; ss.line_search_max_iter = 10
    xg, _, _ = oracle.sqp_solve_batch(oracle.MODEL_ROBOT, 6, 1, 0.0, 2.0, world * B, g["d"], g["lbx"], g["ubx"], sqp_settings=ss, pivot=oracle.PIVOT_STATIC)
    assert np.array_equal(np.concatenate([r[0]["x"], r[1]["x"]]), xg)


# ------------------------------------------------------------------------------------------------ the PRODUCT on N > 1 ranks (GPU)
def _gpu_worker(rank, world, port, out_dir):
    """One rank of the multi-GPU path as bench.py runs it — a context per rank, the rank's shard of the instance stream, device-resident
    solve, statistics combined over the process group — with both ranks on device 0 (the test box has one GPU) and gloo for the group."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import polympc_amd as pa
    from polympc_amd import workloads, sharding
    Bg = 64
    dev = torch.device("cuda", 0)
    wl = workloads.robot_batch(Bg, first=sharding.shard_first_instance(rank, Bg))
    stream = torch.cuda.Stream(dev)
    ctx = pa.Context(0, stream=stream.cuda_stream)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    n, m = wl["n"], wl["m"]
    x = torch.zeros(Bg, n, dtype=torch.float64, device=dev); lam = torch.zeros(Bg, m + n, dtype=torch.float64, device=dev)
    info = torch.zeros(Bg, 48, dtype=torch.uint8, device=dev)
    ss = pa.sqp_settings_default(); ss.max_iter = 10; ss.line_search_max_iter = 10
    ctx.sqp_solve_batch_dev(wl["model"], 6, 1, 0.0, 2.0, Bg, t(wl["d"]), t(wl["lbx"]), t(wl["ubx"]), x, lam, info, ss, pa.qp_settings_sqp_default())
    torch.cuda.synchronize(dev)
    inf = np.frombuffer(info.cpu().numpy().tobytes(), dtype=pa.capi.SQP_INFO_DTYPE)
    dist.barrier()
    (qp_all,), t_max = sharding.combine_stats(dist, torch.device("cpu"), [int(inf["iter"].sum())], elapsed=1.0 + rank)
    np.savez(os.path.join(out_dir, f"g{rank}.npz"), x=x.cpu().numpy(), it=inf["iter"], qp_all=qp_all, t_max=t_max)
    ctx.close()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_two_rank_product_shards_equal_single_process_batch(tmp_path):
    """Two ranks of the HIP library, each solving its shard, against ONE process solving the global batch: bit-identical solutions and
    iteration counts (no cross-instance coupling; the shard boundaries are those of bench.py), and the reductions bench.py reports."""
    import polympc_amd as pa
    from polympc_amd import workloads
    world, Bg = 2, 64
    mp.spawn(_gpu_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r = [np.load(tmp_path / f"g{k}.npz") for k in range(world)]
    g = workloads.robot_batch(world * Bg)
    ctx = pa.Context(0)
    ss = pa.sqp_settings_default(); ss.max_iter = 10; ss.line_search_max_iter = 10
    xg, _, ig = ctx.sqp_solve_batch(g["model"], 6, 1, 0.0, 2.0, world * Bg, g["d"], g["lbx"], g["ubx"], sqp_settings=ss)
    ctx.close()
    assert np.array_equal(np.concatenate([r[0]["x"], r[1]["x"]]), xg)
    assert np.array_equal(np.concatenate([r[0]["it"], r[1]["it"]]), ig["iter"])
    for k in range(world):
        assert r[k]["qp_all"] == ig["iter"].sum() and r[k]["t_max"] == 2.0


@pytest.mark.gpu
def test_bench_gpus_flag_starts_the_ranks(tmp_path):
    """`python bench.py --gpus 2` with no launcher must start two ranks itself and report n_gpus = 2 with the whole-job aggregate
    (developer switches put both ranks on the one GPU of the test box and use gloo for the barrier / reductions)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PMPC_BENCH_SINGLE_DEVICE="1", PMPC_BENCH_BACKEND="gloo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "512",
                          "--cpu-sample", "0", "--configs", "", "--no-replay", "--min-seconds", "0.2", "--detail-out", str(tmp_path / "detail.json")],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    allout = [l for l in out.stdout.splitlines() if l.strip()]
    lines = [l for l in allout if l.startswith("{")]
    # stdout carries ONE JSON line, the compact contract line, and it is the LAST line (the detail goes to the file and to stderr; the gloo
    # backend of this developer configuration prints its own "[Gloo] Rank ..." connection notes in front of it)
    assert len(lines) == 1 and len(lines[0]) < 4096 and allout[-1] == lines[0], allout
    j = json.loads(lines[0])
    assert j["timed_blocks"]["blocks"] >= 2                     # the timed block was repeated; both ranks agreed on the count (the barriers paired up)
    assert json.load(open(tmp_path / "detail.json"))["n_gpus"] == 2
    assert j["n_gpus"] == 2 and j["config"]["global_batch"] == 1024 and j["value"] > 0 and j["scaling"] == "weak"
    # a launcher / flag mismatch is an error, not a silent single-GPU run
    env2 = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    bad = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env2, capture_output=True, text=True, timeout=300)
    assert bad.returncode != 0 and "WORLD_SIZE" in (bad.stderr + bad.stdout)
