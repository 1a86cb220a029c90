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
