"""Batch sharding across GPUs: every OCP instance is independent, so rank r of W simply owns the
contiguous range [r*B, (r+1)*B) of the instance stream and there is NO data-path collective. torch.distributed is used
only for the barrier and to combine per-rank statistics (SUM of counts, MAX of elapsed time)."""


def shard_first_instance(rank, batch_per_rank):
    """Index of the first instance owned by `rank` (weak scaling: fixed batch per rank)."""
    return int(rank) * int(batch_per_rank)


def combine_stats(dist, device, counts, elapsed):
    """counts: list of floats summed over ranks; elapsed: float, max over ranks. Works with nccl (RCCL) and gloo."""
    import torch
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return [float(c) for c in counts], float(elapsed)
    s = torch.tensor([float(c) for c in counts], dtype=torch.float64, device=device)
    m = torch.tensor([float(elapsed)], dtype=torch.float64, device=device)
    dist.all_reduce(s, op=dist.ReduceOp.SUM)
    dist.all_reduce(m, op=dist.ReduceOp.MAX)
    return [float(v) for v in s.tolist()], float(m.item())


def max_over_ranks(dist, device, values):
    """Element-wise MAX over ranks of a list of floats (per timed block: the slowest rank's wall time)."""
    import torch
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return [float(v) for v in values]
    t = torch.tensor([float(v) for v in values], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return [float(v) for v in t.tolist()]
