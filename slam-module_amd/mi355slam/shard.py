"""Sharding of independent units (frames / descriptor pairs / BA windows) over ranks, and the only exchange the path has:
agreeing on the job's wall time.  One process per GPU; no data-path collective.  Works over RCCL
(backend "nccl" on ROCm) on the GPUs and over gloo on CPUs (tests)."""


def shard_range(n_units, rank, world):
    """Contiguous block partition of n_units over `world` ranks (remainder to the low ranks)."""
    base, rem = divmod(n_units, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def sequence_of(unit, world):
    """Round-robin assignment of independent sequences to GPUs (C5: sequence s -> GPU s mod G)."""
    return unit % world


def aggregate(dist, torch, local_units, local_seconds, device=None):
    """Whole-job totals: units summed over ranks, time = MAX over ranks (the contract of bench.py)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(local_units), float(local_seconds)
    t = torch.tensor([float(local_seconds)], dtype=torch.float64, device=device)
    u = torch.tensor([float(local_units)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(u, op=dist.ReduceOp.SUM)
    return float(u.item()), float(t.item())
