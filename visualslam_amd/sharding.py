"""Rank layout for multi-GPU runs: one process per GPU, every rank owns a
contiguous block of the independent units (filter instances, frames).

The point-landmark update of one filter instance and the corner detection of one
frame share nothing with any other instance or frame, so the data path has no
exchange step and no collective; the only cross-rank operations are the barrier
and the max-over-ranks of the wall time that a benchmark or a driver wants.
"""
import os
from dataclasses import dataclass


@dataclass(frozen=True)
class RankLayout:
    rank: int
    local_rank: int
    world: int

    @staticmethod
    def from_env(env=None):
        env = os.environ if env is None else env
        return RankLayout(int(env.get("RANK", "0")), int(env.get("LOCAL_RANK", "0")), int(env.get("WORLD_SIZE", "1")))


def shard_range(total, rank, world):
    """[start, stop) of `rank`'s block when `total` units are dealt to `world` ranks
    in contiguous blocks whose sizes differ by at most one (the first total % world
    ranks hold the larger blocks).  Blocks are disjoint and cover range(total)."""
    if world < 1 or not 0 <= rank < world or total < 0:
        raise ValueError("shard_range: need 0 <= rank < world and total >= 0 (got total=%d rank=%d world=%d)" % (total, rank, world))
    base, extra = divmod(total, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def instance_seed(base_seed, rank, index, per_rank=4096):
    """Seed of the index-th synthetic instance of `rank`; distinct over ranks for index < per_rank."""
    if not 0 <= index < per_rank:
        raise ValueError("instance_seed: index %d outside [0, %d)" % (index, per_rank))
    return base_seed + per_rank * rank + index


def max_over_ranks(seconds, device=None):
    """Largest `seconds` over all ranks of the default process group (the value itself
    when no group is initialised).  `device` is where the one-element tensor lives:
    the rank's GPU under RCCL, None (CPU) under gloo."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(seconds)
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_to_root(tensor, device=None):
    """Results of all ranks on rank 0 (list indexed by rank; None on the other ranks): the one exchange
    a sharded batch needs, after the compute.  Every rank passes a tensor of the same shape and dtype."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [tensor]
    t = tensor if device is None else tensor.to(device)
    if dist.get_rank() == 0:
        out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
        dist.gather(t, gather_list=out, dst=0)
        return out
    dist.gather(t, gather_list=None, dst=0)
    return None
