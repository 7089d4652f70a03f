"""Rank layout for multi-GPU runs: one process per GPU, every rank owns a
contiguous block of the independent units (filter instances, frames).

The point-landmark update of one filter instance and the corner detection of one
frame share nothing with any other instance or frame, so the data path has no
exchange step and no collective; the only cross-rank operations are the barrier,
the max-over-ranks of the wall time that a benchmark or a driver wants, and one
gather-to-root of the results after the compute.  Two result messages exist:

  keypoints  per rank: counts int32[F] (F frames), then records int32[F][K_MAX][3]
             (the 12-byte vslam_feature records x, y, score bits, zero padded);
  filters    per rank: float64[B][n + n(n+1)/2] = posterior mean, then the upper
             triangle of the posterior factor packed by columns (LAPACK 'U' order).

torch.distributed's gather runs as grouped point-to-point transfers (under RCCL each
peer uses its own xGMI link to the root), which is what a gather of results wants:
nothing is reduced and nobody but the root needs the data.
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


K_MAX = 1024      # keypoint records per frame in the gathered message


def _gather(t):
    """dist.gather to rank 0 when a group of more than one rank exists; [t] otherwise; None off the root."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [t]
    if dist.get_rank() == 0:
        out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
        dist.gather(t, gather_list=out, dst=0)
        return out
    dist.gather(t, gather_list=None, dst=0)
    return None


def gather_keypoints_to_root(counts, records, k_max=K_MAX):
    """counts (F,) int32 and records (F, cap, 3) int32 of this rank's F frames (the per-frame output of the
    corner entry points: cap >= count records each) -> on rank 0 (counts_all (W*F,), records_all (W*F, k_max, 3))
    in rank order, i.e. frame r*F + i is frame i of rank r; None elsewhere.  Per frame at most k_max records
    travel (the score-sorted head); counts are clamped accordingly.  Two phases, as the message is specified:
    the counts first, then the padded records."""
    import torch
    F = counts.shape[0]
    c = counts.to(torch.int32).clamp(min=0, max=k_max).contiguous()
    pad = torch.zeros((F, k_max, 3), dtype=torch.int32, device=records.device)
    take = min(k_max, records.shape[1])
    pad[:, :take] = records[:, :take]
    mask = torch.arange(k_max, device=records.device)[None, :] >= c[:, None].to(torch.int64)
    pad[mask] = 0
    got_c = _gather(c)
    got_r = _gather(pad)
    if got_c is None:
        return None
    return torch.cat(got_c), torch.cat(got_r)


def pack_filter_results(mu_post, S_post):
    """mu_post (B, n) and S_post (B, n, n) holding each upper-triangular factor column-major (S_post[b, c, r] =
    S(r, c)) -> (B, n + n(n+1)/2): the mean followed by the upper triangle packed by columns."""
    import torch
    B, n = mu_post.shape
    c, r = torch.tril_indices(n, n, device=S_post.device)            # pairs c >= r, ordered by c then r: column by column
    return torch.cat([mu_post, S_post[:, c, r]], dim=1).contiguous()


def unpack_filter_results(msg, n):
    """Inverse of pack_filter_results: (mu (B, n), S (B, n, n) column-major per instance, zero below the diagonal)."""
    import torch
    B = msg.shape[0]
    c, r = torch.tril_indices(n, n, device=msg.device)
    S = torch.zeros((B, n, n), dtype=msg.dtype, device=msg.device)
    S[:, c, r] = msg[:, n:]
    return msg[:, :n].contiguous(), S


def gather_filter_results_to_root(mu_post, S_post):
    """Posterior means and packed factors of this rank's B filters -> on rank 0 one (W*B, n + n(n+1)/2) tensor
    in rank order (filter r*B + i is filter i of rank r); None elsewhere."""
    import torch
    got = _gather(pack_filter_results(mu_post, S_post))
    return None if got is None else torch.cat(got)
