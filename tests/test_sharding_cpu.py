"""Rank layout of multi-GPU runs, rehearsed with two gloo ranks on the CPU."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from visualslam_amd import sharding


def test_shard_range_partitions():
    for total in (0, 1, 7, 64, 512, 1000):
        for world in (1, 2, 3, 8):
            blocks = [sharding.shard_range(total, r, world) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == total
            for (a0, a1), (b0, b1) in zip(blocks, blocks[1:]):
                assert a1 == b0 and a0 <= a1
            sizes = [b - a for a, b in blocks]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        sharding.shard_range(10, 2, 2)
    with pytest.raises(ValueError):
        sharding.shard_range(-1, 0, 1)


def test_layout_from_env_and_seeds():
    lay = sharding.RankLayout.from_env({"RANK": "3", "LOCAL_RANK": "1", "WORLD_SIZE": "8"})
    assert (lay.rank, lay.local_rank, lay.world) == (3, 1, 8)
    assert sharding.RankLayout.from_env({}) == sharding.RankLayout(0, 0, 1)
    seeds = {sharding.instance_seed(9000, r, i) for r in range(8) for i in range(32)}
    assert len(seeds) == 8 * 32
    assert sharding.max_over_ranks(1.25) == 1.25      # no process group: identity


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, total, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank),
                      WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lay = sharding.RankLayout.from_env()
        a, b = sharding.shard_range(total, lay.rank, lay.world)
        # Every rank evaluates its own block of an embarrassingly parallel job (here: squares);
        # the gather below is the TEST's way of looking at all blocks, not part of the data path.
        mine = torch.arange(a, b, dtype=torch.float64) ** 2
        pad = torch.full(((total + world - 1) // world,), -1.0, dtype=torch.float64)
        pad[: b - a] = mine
        got = [torch.empty_like(pad) for _ in range(world)]
        dist.all_gather(got, pad)
        slowest = sharding.max_over_ranks(0.5 + rank)
        rooted = sharding.gather_to_root(torch.full((3,), float(rank), dtype=torch.float64))
        assert (rooted is None) == (rank != 0)
        if rank == 0:
            assert [r.tolist() for r in rooted] == [[float(q)] * 3 for q in range(world)]
        # the two result messages: keypoint lists (counts + padded records) and filter posteriors (mean + packed
        # factor), checked on the root for content and for the order in which the ranks' blocks are reassembled
        F, cap, kmax = 3, 40, 16
        cnt = torch.tensor([5 + rank, 0, 40], dtype=torch.int32)
        rec = (torch.arange(F * cap * 3, dtype=torch.int32).reshape(F, cap, 3) + 100000 * (rank + 1))
        kp = sharding.gather_keypoints_to_root(cnt, rec, k_max=kmax)
        B, n = 2, 5
        mu = torch.arange(B * n, dtype=torch.float64).reshape(B, n) + 1000.0 * rank
        Su = torch.triu(torch.arange(1, n * n + 1, dtype=torch.float64).reshape(n, n)) + 0.5 * rank     # upper triangular S
        Sp = torch.stack([Su.T.contiguous() + b for b in range(B)])                                      # column-major per instance
        Sp = Sp * (torch.tril(torch.ones(n, n)))[None]                                                   # keep the stored triangle only
        fl = sharding.gather_filter_results_to_root(mu, Sp)
        assert (kp is None) == (rank != 0) and (fl is None) == (rank != 0)
        if rank == 0:
            call, rall = kp
            assert call.tolist() == [5, 0, 16, 6, 0, 16] and rall.shape == (world * F, kmax, 3)
            for q in range(world):
                want = torch.arange(F * cap * 3, dtype=torch.int32).reshape(F, cap, 3) + 100000 * (q + 1)
                for i in range(F):
                    k = int(call[q * F + i])
                    assert torch.equal(rall[q * F + i, :k], want[i, :k]) and not rall[q * F + i, k:].any()
            assert fl.shape == (world * B, n + n * (n + 1) // 2)
            mu_all, S_all = sharding.unpack_filter_results(fl, n)
            for q in range(world):
                for b in range(B):
                    assert torch.equal(mu_all[q * B + b], torch.arange(b * n, (b + 1) * n, dtype=torch.float64) + 1000.0 * q)
                    Sq = (torch.triu(torch.arange(1, n * n + 1, dtype=torch.float64).reshape(n, n)) + 0.5 * q).T + b
                    assert torch.equal(S_all[q * B + b], Sq * torch.tril(torch.ones(n, n)))
            # packed by columns: first entries are S(0,0), S(0,1), S(1,1), S(0,2) ...
            assert fl[0, n:n + 4].tolist() == [Su[0, 0].item(), Su[0, 1].item(), Su[1, 1].item(), Su[0, 2].item()]
        dist.barrier()
        if rank == 0:
            flat = torch.cat([g[g >= 0] for g in got])
            out.put((flat.tolist(), slowest))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("total", [7, 64])
def test_two_ranks_cover_the_job(total):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, out)) for r in range(world)]
    for p in procs:
        p.start()
    flat, slowest = out.get(timeout=120)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert flat == [float(i * i) for i in range(total)]
    assert slowest == 1.5
