"""world_size-2 gloo test of the multi-GPU path's logic: clips shard by rank with no data-path collective, per-clip
RNG makes results independent of the number of ranks, timing is reduced with MAX over ranks."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _clip_result(seed, clip):
    from ctrlv_amd.distributed import clip_generator
    g = clip_generator(seed, clip)
    return torch.randn(4, generator=g).sum().item()      # stands in for one clip's sampling loop


def _worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from ctrlv_amd import distributed as D
    r, w, _ = D.init("gloo")
    assert (r, w) == (rank, world) and dist.get_backend() == "gloo"
    mine = D.shard_clips(7, r, w)
    res = {c: _clip_result(1234, c) for c in mine}
    D.barrier()
    t = D.max_over_ranks(1.0 + rank)                      # slowest rank defines the elapsed time
    n = D.sum_over_ranks(len(mine))
    out.put((rank, mine, res, t, n))
    dist.destroy_process_group()


def test_clip_sharding_over_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    got.sort()
    assert got[0][1] == [0, 2, 4, 6] and got[1][1] == [1, 3, 5]
    assert all(g[3] == 2.0 for g in got) and all(g[4] == 7 for g in got)
    merged = {**got[0][2], **got[1][2]}
    single = {c: _clip_result(1234, c) for c in range(7)}     # what a 1-rank run computes
    assert merged == single


def test_shard_helpers_single_process():
    from ctrlv_amd import distributed as D
    assert D.shard_clips(8, 3, 8) == [3] and D.shard_clips(3, 5, 8) == []
    assert sorted(sum((D.shard_clips(10, r, 4) for r in range(4)), [])) == list(range(10))
    assert D.max_over_ranks(3.5) == 3.5 and D.sum_over_ranks(2) == 2.0
    import pytest
    with pytest.raises(ValueError):
        D.shard_clips(4, 4, 4)
