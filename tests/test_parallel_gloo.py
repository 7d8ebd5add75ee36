"""CPU, world_size 2 over gloo: the N>1 plumbing of bench.py / the agent (sharding, per-rank seeds, max-over-ranks timing,
scalar sums, flat gradient bucket all-reduce)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from imagecompressionlearnedliftingandlearnedtreebasedmodels_amd import parallel


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    r, w = parallel.init(backend="gloo")
    assert (r, w) == (rank, world)
    x = torch.arange(7 * 3, dtype=torch.float32).reshape(7, 3)       # ragged: 7 images over 2 ranks
    mine = parallel.shard_batch(x, rank, world)
    n = torch.tensor([float(mine.shape[0])])
    parallel.sum_over_ranks(n)
    tot = torch.tensor([float(mine.sum())], dtype=torch.float64)
    parallel.sum_over_ranks(tot)
    tmax = parallel.max_over_ranks(1.0 + rank)
    # flat gradient bucket: each rank contributes rank+1; mean over ranks = 1.5
    lin = torch.nn.Linear(3, 2)
    bucket = parallel.FlatGradBucket(lin.parameters())
    bucket.zero_()
    for p in lin.parameters():
        p.grad += float(rank + 1)
    bucket.all_reduce_mean()
    g = torch.Generator().manual_seed(parallel.rank_seed(1337, rank))
    noise0 = float(torch.rand(1, generator=g))
    # a rank-local failure (here: rank 1 "fails") is seen by every rank, and nobody is left waiting in a collective
    ok_all = parallel.all_ranks_ok(rank != 1)
    ok_all2 = parallel.all_ranks_ok(True)
    assert ok_all is False and ok_all2 is True
    parallel.barrier()
    q.put((rank, mine.shape[0], float(n), float(tot), tmax, float(lin.weight.grad.mean()), float(lin.bias.grad.mean()), noise0))
    dist.destroy_process_group()


def test_world_size_2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, n0, tot_n0, tot0, tmax0, gw0, gb0, z0), (r1, n1, tot_n1, tot1, tmax1, gw1, gb1, z1) = res
    assert (n0, n1) == (4, 3) and tot_n0 == tot_n1 == 7.0                 # every image exactly once
    assert tot0 == tot1 == float(torch.arange(21.0).sum())
    assert tmax0 == tmax1 == 2.0                                          # max over ranks
    assert abs(gw0 - 1.5) < 1e-6 and abs(gw1 - 1.5) < 1e-6 and abs(gb0 - 1.5) < 1e-6
    assert z0 != z1                                                       # per-rank noise streams differ


def test_shard_range_properties():
    for n in (0, 1, 7, 8, 33):
        for w in (1, 2, 3, 8):
            spans = [parallel.shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1
