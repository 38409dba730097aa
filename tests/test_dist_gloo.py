"""world_size-2 gloo test (CPU) of the N>1 path: shard independence + the barrier / max-over-ranks / sum
aggregation bench.py uses (raven_amd/dist.py).  The data path itself has no collective."""
import os
import socket

import numpy as np
import pytest


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import torch.distributed as dist

    from oracle import oracle
    from raven_amd import dist as rdist
    from raven_amd import synth
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    gs, rs_seed = rdist.shard_seeds(rank)
    g = synth.make_genome(40_000, seed=gs)
    rs, _ = synth.make_reads(g, 8, 3000, seed=rs_seed)
    # the shard's result depends on nothing outside the shard: run the CPU oracle on it (stands in for the
    # device pass, which needs a GPU) and checksum
    r = oracle.Engine(15, 5).find_overlaps_and_create_piles(rs)
    dist.barrier()
    dt = 1.0 + rank  # pretend rank 1 is slower
    dt_max, bases = rdist.aggregate(dt, float(rs.total_bases), dist)
    q.put((rank, rs.total_bases, int(r["pile_data"].astype(np.uint64).sum()), len(r["overlaps"]), dt_max, bases))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_shards_and_aggregation():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, b0, c0, o0, t0, tot0), (r1, b1, c1, o1, t1, tot1) = res
    assert (r0, r1) == (0, 1)
    assert c0 != c1 and o0 > 0 and o1 > 0           # different, non-trivial shards
    assert t0 == t1 == 2.0                          # max over ranks
    assert tot0 == tot1 == float(b0 + b1)           # whole-job units
    from raven_amd import dist as rdist
    assert rdist.throughput(t0, tot0, 4) == (b0 + b1) * 4 / 2.0
    # single-process path is the identity
    assert rdist.aggregate(3.0, 7.0, None) == (3.0, 7.0)
    # shard 0 recomputed alone gives the same checksum (independence from the other rank)
    from oracle import oracle
    from raven_amd import synth
    gs, rs_seed = rdist.shard_seeds(0)
    rs, _ = synth.make_reads(synth.make_genome(40_000, seed=gs), 8, 3000, seed=rs_seed)
    r = oracle.Engine(15, 5).find_overlaps_and_create_piles(rs)
    assert int(r["pile_data"].astype(np.uint64).sum()) == c0 and len(r["overlaps"]) == o0
