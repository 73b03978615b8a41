"""N>1 control plane on CPU: world_size 2, gloo. The data path has no collective; what must hold is
that units are partitioned exactly once across ranks and that the timing reduce is a max."""
import os
import socket
import sys

import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, os.path.join(ROOT, "city-rollup_amd"))
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from cityprover import dist as D
    dist = D.init("gloo")
    units = D.shard_units(13, rank, world)
    D.barrier(dist)
    t = D.max_over_ranks(dist, 1.0 + rank)  # rank 1 is "slower"
    total = D.sum_over_ranks(dist, len(units))
    q.put((rank, units, t, total, [D.unit_seed(7, u) for u in units]))
    D.barrier(dist)
    dist.destroy_process_group()


def test_two_rank_sharding_and_timing_reduce():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    units = sorted(u for r in res for u in r[1])
    assert units == list(range(13))               # every unit exactly once
    assert all(r[2] == 2.0 for r in res)           # max over ranks
    assert all(r[3] == 13.0 for r in res)
    seeds = [s for r in res for s in r[4]]
    assert len(set(seeds)) == 13                   # seeds depend on the unit, not on the rank


def test_single_process_is_a_noop():
    sys.path.insert(0, os.path.join(ROOT, "city-rollup_amd"))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        os.environ.pop(k, None)
    from cityprover import dist as D
    assert D.init() is None
    assert D.shard_units(5, 0, 1) == [0, 1, 2, 3, 4]
    assert D.max_over_ranks(None, 3.5) == 3.5
