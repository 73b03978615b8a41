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


def _bench(args, env_extra, timeout=300):
    import subprocess
    env = dict(os.environ, CITYPROVER_BENCH_STUB="1", **env_extra)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=timeout)


def test_bench_gpus_branch_spawns_its_ranks_and_only_rank_0_prints():
    """`python bench.py --gpus N` without a launcher: N ranks of itself, one rendezvous on 127.0.0.1, the control
    plane's barrier / max / sum / broadcast, ONE JSON line from rank 0 — with a stub in place of the GPU section
    (CITYPROVER_BENCH_STUB), so that it runs here."""
    import json
    for n in (2, 8):
        r = _bench(["--gpus", str(n), "--steps", "3", "--warmup", "1"], {})
        assert r.returncode == 0, r.stderr[-2000:]
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1, r.stdout
        # the contract is ONE line on stdout: gloo's "[Gloo] Rank 0 is connected ..." chatter must go to stderr
        assert r.stdout.strip().splitlines() == lines, r.stdout
        out = json.loads(lines[0])
        assert out["n_gpus"] == n and out["units"] == float(n) and out["broadcast"] == "pack-of-rank-0" and out["local_rank"] == 0
        assert out["elapsed_s"] >= 0.01 * n            # the slowest rank's time, not rank 0's


def test_bench_gpus_branch_propagates_failures_without_hanging():
    # a rank whose side measurement raises: EVERY rank leaves (nobody waits in the next collective), the launcher exits non-zero
    r = _bench(["--gpus", "4"], {"CITYPROVER_BENCH_STUB_FAIL": "2"}, timeout=120)
    assert r.returncode != 0 and "failed on 1 rank(s)" in r.stderr and "rank exit codes" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
    # a rank that exits non-zero at the very end: the line is there, the exit code says so all the same
    r = _bench(["--gpus", "3"], {"CITYPROVER_BENCH_STUB_EXIT": "1"}, timeout=120)
    assert r.returncode != 0 and "rank exit codes" in r.stderr and "3" in r.stderr
    # the flag must agree with a launcher's world size
    env = {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"}
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], env=dict(os.environ, CITYPROVER_BENCH_STUB="1", **env),
                       capture_output=True, text=True, timeout=60)
    assert r.returncode != 0 and "WORLD_SIZE" in r.stderr
