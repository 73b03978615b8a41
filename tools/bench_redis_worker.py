#!/usr/bin/env python3
"""The harness as a worker of a live deployment (--mode redis-worker) against the in-process Redis of tests/fake_redis.py: the
reference's example block drained through RSMQ + the Redis proof store, one job per round (the reference's loop) and in rounds of
up to N messages. Prints one JSON line per configuration: ms per block (mean of `reps` blocks, each loaded afresh)."""
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for d in ("city-rollup_amd", "tests", "tools"):
    sys.path.insert(0, os.path.join(ROOT, d))
from fake_redis import FakeRedis  # noqa: E402
import test_qbench_redis as T  # noqa: E402
import make_circuit_pack  # noqa: E402


def main():
    golden = os.path.join(ROOT, "tests", "golden")
    T.build_harness()
    with tempfile.TemporaryDirectory() as tmp:
        pack = make_circuit_pack.make_pack(os.path.join(tmp, "pack"), db=12)
        for batch in (1, 4, 16, 64):
            times, res = [], None
            for _ in range(3):
                with FakeRedis() as r:
                    T.load_block(r, golden)
                    t0 = time.perf_counter()
                    p = subprocess.run([T.EXE, "--mode", "redis-worker", "--redis", r.uri, "--pack", pack, "--drain", "--redis-batch", str(batch)],
                                       capture_output=True, text=True)
                    assert p.returncode == 0, p.stderr
                    res = json.loads(p.stdout.strip().splitlines()[-1])
                    times.append(res["wall_s"])
                    assert res["jobs"] == 60 and res["queue_left"] == 0
            print(json.dumps({"redis_batch": batch, "ms_per_block": [round(1e3 * t, 1) for t in times], "rounds": res["rounds"], "launches": res["launches"],
                              "note": "wall time of the worker's loop for one example block (46 jobs = 64 proofs), circuits resident, the fake "
                                      "server in the same process as this script"}), flush=True)


if __name__ == "__main__":
    main()
