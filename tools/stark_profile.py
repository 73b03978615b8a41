#!/usr/bin/env python3
"""Where the time of one cp_stark_prove goes at the SHA-256 STARK's shape (418 + 912 columns): per-kernel HIP-event milliseconds and
launch counts of one call, beside its wall time. usage: stark_profile.py [log_rows ...]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "city-rollup_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np  # noqa: E402
import cityprover as cp  # noqa: E402
import air_programs as A  # noqa: E402
import bench_stark_air as B  # noqa: E402
from bench_stark_fri import arity_for  # noqa: E402


def run(prover, log_rows):
    rb, ch, q, na = 1, 4, 1, 2
    n = 1 << log_rows
    cons_b, map_b = B.programs()
    cons, mp = cons_b.gpu(prover), map_b.gpu(prover)
    rng = np.random.default_rng(1)
    pub = rng.integers(0, cp.P, 4, dtype=np.uint64)
    trace = rng.integers(0, cp.P, size=(B.K0, n), dtype=np.uint64)
    desc, keep = cp.stark_desc(log_rows, q, na, cp.fri_params(log_rows, rb, ch, 16, 84, arity_for(log_rows, rb, ch)), B.K0, cons, B.K1, 6, n_public=4,
                               steps=[("map", mp), ("cubic_inverse", 0, B.K1 // 3, A.CUBIC_MODULUS), ("prefix_sum", 0, B.K1, False)])
    for _ in range(2):
        cp.stark_prove(prover, desc, trace, cp.ChallengerState(), publics=pub)
    prover.profile_begin()
    t0 = time.perf_counter()
    cp.stark_prove(prover, desc, trace, cp.ChallengerState(), publics=pub)
    wall = (time.perf_counter() - t0) * 1e3
    prof = prover.profile_end()
    cons.close()
    mp.close()
    kern = {k: (round(v["total_ms"], 4), v["launches"]) for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["total_ms"]) if not k.startswith(("host:", "wait:"))}
    return {"log_rows": log_rows, "wall_ms": wall, "kernels_ms_sum": sum(v[0] for v in kern.values()), "launches": sum(v[1] for v in kern.values()),
            "kernels (ms, launches)": kern}


if __name__ == "__main__":
    p = cp.Prover(0)
    print(json.dumps([run(p, k) for k in ([int(a) for a in sys.argv[1:]] or [10, 14])], indent=1))
    p.close()
