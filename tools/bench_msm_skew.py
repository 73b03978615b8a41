#!/usr/bin/env python3
"""G1 MSM with witness-like scalars: 70 % of them in {0, 1} (gnark witnesses are mostly bits), the rest uniform — the
ones all land in one bucket of window 0, which exercises the chunked heavy-bucket path (msm.h: k_heavy_*)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "city-rollup_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import cityprover as cp
from bench_msm import G, R
p = cp.Prover(0)
out = []
for log_n in (20, 22):
    n = 1 << log_n
    rng = np.random.default_rng(log_n)
    k = rng.integers(0, 2**64, (n, 4), dtype=np.uint64)
    k[:, 3] >>= np.uint64(1)
    small = rng.random(n) < 0.7
    k[small, 1:] = 0
    k[small, 0] = rng.choice(np.array([0, 1], dtype=np.uint64), int(small.sum()))
    pts = cp.G1Points.synthetic(p, G, 7, 3, n)
    ds = p.to_device(k)
    pts.msm_dev(ds.ptr)
    p.profile_begin()
    t0 = time.perf_counter()
    for _ in range(3):
        pts.msm_dev(ds.ptr)
    dt = (time.perf_counter() - t0) / 3
    prof = p.profile_end()
    out.append({"log_n": log_n, "ms": dt * 1e3, "kernels_ms": {kk: round(v["total_ms"] / 3, 3) for kk, v in prof.items() if kk.startswith("msm") and v["total_ms"] > 0.6}})
    ds.free(); pts.free()
print(json.dumps(out))
