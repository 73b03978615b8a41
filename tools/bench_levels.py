#!/usr/bin/env python3
"""Merkle tree levels of a product-shaped batch (n_trees commitments of k polynomials, n = 2^12, rate 8, cap height 4):
per-kernel time of the level launches with the lane-cooperative permutation for small levels (poseidon_coop.h) and
without (CITYPROVER_COOP_MAX=0), caps checked against the oracle. usage: bench_levels.py [n_trees=32] [k=20]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "city-rollup_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def run(n_trees, k):
    import numpy as np
    import cityprover as cp
    import oracle_lib as O
    p = cp.Prover(0)
    log_n, rate, cap_h = 12, 3, 4
    n, N = 1 << log_n, 1 << (log_n + rate)
    vals = O.splitmix64_felts(7, n_trees * k * n).reshape(n_trees * k, n)
    dv, dl, dc = p.to_device(vals), p.alloc(n_trees * k * N), p.alloc(n_trees * (4 << cap_h))
    p.commit_batch_dev(dv.ptr, k, n_trees, log_n, rate, cap_h, dl.ptr, dc.ptr)
    caps = dc.download().reshape(n_trees, -1, 4)
    for t in (0, n_trees - 1):
        want = O.commit_batch(vals[t * k:(t + 1) * k], rate, cap_h, want=("cap",))["cap"]
        assert (caps[t] == want).all(), "cap mismatch"
    p.profile_begin()
    reps = 10
    for _ in range(reps):
        p.commit_batch_dev(dv.ptr, k, n_trees, log_n, rate, cap_h, dl.ptr, dc.ptr)
    prof = p.profile_end()
    out = {name: {"launches_per_commit": d["launches"] / reps, "ms_per_commit": d["total_ms"] / reps} for name, d in prof.items()
           if "merkle" in name or "leaf" in name}
    p.close()
    return out


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        print(json.dumps(run(int(sys.argv[2]), int(sys.argv[3]))))
        sys.exit(0)
    n_trees = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    k = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    res = {}
    for label, coop, fuse in (("lane_per_state", "0", "5"), ("cooperative_one_level_per_launch", "32768", "0"), ("cooperative_fused_2", "32768", "2"),
                              ("cooperative_fused_3", "32768", "3"), ("cooperative_fused_5", "32768", "5"), ("cooperative_fused_5_le_65536", "65536", "5"),
                              ("cooperative_fused_5_le_16384", "16384", "5")):
        env = dict(os.environ, CITYPROVER_COOP_MAX=coop, CITYPROVER_COOP_FUSE=fuse)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(n_trees), str(k)], env=env, capture_output=True, text=True)
        if r.returncode != 0:
            sys.exit(r.stderr)
        res[label] = json.loads(r.stdout.strip().splitlines()[-1])
    print(json.dumps({"n_trees": n_trees, "k": k, "results": res}))
