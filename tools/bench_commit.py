#!/usr/bin/env python3
"""Product-shape commitment throughput: for B proofs in flight, the three oracle commitments of a
plonky2 proof at n = 2^12, rate 8, cap height 4 (135 wires, 20 Z/partial-product, 16 quotient columns;
SURVEY.md §8(a) A3+A4). Prints per-kernel HIP-event times and commits/s."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "city-rollup_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import cityprover as cp  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    p = cp.Prover(0)
    log_n, rate, cap_h, n = 12, 3, 4, 4096
    N = n << rate
    rng = np.random.default_rng(1)
    shapes = [135, 20, 16]
    bufs = []
    for k in shapes:
        vals = (rng.integers(0, 2**63, (B * k, n), dtype=np.uint64) % np.uint64(cp.P))
        bufs.append((k, p.to_device(vals), p.alloc(B * k * N), p.alloc(B * k * n),
                     p.alloc(B * (2 * N - (2 << cap_h)) * 4), p.alloc(B * (4 << cap_h))))

    def run():
        for k, dv, dl, dco, dd, dcap in bufs:
            p.commit_batch_dev(dv.ptr, k, B, log_n, rate, cap_h, dl.ptr, dcap.ptr, dco.ptr, dd.ptr)

    run()
    p.sync()
    p.profile_begin()
    t0 = time.perf_counter()
    for _ in range(iters):
        run()
    p.sync()
    t1 = time.perf_counter()
    prof = p.profile_end()
    ms = (t1 - t0) * 1e3 / iters
    print(json.dumps({"B": B, "ms_per_batch": ms, "ms_per_proof_commitments": ms / B,
                      "proof_commitments_per_s": B / (ms * 1e-3),
                      "kernels_ms_per_batch": {k: v["total_ms"] / iters for k, v in prof.items()},
                      "launches_per_batch": {k: v["launches"] / iters for k, v in prof.items()}}))


if __name__ == "__main__":
    main()
