#!/usr/bin/env python3
"""Throughput / latency of cp_prove_tail_batch at the product shape (standard_recursion_config,
n = 2^12, 130 360-byte proofs) for several batch sizes B on one context."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "city-rollup_amd"))
import numpy as np  # noqa: E402
import cityprover as cp  # noqa: E402


def felts(shape, seed):
    rng = np.random.default_rng(seed)
    return rng.integers(0, 2**63, shape, dtype=np.uint64) % np.uint64(cp.P)


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    batches = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 4, 16, 64]
    p = cp.Prover(0)
    sh = cp.standard_recursion_shape(num_public_inputs=8)
    n = 4096
    circ = cp.Circuit(p, sh, [1, 2, 3, 4], felts((85, n), 1))
    res = {}
    for B in batches:
        w, z, q = p.to_device(felts((B, 135, n), 2)), p.to_device(felts((B, 20, n), 3)), p.to_device(felts((B, 16, n), 4))
        pis = [np.arange(8, dtype=np.uint64) + i for i in range(B)]
        circs = [circ] * B
        cp.prove_tail_batch_dev(p, circs, pis, w.ptr, z.ptr, q.ptr)  # warm-up (arena, tables)
        t0 = time.perf_counter()
        for _ in range(iters):
            proofs = cp.prove_tail_batch_dev(p, circs, pis, w.ptr, z.ptr, q.ptr)
        t1 = time.perf_counter()
        p.profile_begin()
        cp.prove_tail_batch_dev(p, circs, pis, w.ptr, z.ptr, q.ptr)
        prof = p.profile_end()
        ms = (t1 - t0) * 1e3 / iters
        res[f"B={B}"] = {"ms_per_batch": round(ms, 3), "ms_per_proof": round(ms / B, 4), "proofs_per_s": round(B / ms * 1e3, 1),
                        "proof_len": len(proofs[0]),
                        "kernel_ms_per_batch": round(sum(v["total_ms"] for v in prof.values()), 3),
                        "top_kernels_ms": {k: round(v["total_ms"], 3) for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["total_ms"])[:6]},
                        "launches": int(sum(v["launches"] for v in prof.values()))}
        for b in (w, z, q):
            b.free()
    print(json.dumps(res))


if __name__ == "__main__":
    main()
