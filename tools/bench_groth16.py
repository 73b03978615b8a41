#!/usr/bin/env python3
"""Timing of cp_groth16_prove_bls12381 (five MSMs + the quotient + host assembly) at 2^k constraints / wires with a
synthetic proving key (points (a i + b) G built on the device) and a witness-like scalar vector: 60 % of the wires in
{0, 1}, the rest uniform. Correctness of the assembly is tests/test_gpu_groth16.py; this only measures."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "city-rollup_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np  # noqa: E402
import cityprover as cp  # noqa: E402
from bench_msm import G, G2  # noqa: E402


def run(prover, log_n, reps=3):
    n = 1 << log_n
    rng = np.random.default_rng(log_n)
    w = rng.integers(0, 2**64, (n, 4), dtype=np.uint64)
    w[:, 3] &= np.uint64((1 << 62) - 1)
    small = rng.random(n) < 0.6
    w[small, 1:] = 0
    w[small, 0] = rng.integers(0, 2, int(small.sum()), dtype=np.uint64)
    ev = rng.integers(0, 2**64, (n, 4), dtype=np.uint64)
    ev[:, 3] &= np.uint64((1 << 62) - 1)
    sets = [cp.G1Points.synthetic(prover, G, 3, 1, n), cp.G1Points.synthetic(prover, G, 5, 2, n), cp.G2Points.synthetic(prover, G2, 7, 3, n),
            cp.G1Points.synthetic(prover, G, 11, 4, n), cp.G1Points.synthetic(prover, G, 13, 5, n)]
    pk = cp.Groth16Pk()
    pk.n_wires, pk.n_private, pk.log_domain = n, n - 16, log_n
    pk.a_g1, pk.b_g1, pk.b_g2, pk.k_g1, pk.z_g1 = (s.buf.ptr for s in sets)
    pk.a_inf = pk.b_inf = None
    g1 = [(int(G[h]) >> (64 * i)) & (2**64 - 1) for h in range(2) for i in range(6)]
    g2 = [(int(c) >> (64 * i)) & (2**64 - 1) for c in (G2[0][0], G2[0][1], G2[1][0], G2[1][1]) for i in range(6)]
    pk.alpha_g1[:], pk.beta_g1[:], pk.delta_g1[:] = g1, g1, g1
    pk.beta_g2[:], pk.delta_g2[:] = g2, g2
    dw = prover.to_device(w)
    bufs = [prover.to_device(ev) for _ in range(3)]
    ts = []
    for _ in range(reps + 1):
        for b in bufs:
            b.upload(ev)
        prover.sync()
        t0 = time.perf_counter()
        cp.groth16_prove(prover, pk, dw.ptr, bufs[0].ptr, bufs[1].ptr, bufs[2].ptr, 12345, 67890)
        ts.append(time.perf_counter() - t0)
    for d in [dw] + bufs:
        d.free()
    for s in sets:
        s.free()
    return {"log_constraints": log_n, "wires": n, "prove_ms": sorted(ts[1:])[len(ts[1:]) // 2] * 1e3}


if __name__ == "__main__":
    sizes = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [16, 18, 20, 22]
    p = cp.Prover(0)
    out = [run(p, s) for s in sizes]
    p.close()
    print(json.dumps(out))
