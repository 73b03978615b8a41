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


R = (lambda x: x**4 - x**2 + 1)(-0xd201000000010000)   # the group order


def limbs_sum(k, lo=0, hi=None, chunk=1 << 20):
    """(sum_i k_i, sum_i i * k_i) over rows lo..hi of an (n, 4) uint64 limb array, i counted from 0 at row lo — exact, in numpy:
    32-bit half-limbs, rows taken in chunks of 2^20 (the index inside a chunk split at 2^10) so that no partial sum passes 2^63
    whatever n is: 2^32 * 2^10 * 2^20 = 2^62"""
    k = k[lo:hi]
    s0 = s1 = 0
    for base in range(0, k.shape[0], chunk):
        kc = k[base:base + chunk]
        idx = np.arange(kc.shape[0], dtype=np.uint64)
        i_lo, i_hi = idx & np.uint64(1023), idx >> np.uint64(10)
        c0 = c1 = 0
        for j in range(4):
            for h in range(2):
                half = (kc[:, j] >> np.uint64(32 * h)) & np.uint64(0xFFFFFFFF)
                w = 1 << (64 * j + 32 * h)
                c0 += int(half.sum(dtype=np.uint64)) * w
                c1 += (int((half * i_lo).sum(dtype=np.uint64)) + (int((half * i_hi).sum(dtype=np.uint64)) << 10)) * w
        s0 += c0
        s1 += c1 + base * c0
    return s0, s1


def scalar_mul_g1(prover, log):
    """[log] G through a one-point MSM (itself held against the CPU oracle by tests/test_gpu_msm.py)"""
    one = cp.G1Points.synthetic(prover, G, 1, 1, 1)
    de = prover.to_device(np.array([[(log >> (64 * j)) & (2**64 - 1) for j in range(4)]], dtype=np.uint64))
    out = one.msm_dev(de.ptr)
    de.free()
    one.free()
    return out


def scalar_mul_g2(prover, log):
    one = cp.G2Points.synthetic(prover, G2, 1, 1, 1)
    de = prover.to_device(np.array([[(log >> (64 * j)) & (2**64 - 1) for j in range(4)]], dtype=np.uint64))
    out = one.msm_dev(de.ptr)
    de.free()
    one.free()
    return out


def run(prover, log_n, reps=3, check=True):
    """check: the three proof elements against the trapdoor of the synthetic key — every key point is [a i + b] G with known
    (a, b), alpha = beta = delta = G, so the discrete logarithm of A, B and C is a closed form in the witness, the quotient
    coefficients (read back from the device) and r, s:  A = 1 + sum w_i (3 i + 1) + r,  B = 1 + sum w_i (7 i + 3) + s,
    C = sum_priv w_i (11 i' + 4) + sum h_j (13 j + 5) + s A + r B1 - r s  with B1 = 1 + sum w_i (5 i + 2) + s."""
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
    rr, ss = 12345, 67890
    kern = None
    for it in range(reps + 1):
        for b in bufs:
            b.upload(ev)
        prover.sync()
        if it == reps:
            prover.profile_begin()   # the last repetition with per-kernel HIP events (the calling context's chain: A, B1, quotient, Z)
        t0 = time.perf_counter()
        A, B, C = cp.groth16_prove(prover, pk, dw.ptr, bufs[0].ptr, bufs[1].ptr, bufs[2].ptr, rr, ss)
        ts.append(time.perf_counter() - t0)
        if it == reps:
            prof = prover.profile_end()
            kern = {k: round(v["total_ms"], 3) for k, v in prof.items() if k.startswith(("msm", "fr_", "groth16"))}
    checked = False
    if check:
        n_pub = n - pk.n_private
        h = bufs[0].download().reshape(n, 4)          # the quotient's coefficients replace the first evaluation vector
        w0, w1 = limbs_sum(w)
        a_log = (1 + 3 * w1 + 1 * w0 + rr) % R
        b_log = (1 + 7 * w1 + 3 * w0 + ss) % R
        b1_log = (1 + 5 * w1 + 2 * w0 + ss) % R
        p0, p1 = limbs_sum(w, n_pub, n)
        h0, h1 = limbs_sum(h, 0, n - 1)
        c_log = (11 * p1 + 4 * p0 + 13 * h1 + 5 * h0 + ss * a_log + rr * b1_log - rr * ss) % R
        assert A == scalar_mul_g1(prover, a_log), "Groth16 A differs from the trapdoor's"
        assert B == scalar_mul_g2(prover, b_log), "Groth16 B differs from the trapdoor's"
        assert C == scalar_mul_g1(prover, c_log), "Groth16 C differs from the trapdoor's"
        checked = True
    for d in [dw] + bufs:
        d.free()
    for s in sets:
        s.free()
    tail = sum(v for k, v in (kern or {}).items() if k in ("msm_segment_reduce", "msm_pair_reduce"))
    allk = sum((kern or {}).values())
    return {"log_constraints": log_n, "wires": n, "prove_ms": sorted(ts[1:])[len(ts[1:]) // 2] * 1e3, "checked": checked,
            "main_chain_kernels_ms": kern, "reduction_tail_share_of_main_chain": tail / allk if allk else None}


if __name__ == "__main__":
    sizes = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [16, 18, 20, 22]
    p = cp.Prover(0)
    out = [run(p, s) for s in sizes]
    p.close()
    print(json.dumps(out))
