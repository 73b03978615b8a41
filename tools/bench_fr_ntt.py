#!/usr/bin/env python3
"""BLS12-381 scalar-field NTT timing (cp_ntt_bls12381_fr_dev: data resident in HBM in canonical form, in place,
natural order in and out), with an inverse-of-forward round-trip check at every size."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "city-rollup_amd"))
import numpy as np  # noqa: E402
import cityprover as cp  # noqa: E402


def run(prover, log_n, reps=5):
    n = 1 << log_n
    rng = np.random.default_rng(log_n)
    a = rng.integers(0, 2**64, (n, 4), dtype=np.uint64)
    a[:, 3] &= np.uint64((1 << 62) - 1)
    d = prover.to_device(a)
    cp.fr_ntt_dev(prover, d.ptr, log_n)
    cp.fr_ntt_dev(prover, d.ptr, log_n, inverse=True)
    prover.sync()
    assert (d.download().reshape(n, 4) == a).all(), "inverse(forward(x)) != x"
    out = {"log_n": log_n}
    def median_ms(fn):
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            prover.sync()
            ts.append(time.perf_counter() - t0)
        return sorted(ts)[len(ts) // 2] * 1e3

    for name, kw in (("forward", {}), ("inverse", {"inverse": True}), ("coset_forward", {"shift": 7})):
        cp.fr_ntt_dev(prover, d.ptr, log_n, **kw)
        prover.profile_begin()
        out[name + "_ms"] = median_ms(lambda: cp.fr_ntt_dev(prover, d.ptr, log_n, **kw))
        prof = prover.profile_end()
        if name == "forward":
            out["kernels_ms"] = {k: round(v["total_ms"] / reps, 3) for k, v in prof.items() if k.startswith("fr_")}
            out["algorithmic_GBs"] = 64.0 * n / (out[name + "_ms"] * 1e-3) / 1e9     # read + write the canonical array once
    # Groth16 quotient (3 x (iNTT + coset NTT) + pointwise + coset iNTT) on the same data
    d2, d3 = prover.to_device(a), prover.to_device(a)
    cp.groth16_quotient_dev(prover, d.ptr, d2.ptr, d3.ptr, log_n)
    prover.sync()
    out["groth16_quotient_ms"] = median_ms(lambda: cp.groth16_quotient_dev(prover, d.ptr, d2.ptr, d3.ptr, log_n))
    for x in (d, d2, d3):
        x.free()
    return out


if __name__ == "__main__":
    sizes = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [16, 20, 22, 24]
    p = cp.Prover(0)
    out = [run(p, s) for s in sizes]
    p.close()
    print(json.dumps(out))
