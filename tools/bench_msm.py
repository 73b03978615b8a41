#!/usr/bin/env python3
"""BLS12-381 G1 MSM throughput (cp_msm_bls12381_g1_dev, points resident in the library's internal form, scalars
resident on the device, result on the host) with a closed-form correctness check at every size:
points P_i = (a i + b) G, uniformly random 255-bit scalars  =>  MSM = (sum k_i (a i + b) mod r) * G."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "city-rollup_amd"))
import numpy as np  # noqa: E402
import cityprover as cp  # noqa: E402

X = -0xd201000000010000
R = X**4 - X**2 + 1
P = (X - 1)**2 * R // 3 + X
G = (0x17f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb,
     0x08b3f481e3aaa0f1a09e30ed741d8ae4fcf5e095d5d00af600db18cb2c04b3edd03cc744a2888ae40caa232946c5e7e1)


def run(prover, log_n, reps=3, a=7, b=3):
    n = 1 << log_n
    pts = cp.G1Points.synthetic(prover, G, a, b, n)
    rng = np.random.default_rng(log_n)
    k = rng.integers(0, 2**64, (n, 4), dtype=np.uint64)
    k[:, 3] >>= np.uint64(1)
    ds = prover.to_device(k)
    got = pts.msm_dev(ds.ptr)   # warm-up + check
    ks = [sum(int(k[i, j]) << (64 * j) for j in range(4)) for i in range(n)] if n <= (1 << 18) else None
    expected_scalar = None
    if ks is not None:
        expected_scalar = sum(kk * (a * i + b) for i, kk in enumerate(ks)) % R
        one = cp.G1Points.synthetic(prover, G, 1, 1, 1)   # the single point 1*G ... (1*0 + 1) G
        e = np.array([[(expected_scalar >> (64 * j)) & (2**64 - 1) for j in range(4)]], dtype=np.uint64)
        de = prover.to_device(e)
        want = one.msm_dev(de.ptr)
        de.free()
        one.free()
        assert got == want, "MSM result differs from (sum k_i (a i + b)) * G"
    prover.profile_begin()
    t0 = time.perf_counter()
    for _ in range(reps):
        pts.msm_dev(ds.ptr)
    dt = (time.perf_counter() - t0) / reps
    prof = prover.profile_end()
    ds.free()
    pts.free()
    return {"log_n": log_n, "ms": dt * 1e3, "Mpoints_per_s": n / dt / 1e6, "checked": ks is not None,
            "kernels_ms": {kk: round(v["total_ms"] / reps, 3) for kk, v in prof.items() if kk.startswith("msm")}}


G2 = ((0x024aa2b2f08f0a91260805272dc51051c6e47ad4fa403b02b4510b647ae3d1770bac0326a805bbefd48056c8c121bdb8,
       0x13e02b6052719f607dacd3a088274f65596bd0d09920b61ab5da61bbdc7f5049334cf11213945d57e5ac7d055d042b7e),
      (0x0ce5d527727d6e118cc9cdc6da2e351aadfd9baa8cbdd3a76d429a695160d12c923ac9cc3baca289e193548608b82801,
       0x0606c4a02ea734cc32acd2b02bc28b99cb3e287e85a763af267492ab572e99ab3f370d275cec1da1aaa9075ff05f79be))


def run_g2(prover, log_n, reps=3, a=7, b=3):
    """Same on G2; the check compares with a one-point MSM (expected scalar) * G2 computed by the same library."""
    n = 1 << log_n
    pts = cp.G2Points.synthetic(prover, G2, a, b, n)
    rng = np.random.default_rng(log_n)
    k = rng.integers(0, 2**64, (n, 4), dtype=np.uint64)
    k[:, 3] >>= np.uint64(1)
    ds = prover.to_device(k)
    got = pts.msm_dev(ds.ptr)
    checked = n <= (1 << 18)
    if checked:
        ks = [sum(int(k[i, j]) << (64 * j) for j in range(4)) for i in range(n)]
        e = sum(kk * (a * i + b) for i, kk in enumerate(ks)) % R
        one = cp.G2Points.synthetic(prover, G2, 1, 1, 1)
        de = prover.to_device(np.array([[(e >> (64 * j)) & (2**64 - 1) for j in range(4)]], dtype=np.uint64))
        assert got == one.msm_dev(de.ptr), "G2 MSM result differs from (sum k_i (a i + b)) * G2"
        de.free()
        one.free()
    prover.profile_begin()
    t0 = time.perf_counter()
    for _ in range(reps):
        pts.msm_dev(ds.ptr)
    dt = (time.perf_counter() - t0) / reps
    prof = prover.profile_end()
    ds.free()
    pts.free()
    return {"group": "G2", "log_n": log_n, "ms": dt * 1e3, "Mpoints_per_s": n / dt / 1e6, "checked": checked,
            "kernels_ms": {kk: round(v["total_ms"] / reps, 3) for kk, v in prof.items() if kk.startswith("msm")}}


if __name__ == "__main__":
    sizes = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [12, 16, 18, 20]
    p = cp.Prover(0)
    out = [dict(run(p, s), group="G1") for s in sizes]
    out += [run_g2(p, s) for s in sizes if s <= 20]
    p.close()
    print(json.dumps(out))
