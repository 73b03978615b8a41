"""A/B timing of the NTT passes: 135 forward 2^20 transforms (BASELINE configs[1]'s NTT half) and the product-shape commit (2^12 values ->
2^15 LDE), per-kernel HIP-event times from cp_profile. usage: python tools/bench_ntt.py [--reps 20] [--check]"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "city-rollup_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import cityprover  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--check", action="store_true", help="compare one 2^20 transform and one 2^12 commit with the CPU oracle")
    a = ap.parse_args()
    p = cityprover.Prover(0)
    out = {}
    k, log_n = 135, 20
    n = 1 << log_n
    rng = np.random.default_rng(1)
    vals = rng.integers(0, cityprover.P, size=(k, n), dtype=np.uint64)
    d = p.to_device(vals)
    if a.check:
        import oracle_lib as O
        O.lib().or_set_threads(8)
        p.ntt_dev(d.ptr, log_n, 1, n, cityprover.NTT_BITREV_OUT)
        got = d.download(n)
        want = O.bit_reverse(O.ntt(vals[0]))
        assert (got == want).all(), "2^20 NTT differs from the oracle"
        d.upload(vals)
    for _ in range(3):
        p.ntt_dev(d.ptr, log_n, k, n, cityprover.NTT_BITREV_OUT)
    p.sync()
    p.profile_begin()
    e0, e1 = p.event(), p.event()
    p.record(e0)
    for _ in range(a.reps):
        p.ntt_dev(d.ptr, log_n, k, n, cityprover.NTT_BITREV_OUT)
    p.record(e1)
    ms = p.elapsed_ms(e0, e1)
    prof = p.profile_end()
    out["ntt_2^20_x135"] = {"ms_per_step": ms / a.reps, "us_per_ntt": 1000 * ms / a.reps / k,
                            "kernels_ms": {kk: v["total_ms"] / v["launches"] for kk, v in prof.items() if not kk.startswith(("host:", "wait:"))},
                            "tb_per_s_vs_algorithmic": 16 * n * k / (ms / a.reps * 1e-3) / 1e12}
    d.free()
    # product shape: 32 proofs x 135 wires, 2^12 -> 2^15 (values -> coefficients -> LDE), no hashing
    B, kw, ln, rb = 32, 135, 12, 3
    v2 = rng.integers(0, cityprover.P, size=(B * kw, 1 << ln), dtype=np.uint64)
    dv, dc, dl = p.to_device(v2), p.alloc(B * kw << ln), p.alloc(B * kw << (ln + rb))
    if a.check:
        import oracle_lib as O
        p._check(p.lib.cp_d2d(p.ctx, dc.ptr, dv.ptr, 8 << ln))
        p.ntt_dev(dc.ptr, ln, 1, 1 << ln, cityprover.NTT_INVERSE)
        co = dc.download(1 << ln)
        assert (co == O.intt(v2[0])).all(), "2^12 iNTT differs from the oracle"
        p.lde_dev(dc.ptr, ln, rb, 1, dl.ptr)
        assert (dl.download(1 << (ln + rb)) == O.bit_reverse(O.coset_lde(co, rb))).all(), "LDE differs from the oracle"

    def step():
        p._check(p.lib.cp_d2d(p.ctx, dc.ptr, dv.ptr, (B * kw << ln) * 8))
        p.ntt_dev(dc.ptr, ln, B * kw, 1 << ln, cityprover.NTT_INVERSE)
        p.lde_dev(dc.ptr, ln, rb, B * kw, dl.ptr)
    for _ in range(3):
        step()
    p.sync()
    p.profile_begin()
    p.record(e0)
    for _ in range(a.reps):
        step()
    p.record(e1)
    ms = p.elapsed_ms(e0, e1)
    prof = p.profile_end()
    out["commit_ntt_32x135_2^12_to_2^15"] = {"ms_per_step": ms / a.reps,
                                            "kernels_ms": {kk: v["total_ms"] / v["launches"] for kk, v in prof.items() if not kk.startswith(("host:", "wait:"))}}
    for b in (dv, dc, dl):
        b.free()
    out["env"] = {kk: v for kk, v in os.environ.items() if kk.startswith("CITYPROVER_")}
    print(json.dumps(out))
    p.close()


if __name__ == "__main__":
    main()
