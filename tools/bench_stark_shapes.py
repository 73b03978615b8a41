#!/usr/bin/env python3
"""SURVEY.md §8(a) A13 / §8(f) N3 scoping: what the commitments of the starkyx SHA-256 ByteStark cost on the kernels that
exist (`PolynomialBatch::from_values` = iNTT + coset LDE + Poseidon Merkle cap, csrc cp_commit_dev) at the STARK's shape:
418 free + 912 extended trace columns (city_common_circuit/src/hash/accelerator/sha256/smartgadget.rs:55-79), 2^k rows
(k = ceil(log2(64 * number of 64-byte SHA chunks)), smartgadget.rs:310-312), blow-up 2 (Starky's standard_fast_config,
UPSTREAM-MEMORY). The first (k, columns) case is checked against the CPU oracle. One JSON line."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "city-rollup_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import cityprover as cp  # noqa: E402
import oracle_lib as O  # noqa: E402


def main():
    p = cp.Prover(0)
    rate_bits, cap_h = 1, 4
    out = []
    checked = False
    for log_n in (12, 14, 16):
        n, N = 1 << log_n, 1 << (log_n + rate_bits)
        for k in (418, 912):
            vals = O.splitmix64_felts(log_n * 1000 + k, k * n).reshape(k, n)
            dv, dl, dc = p.to_device(vals), p.alloc(k * N), p.alloc(4 << cap_h)
            p.commit_dev(dv.ptr, k, log_n, rate_bits, cap_h, dl.ptr, dc.ptr)
            cap = dc.download().reshape(-1, 4)
            if not checked:
                O.lib().or_set_threads(os.cpu_count() or 1)
                want = O.commit_batch(vals, rate_bits, cap_h, want=("cap",))["cap"]
                O.lib().or_set_threads(1)
                assert (cap == want).all(), "STARK-shaped commitment differs from the oracle's"
                checked = True
            p.sync()
            reps = 5
            t0 = time.perf_counter()
            for _ in range(reps):
                p.commit_dev(dv.ptr, k, log_n, rate_bits, cap_h, dl.ptr, dc.ptr)
            p.sync()
            ms = (time.perf_counter() - t0) * 1e3 / reps
            out.append({"log_rows": log_n, "columns": k, "rate_bits": rate_bits, "commit_ms": round(ms, 3),
                        "leaf_permutations": N * ((k + 7) // 8), "lde_MB": k * N * 8 / 1e6})
            for b in (dv, dl, dc):
                b.free()
    p.close()
    print(json.dumps({"what": "PolynomialBatch::from_values at the SHA-256 ByteStark's trace shapes (iNTT + coset LDE x2 + Poseidon Merkle cap), "
                              "values resident in HBM", "first_case_checked_against_oracle": checked, "cases": out}))


if __name__ == "__main__":
    main()
