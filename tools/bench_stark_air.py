#!/usr/bin/env python3
"""The STARK's own two steps on the device (include/cityprover.h cp_air_quotient_commit, cp_stark_prove) at the shapes of
city-rollup's SHA-256 `ByteStark` (city_common_circuit/src/hash/accelerator/sha256/smartgadget.rs:55-79: 418 + 912 columns;
:310-312: 2^k rows; :518-524: the prove call): a seeded straight-line constraint program of >= 10^4 ops in the shape of an
instruction-list AIR (tests/air_programs.py gadget_program: gadgets of 4-12 columns, 10-40 arithmetic ops, 3-8 constraints, a few
shared subexpressions; degree 3, every column read, 2 challenges, rate_bits 1 — the AIR itself lives in an absent crate, so the
program is synthetic; tests/test_gpu_air.py
holds the same programs against the CPU oracle's bytes) evaluated on the quotient coset straight from the committed traces,
alpha-folded, divided by Z_H, transformed and committed. Reported: the quotient alone (kernel time of the interpreter launch and
the whole cp_air_quotient_commit call), and the whole prover (cp_stark_prove: 418-column trace in HBM -> extended columns by a
map program + 304 cubic inversions + 912 prefix sums -> commitments -> quotient -> openings -> FRI with 84 queries) whose proof
cp_stark_verify's transcript / FRI half accepts (the random program is not satisfied by a random trace, so the constraint
identity at zeta is NOT part of the check here; the toy AIR with a lookup in tests/test_gpu_air.py is proved AND verified).
One JSON line, or run(prover, log_rows) from bench.py."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "city-rollup_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import cityprover as cp  # noqa: E402
import air_programs as A  # noqa: E402
from bench_stark_fri import arity_for  # noqa: E402

K0, K1 = 418, 912
_cache = {}


def programs(n_ops=10500, seed=77):
    """(constraint program, map program) as Builders; cached (building 10^4 ops in Python takes a moment)"""
    key = (n_ops, seed)
    if key not in _cache:
        cons = A.gadget_program(seed, K0 + K1, n_ops, n_public=4, n_global=0, n_challenge=6, max_degree=3)
        m = A.Builder(A.MAP, K0 + K1, n_public=4, n_challenge=6, n_out_columns=K1)
        ch = [m.challenge(i) for i in range(6)]
        for j in range(K1):
            v = m.add(m.mul(m.local(j % K0), ch[j % 6]), m.next((7 * j + 1) % K0))
            m.store(j, m.sub(v, m.public(j % 4)) if j % 3 else v)
        _cache[key] = (cons, m)
    return _cache[key]


def run(prover, log_rows, reps=3, seed=1, n_ops=10500):
    rb, ch, pow_bits, nq, q, na = 1, 4, 16, 84, 1, 2
    n = 1 << log_rows
    rng = np.random.default_rng(seed)
    cons_b, map_b = programs(n_ops)
    cons, mp = cons_b.gpu(prover), map_b.gpu(prover)
    info = cons.info()
    pub = rng.integers(0, cp.P, 4, dtype=np.uint64)
    cha = rng.integers(0, cp.P, 6, dtype=np.uint64)
    alphas = rng.integers(0, cp.P, na, dtype=np.uint64)
    t0v = rng.integers(0, cp.P, size=(K0, n), dtype=np.uint64)
    t1v = rng.integers(0, cp.P, size=(K1, n), dtype=np.uint64)
    T = [cp.PolyBatch(prover, t0v, rb, ch), cp.PolyBatch(prover, t1v, rb, ch)]
    try:
        # ---- the quotient alone ----
        t_call, prof = [], None
        for it in range(reps + 1):
            if it == reps:
                prover.profile_begin()
            prover.sync()
            t0 = time.perf_counter()
            Q = cp.air_quotient_commit(prover, cons, T, q, alphas, pub, None, cha)
            t1 = time.perf_counter()
            if it == reps:
                prof = prover.profile_end()
            elif it:
                t_call.append(t1 - t0)
            Q.close()
        t_call.append(t1 - t0)
        kern = {k: v["total_ms"] / max(v["launches"], 1) for k, v in prof.items() if not k.startswith(("host:", "wait:"))}
        # ---- the whole prover ----
        fri = cp.fri_params(log_rows, rb, ch, pow_bits, nq, arity_for(log_rows, rb, ch))
        desc, keep = cp.stark_desc(log_rows, q, na, fri, K0, cons, K1, 6, n_public=4,
                                   steps=[("map", mp), ("cubic_inverse", 0, K1 // 3, A.CUBIC_MODULUS), ("prefix_sum", 0, K1, False)])
        t_prove, proof = [], b""
        for it in range(reps + 1):
            st = cp.ChallengerState()
            prover.sync()
            t0 = time.perf_counter()
            proof = cp.stark_prove(prover, desc, t0v, st, publics=pub)
            t1 = time.perf_counter()
            if it:
                t_prove.append(t1 - t0)
        checked = False
        try:
            cp.stark_verify(desc, cp.ChallengerState(), proof, publics=pub)
        except cp.CityProverError as e:
            # the only acceptable refusal: the (unsatisfied) constraints at zeta; transcript, shape and parsing got that far
            checked = "quotient identity fails at zeta" in str(e)
        med = lambda v: sorted(v)[len(v) // 2] * 1e3
        M = n << q
        return {"log_rows": log_rows, "columns": [K0, K1], "rate_bits": rb, "quotient_degree_bits": q, "num_challenges": na,
                "program": {k: info[k] for k in ("n_ops", "n_live_ops", "n_constraints", "max_constraint_degree", "n_slots", "n_instructions")},
                "stark_quotient_ms": med(t_call), "interpreter_kernel_ms": kern.get("air_quotient"), "quotient_kernels_ms": kern,
                "points": M, "lane_instructions": M * info["n_instructions"],
                "G_interpreted_instructions_per_s": M * info["n_instructions"] / (kern["air_quotient"] * 1e-3) / 1e9 if kern.get("air_quotient") else None,
                "column_bytes_read": 8 * M * sum(1 for o in cons_b.ops if o[0] in (A.LOCAL, A.NEXT)),
                "stark_prove_ms": med(t_prove), "stark_proof_bytes": len(proof), "prover_ran_to_the_last_byte": checked,
                "note": "stark_quotient_ms = cp_air_quotient_commit (interpreter launch + finish + coset iNTT + commitment of the 4 chunk "
                        "polynomials) from committed traces; stark_prove_ms = cp_stark_prove from a host trace (PCIe-inclusive: 418 x n x 8 B in)"}
    finally:
        for t in T:
            t.close()
        cons.close()
        mp.close()


def run_sha256(prover, log_rows, reps=3, seed=9):
    """A SATISFIED AIR through the same prover: SHA-256 of one message of 2^log_rows / 64 - 1 blocks (tests/sha256_air.py: this
    repository's own AIR — 424 columns, 522 constraints of degree 3, 5.8 K recorded ops, no extended round; starkyx's is in an
    absent crate), at the STARK's FRI configuration (rate 2, 84 queries, 16-bit PoW). Here cp_stark_verify checks EVERYTHING — the
    constraint identity at zeta included — and the exposed digest is compared with hashlib: the assertion of the reference's own
    test (smartgadget.rs:505-513)."""
    import hashlib
    import sha256_air as S
    rb, ch, pow_bits, nq, q, na = 1, 4, 16, 84, 1, 2
    t0 = time.perf_counter()
    msg = S.random_message(seed, log_rows)
    trace, dg = S.trace(msg, log_rows)
    t_trace = time.perf_counter() - t0
    prog = S.program().gpu(prover)
    try:
        fri = cp.fri_params(log_rows, rb, ch, pow_bits, nq, arity_for(log_rows, rb, ch))
        desc, keep = cp.stark_desc(log_rows, q, na, fri, S.N_COLUMNS, prog, 0, 0, n_public=S.N_PUBLIC)
        ts, proof = [], b""
        for it in range(reps + 1):
            st = cp.ChallengerState()
            st.observe(dg)
            prover.sync()
            t0 = time.perf_counter()
            proof = cp.stark_prove(prover, desc, trace, st, publics=dg)
            if it:
                ts.append(time.perf_counter() - t0)
        v = cp.ChallengerState()
        v.observe(dg)
        t0 = time.perf_counter()
        cp.stark_verify(desc, v, proof, publics=dg)     # raises unless transcript, constraint identity at zeta and FRI all hold
        t_verify = time.perf_counter() - t0
        info = prog.info()
        return {"log_rows": log_rows, "columns": S.N_COLUMNS, "constraints": info["n_constraints"], "ops": info["n_ops"],
                "max_constraint_degree": info["max_constraint_degree"], "message_bytes": len(msg), "sha256_blocks": (1 << log_rows) // 64 - 1,
                "stark_prove_ms": sorted(ts)[len(ts) // 2] * 1e3, "proof_bytes": len(proof), "verified": True, "host_verify_ms": t_verify * 1e3,
                "digest_equals_hashlib": S.digest_bytes(dg) == hashlib.sha256(msg).digest(), "host_trace_generation_s_python": t_trace,
                "note": "cp_stark_prove from a host trace (PCIe-inclusive), proof accepted by cp_stark_verify in full"}
    finally:
        prog.close()


if __name__ == "__main__":
    p = cp.Prover(0)
    sizes = [int(a) for a in sys.argv[1:]] or [10, 12, 14, 16]
    print(json.dumps({"what": "SHA-256-STARK-shaped quotient and whole prover through cp_air_quotient_commit / cp_stark_prove",
                      "cases": [run(p, k) for k in sizes], "sha256": [run_sha256(p, k) for k in sizes]}))
    p.close()
