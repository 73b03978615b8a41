"""A SHA-256 AIR (tests/sha256_air.py: this repository's own, NOT starkyx's) on the CPU side of the generic AIR machinery: the trace
generator against hashlib, the constraints in Python integers on every row, the recorded program through the oracle's interpreter
(oracle/stark_air.c) against the same constraints evaluated directly — over F_p and over F_p^2 —, and the oracle's whole STARK prover
and verifier on it (no extended round: k1 = 0). The assertion the reference's own test of its SHA-256 STARK makes — the exposed
digest is the SHA-256 of the input (city_common_circuit/src/hash/accelerator/sha256/smartgadget.rs:505-513) — is made here too.
The GPU half is tests/test_gpu_sha256_air.py."""
import hashlib

import numpy as np
import pytest

import air_programs as A
import oracle_lib as O
import sha256_air as S

P = A.P


def holds(when, i, n):
    return when == "all" or (when == "transition" and i < n - 1) or (when == "first" and i == 0) or (when == "last" and i == n - 1)


def sinks_of(b, vals):
    return [int(vals[a]) for (op, a, _, _) in b.ops if A.ASSERT_ZERO <= op <= A.ASSERT_ZERO_LAST_ROW]


@pytest.mark.parametrize("message,log_rows", [(b"abc", 7), (b"", 7), (bytes(range(55)), 7), (None, 8), (None, 9)])
def test_trace_generator_gives_the_digest_and_satisfies_every_constraint(message, log_rows):
    if message is None:
        message = S.random_message(log_rows, log_rows)
    t, dg = S.trace(message, log_rows)
    assert S.digest_bytes(dg) == hashlib.sha256(message).digest()
    assert t.shape == (S.N_COLUMNS, 1 << log_rows) and int(t.max()) < 2**32
    n = t.shape[1]
    for i in range(n):
        loc, nxt = [int(v) for v in t[:, i]], [int(v) for v in t[:, (i + 1) % n]]
        bad = []
        S.constraints(A.IntField, loc, nxt, dg, lambda v, when: bad.append((v, when)))
        assert len(bad) == 522
        assert all(v == 0 for v, when in bad if holds(when, i, n)), i


def test_a_wrong_trace_breaks_a_constraint():
    msg = b"the quick brown fox"
    t, dg = S.trace(msg, 7)
    n = t.shape[1]

    def broken(tt, d):
        out = 0
        for i in range(n):
            got = []
            S.constraints(A.IntField, [int(v) for v in tt[:, i]], [int(v) for v in tt[:, (i + 1) % n]], d, lambda v, when: got.append((v, when)))
            out += sum(1 for v, when in got if holds(when, i, n) and v != 0)
        return out
    assert broken(t, dg) == 0
    wrong = list(dg)
    wrong[3] ^= 1
    assert broken(t, wrong) == 1                      # only the last-row constraint on that digest word
    for col, row in ((S.W0B + 5, 3), (S.X0 + 32 * 4 + 7, 20), (S.CA, 40), (S.SEL + 9, 9), (S.HC + 2, 70), (S.WIN + 9, 30)):
        tt = t.copy()
        tt[col, row] ^= 1
        assert broken(tt, dg) > 0, (col, row)


def test_recorded_program_equals_the_direct_evaluation():
    c = S.program()
    oc = c.oracle()
    assert oc.check() == 0
    assert oc.num_constraints() == 522
    assert len(c.ops) > 5000
    t, dg = S.trace(b"abc", 7)
    n = t.shape[1]
    rng = np.random.default_rng(3)
    for i in (0, 1, 17, 62, 63, 64, 126, 127):
        loc, nxt = [int(v) for v in t[:, i]], [int(v) for v in t[:, (i + 1) % n]]
        want = []
        S.constraints(A.IntField, loc, nxt, dg, lambda v, when: want.append(v))
        assert sinks_of(c, oc.eval_row(loc, nxt, publics=dg)) == want
    for _ in range(3):   # random rows: nothing vanishes, every op matters
        loc = [int(v) for v in rng.integers(0, P, S.N_COLUMNS, dtype=np.uint64)]
        nxt = [int(v) for v in rng.integers(0, P, S.N_COLUMNS, dtype=np.uint64)]
        pub = [int(v) for v in rng.integers(0, P, 8, dtype=np.uint64)]
        want = []
        S.constraints(A.IntField, loc, nxt, pub, lambda v, when: want.append(v))
        assert sinks_of(c, oc.eval_row(loc, nxt, publics=pub)) == want and all(want)
        le = [(int(a), int(b)) for a, b in rng.integers(0, P, (S.N_COLUMNS, 2), dtype=np.uint64)]
        ne = [(int(a), int(b)) for a, b in rng.integers(0, P, (S.N_COLUMNS, 2), dtype=np.uint64)]
        wante = []
        S.constraints(A.ExtField, le, ne, [(x, 0) for x in pub], lambda v, when: wante.append(v))
        out, kinds = oc.eval_ext(np.array(le, dtype=np.uint64), np.array(ne, dtype=np.uint64), publics=np.array([(x, 0) for x in pub], dtype=np.uint64))
        assert [tuple(int(x) for x in r) for r in out] == wante


def sha_descs(log_rows, backend, prog):
    rb, ch, pow_bits, nq = 1, 2, 8, 20
    arity = (3,) if log_rows >= 7 else ()
    mod = backend
    return mod.stark_desc(log_rows, 1, 2, mod.fri_params(log_rows, rb, ch, pow_bits, nq, arity), S.N_COLUMNS, prog, 0, 0, n_public=S.N_PUBLIC)


def test_oracle_proves_and_verifies_sha256():
    msg = b"city-rollup: a SHA-256 STARK on the generic AIR machinery"[:55]
    t, dg = S.trace(msg, 7)
    prog = S.program().oracle()
    d, keep = sha_descs(7, O, prog)
    c = O.challenger_new()
    O.challenger_observe(c, dg)           # the caller's protocol observes the public inputs first
    proof = O.stark_prove(d, t, c, publics=dg)
    v = O.challenger_new()
    O.challenger_observe(v, dg)
    assert O.stark_verify(d, v, proof, publics=dg) == 0
    assert O.challenger_tuple(v) == O.challenger_tuple(c)
    assert S.digest_bytes(dg) == hashlib.sha256(msg).digest()
    # another digest: the last-row constraint fails at zeta
    wrong = list(dg)
    wrong[0] ^= 1
    v = O.challenger_new()
    O.challenger_observe(v, dg)
    assert O.stark_verify(d, v, proof, publics=wrong) != 0
    # a trace with one message bit flipped and nothing else recomputed: proved all the same, refused by the verifier
    tt = t.copy()
    tt[S.W0B + 1, 2] ^= 1
    c2 = O.challenger_new()
    p2 = O.stark_prove(d, tt, c2, publics=dg)
    assert O.stark_verify(d, O.challenger_new(), p2, publics=dg) != 0
