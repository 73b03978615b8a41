"""A SHA-256 STARK proved on the GPU through the generic AIR machinery (cp_air_program + cp_stark_prove, include/cityprover.h) and
verified: the assertion the reference's own test of its SHA-256 STARK makes — the exposed digests are the SHA-256 of the inputs
(city_common_circuit/src/hash/accelerator/sha256/smartgadget.rs:505-513, after the prove / verify pair at :518-524) — on an AIR of
this repository's own (tests/sha256_air.py: 424 columns, 522 constraints of degree 3, one row per round; starkyx's AIR lives in an
absent crate and is not restated). Held three ways: proof bytes == the CPU oracle's prover (oracle/stark_air.c), both verifiers
accept, digest == hashlib; a wrong digest, a flipped message bit and a flipped carry are refused. No extended round (k1 = 0)."""
import hashlib
import os

import numpy as np
import pytest

import air_programs as A
import oracle_lib as O
import sha256_air as S

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def prover():
    import cityprover
    p = cityprover.Prover(0)
    O.lib().or_set_threads(min(16, os.cpu_count() or 1))
    yield p
    O.lib().or_set_threads(1)
    p.close()


def descs(prover, log_rows, nq=20):
    import cityprover
    rb, ch, pow_bits = 1, 2, 8
    arity = {7: (3,), 8: (4,), 10: (4,), 12: (4, 4)}[log_rows]
    b = S.program()
    gp, op = b.gpu(prover), b.oracle()
    gd = cityprover.stark_desc(log_rows, 1, 2, cityprover.fri_params(log_rows, rb, ch, pow_bits, nq, arity), S.N_COLUMNS, gp, 0, 0, n_public=S.N_PUBLIC)
    od = O.stark_desc(log_rows, 1, 2, O.fri_params(log_rows, rb, ch, pow_bits, nq, arity), S.N_COLUMNS, op, 0, 0, n_public=S.N_PUBLIC)
    return gd, od, gp


def test_the_compiled_program(prover):
    g = S.program().gpu(prover)
    try:
        info = g.info()
        assert info["n_constraints"] == 522 and info["max_constraint_degree"] == 3 and info["n_ops"] > 5000
        assert info["n_live_ops"] <= info["n_ops"]
    finally:
        g.close()


@pytest.mark.parametrize("log_rows,device_transcript", [(7, 0), (8, 1), (10, 0), (12, 1)])
def test_sha256_stark_proved_and_verified(prover, log_rows, device_transcript):
    import cityprover
    msg = S.random_message(100 + log_rows, log_rows)
    trace, dg = S.trace(msg, log_rows)
    assert S.digest_bytes(dg) == hashlib.sha256(msg).digest()
    (gd, gk), (od, ok), gp = descs(prover, log_rows)
    prover.set_device_transcript(device_transcript)
    try:
        gc = cityprover.ChallengerState()
        gc.observe(dg)                        # the caller's protocol observes the public inputs first
        proof = cityprover.stark_prove(prover, gd, trace, gc, publics=dg)
        oc = O.challenger_new()
        O.challenger_observe(oc, dg)
        assert proof == O.stark_prove(od, trace, oc, publics=dg)
        assert gc.as_tuple() == O.challenger_tuple(oc)
        v = cityprover.ChallengerState()
        v.observe(dg)
        cityprover.stark_verify(gd, v, proof, publics=dg)
        assert v.as_tuple() == gc.as_tuple()
        w = O.challenger_new()
        O.challenger_observe(w, dg)
        assert O.stark_verify(od, w, proof, publics=dg) == 0
        # the same proof against another digest: the last-row constraint fails at zeta
        wrong = list(dg)
        wrong[7] ^= 1 << 31
        v = cityprover.ChallengerState()
        v.observe(dg)
        with pytest.raises(cityprover.CityProverError, match="quotient identity fails at zeta"):
            cityprover.stark_verify(gd, v, proof, publics=wrong)
        if log_rows <= 8:
            # cheating traces are proved all the same and refused by the verifier: a message bit, a carry, a round selector
            for col, row in ((S.W0B + 3, 5), (S.CE + 1, 77), (S.SEL + 1, 0)):
                tt = trace.copy()
                tt[col, row] ^= 1
                p2 = cityprover.stark_prove(prover, gd, tt, cityprover.ChallengerState(), publics=dg)
                with pytest.raises(cityprover.CityProverError, match="quotient identity fails at zeta"):
                    cityprover.stark_verify(gd, cityprover.ChallengerState(), p2, publics=dg)
    finally:
        prover.set_device_transcript(-1)
        gp.close()


def test_quotient_of_the_sha256_air_matches_the_oracle(prover):
    """cp_air_quotient_commit alone on the satisfied trace: coefficients and cap == the oracle's quotient"""
    import cityprover
    log_rows = 8
    trace, dg = S.trace(S.random_message(5, log_rows), log_rows)
    b = S.program()
    g, o = b.gpu(prover), b.oracle()
    T = cityprover.PolyBatch(prover, trace, 1, 2)
    ot = O.Batch(trace, 1, 2)
    try:
        alphas = [3, 5]
        Q = cityprover.air_quotient_commit(prover, g, [T], 1, alphas, publics=dg)
        oq = O.air_quotient(o, [ot], 1, alphas, publics=dg)
        want = O.Batch(oq, 1, 2, from_coeffs=True)
        assert (Q.coeffs() == want.coeffs()).all()
        assert (Q.cap() == want.cap()).all()
        # a satisfied AIR: the quotient of degree < 2n has nothing above it — every chunk is a polynomial of degree < n by construction,
        # and the top chunk of a degree-3 constraint set is not identically zero
        assert Q.coeffs().any()
        Q.close()
        want.close()
    finally:
        T.close()
        ot.close()
        g.close()
