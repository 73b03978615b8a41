"""The generic FRI seam on the CPU (no GPU needed): the oracle's `prove_openings` over arbitrary oracles / opening batches
(oracle/plonky2_tail.c or_fri_prove) against BOTH verifiers — the oracle's own (or_fri_verify) and the product's host verifier
(cp_fri_verify, the code cp_verify runs) — on seeded random instances, their transcripts (cp_challenger_* vs or_ch_*), every
kind of corruption, and a toy AIR proved the way a STARK prover built on plonky2's FRI proves
(city_common_circuit/src/hash/accelerator/sha256/smartgadget.rs:518-524 is that shape of client).
Parity note: plonky2 is not in the reference tree; what pins this code on reference data is that cp_verify / or_verify_tail run
through the very same functions on all ten reference proofs (tests/test_reference_product_parity.py, test_oracle_*_reference.py)."""
import numpy as np
import pytest

import cityprover
import oracle_lib as O
import fri_instances as F


@pytest.mark.parametrize("seed", range(12))
def test_oracle_prover_accepted_by_both_verifiers(seed):
    spec = F.random_instance(seed)
    res = F.run_instance(F.OracleBackend(), spec)
    F.verify_both(spec, res)


def test_challenger_state_matches_oracle_step_by_step():
    rng = np.random.default_rng(7)
    c, s = O.challenger_new(), cityprover.ChallengerState()
    for step in range(200):
        if rng.random() < 0.6:
            e = O.splitmix64_felts(step, int(rng.integers(0, 20)))
            O.challenger_observe(c, e)
            s.observe(e)
        else:
            k = int(rng.integers(1, 12))
            assert (O.challenger_challenges(c, k) == s.challenges(k)).all()
        assert O.challenger_tuple(c) == s.as_tuple()


def test_challenger_state_is_validated():
    s = cityprover.ChallengerState()
    s.n_input = 8
    with pytest.raises(cityprover.CityProverError, match="input_buffer"):
        s.observe([1])
    s = cityprover.ChallengerState()
    s.sponge_state[3] = O.P
    with pytest.raises(cityprover.CityProverError, match="canonical"):
        s.challenges(1)
    with pytest.raises(cityprover.CityProverError, match="canonical"):
        cityprover.ChallengerState().observe([O.P])


def _reject(spec, res, **changes):
    r = dict(res)
    r.update(changes)
    pc = cityprover.fri_params(spec["degree_bits"], spec["rate_bits"], spec["cap_height"], spec["pow_bits"], spec["num_query_rounds"],
                               spec["arity_bits"])
    s = F.replay_challenger("product", r)
    with pytest.raises(cityprover.CityProverError):
        cityprover.fri_verify(pc, r["infos"], r["caps"], r["batches"], r["opened"], s, r["proof"])
    po = O.fri_params(spec["degree_bits"], spec["rate_bits"], spec["cap_height"], spec["pow_bits"], spec["num_query_rounds"], spec["arity_bits"])
    c = F.replay_challenger("oracle", r)
    rc, _ = O.fri_verify(po, r["infos"], r["caps"], r["batches"], r["opened"], c, r["proof"])
    assert rc != 0


def test_every_corruption_is_refused_by_both_verifiers():
    spec = dict(seed=99, degree_bits=6, rate_bits=2, cap_height=2, arity_bits=(2, 1), pow_bits=3, num_query_rounds=5, ks=[4, 7, 2],
                blinding=[False, True, False], batches=[[(0, 0, 4), (1, 0, 7), (2, 0, 2)], [(1, 2, 3)], [(2, 1, 1), (0, 1, 2)]])
    res = F.run_instance(F.OracleBackend(), spec)
    F.verify_both(spec, res)
    proof = res["proof"]
    rng = np.random.default_rng(5)
    # any flipped bit of the proof (caps, leaves, siblings, layer values, final polynomial, witness, length prefixes)
    for _ in range(60):
        bad = bytearray(proof)
        pos = int(rng.integers(0, len(bad)))
        bad[pos] ^= 1 << int(rng.integers(0, 8))
        _reject(spec, res, proof=bytes(bad))
    _reject(spec, res, proof=proof[:-8])
    _reject(spec, res, proof=proof + b"\0" * 8)
    # a wrong claimed opening, a wrong cap, a wrong point, a different polynomial list
    op = [o.copy() for o in res["opened"]]
    op[1][0, 0] = (int(op[1][0, 0]) + 1) % O.P
    _reject(spec, res, opened=op)
    caps = [c.copy() for c in res["caps"]]
    caps[2][0, 0] = (int(caps[2][0, 0]) + 1) % O.P
    _reject(spec, res, caps=caps)
    b2 = list(res["batches"])
    b2[0] = (((b2[0][0][0] + 1) % O.P, b2[0][0][1]), b2[0][1])
    _reject(spec, res, batches=b2)
    b3 = list(res["batches"])
    b3[1] = (b3[1][0], [(1, 3, 3)])
    _reject(spec, res, batches=b3)
    # blinding flag of an oracle: the leaf length no longer matches
    infos = list(res["infos"])
    infos[1] = (7, False)
    _reject(spec, res, infos=infos)


def test_malformed_arguments_are_statuses():
    pc = cityprover.fri_params(6, 1, 2, 0, 3, (7,))   # arity out of range
    with pytest.raises(cityprover.CityProverError, match="arity"):
        cityprover.fri_verify(pc, [(1, 0)], [np.zeros((4, 4), np.uint64)], [((3, 0), [(0, 0, 1)])], [np.zeros((1, 2), np.uint64)],
                              cityprover.ChallengerState(), b"")
    pc = cityprover.fri_params(6, 1, 2, 0, 3, ())
    with pytest.raises(cityprover.CityProverError, match="out of range"):
        cityprover.fri_verify(pc, [(1, 0)], [np.zeros((4, 4), np.uint64)], [((3, 0), [(0, 1, 1)])], [np.zeros((1, 2), np.uint64)],
                              cityprover.ChallengerState(), b"")
    with pytest.raises(cityprover.CityProverError, match="malformed|truncated"):
        cityprover.fri_verify(pc, [(1, 0)], [np.zeros((4, 4), np.uint64)], [((3, 0), [(0, 0, 1)])], [np.zeros((1, 2), np.uint64)],
                              cityprover.ChallengerState(), b"\1\2\3")


def test_toy_air_on_the_oracle_verifies_and_a_wrong_trace_does_not():
    pr = F.toy_stark_prove(F.OracleBackend())
    assert F.toy_stark_verify(pr, use_product=True) is None
    assert F.toy_stark_verify(pr, use_product=False) is None
    # a proof for a trace that violates a transition constraint: the quotient is no polynomial of the claimed degree, so either the
    # constraint check at zeta or the low-degree test must fail
    orig = F.toy_trace

    def broken(n):
        t = orig(n)
        t[2, n // 2] = (int(t[2, n // 2]) + 1) % O.P
        return t
    F.toy_trace = broken
    try:
        bad = F.toy_stark_prove(F.OracleBackend())
    finally:
        F.toy_trace = orig
    assert F.toy_stark_verify(bad, use_product=True) is not None
    assert F.toy_stark_verify(bad, use_product=False) is not None
