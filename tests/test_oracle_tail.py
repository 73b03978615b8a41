"""Oracle prover tail (transcript, openings, FRI, proof bytes) — self-consistency: the restated
verifier accepts the restated prover's bytes, rejects tampering, and the bytes parse with the bincode
layout recovered from the reference proofs."""
import numpy as np
import pytest

import oracle_lib as O
from proof_format import parse_proof, serialize_proof

P = O.P


def small_case(db=6, seed=3, **kw):
    sh = O.standard_shape(degree_bits=db, num_wires=12, num_routed=8, num_constants=3, num_challenges=2,
                          num_partial_products=2, quotient_degree_factor=4, rate_bits=3, cap_height=2,
                          pow_bits=6, num_query_rounds=5, arity_bits=(2, 2), **kw)
    n = 1 << db
    f = lambda k, s: O.splitmix64_felts(seed * 1000 + s, k * n).reshape(k, n)
    cs, w, z, q = f(11, 1), f(12, 2), f(6, 3), f(8, 4)
    return sh, cs, w, z, q


def test_prove_then_verify_small():
    sh, cs, w, z, q = small_case()
    digest = [1, 2, 3, 4]
    proof, dbg = O.prove_tail(sh, digest, [7, 8, 9], cs, w, z, q)
    cs_cap = O.commit_batch(cs, 3, 2, want=("cap",))["cap"]
    rc, vdbg = O.verify_tail(sh, digest, cs_cap, proof)
    assert rc == 0
    assert list(vdbg.zeta) == list(dbg.zeta)
    assert [list(b) for b in vdbg.fri_betas][:2] == [list(b) for b in dbg.fri_betas][:2]
    assert list(vdbg.query_indices)[:5] == list(dbg.query_indices)[:5]
    assert dbg.pow_response >> (64 - 6) == 0
    # deterministic: same inputs, same bytes
    proof2, _ = O.prove_tail(sh, digest, [7, 8, 9], cs, w, z, q)
    assert proof2 == proof
    # tampering is rejected (flip one public input, one opening, one final-poly coefficient)
    p = parse_proof(proof)
    assert serialize_proof(p) == proof
    for mutate in (lambda d: d["public_inputs"].__setitem__(0, 8),
                   lambda d: d["openings"]["wires"][3].__setitem__(0, (d["openings"]["wires"][3][0] + 1) % P),
                   lambda d: d["final_poly"][0].__setitem__(1, (d["final_poly"][0][1] + 1) % P),
                   lambda d: d["queries"][2]["initial"][1][0].__setitem__(4, 5)):
        d = parse_proof(proof)
        mutate(d)
        rc, _ = O.verify_tail(sh, digest, cs_cap, serialize_proof(d))
        assert rc != 0
    # wrong circuit digest changes every challenge
    rc, _ = O.verify_tail(sh, [1, 2, 3, 5], cs_cap, proof)
    assert rc != 0


def test_pow_override_is_injected_verbatim():
    sh, cs, w, z, q = small_case(seed=5)
    proof, dbg = O.prove_tail(sh, [0, 0, 0, 1], [], cs, w, z, q)
    nonce = parse_proof(proof)["pow_witness"]
    proof2, _ = O.prove_tail(sh, [0, 0, 0, 1], [], cs, w, z, q, pow_override=nonce)
    assert proof2 == proof
    # smallest-nonce rule: no smaller witness satisfies the PoW
    assert nonce < (1 << 12)


@pytest.mark.parametrize("db", [4, 7])
def test_other_sizes(db):
    sh, cs, w, z, q = small_case(db=db, seed=db)
    proof, _ = O.prove_tail(sh, [9, 9, 9, 9], [1], cs, w, z, q)
    cs_cap = O.commit_batch(cs, 3, 2, want=("cap",))["cap"]
    assert O.verify_tail(sh, [9, 9, 9, 9], cs_cap, proof)[0] == 0


def test_product_shape_proof_has_reference_layout(golden_dir):
    """standard_recursion_config shape: the bytes must have exactly the reference proof's length/shape."""
    sh = O.standard_shape()
    n = 1 << 12
    f = lambda k, s: O.splitmix64_felts(77 + s, k * n).reshape(k, n)
    O.lib().or_set_threads(8)
    proof, _ = O.prove_tail(sh, [5, 6, 7, 8], list(range(8)), f(85, 1), f(135, 2), f(20, 3), f(16, 4))
    cs_cap = O.commit_batch(f(85, 1), 3, 4, want=("cap",))["cap"]
    O.lib().or_set_threads(1)
    assert len(proof) == 130360  # == the reference WrappedSignatureProof size (8 public inputs)
    assert O.verify_tail(sh, [5, 6, 7, 8], cs_cap, proof)[0] == 0
    p = parse_proof(proof)
    assert [len(e[0]) for e in p["queries"][0]["initial"]] == [85, 135, 20, 16]
    assert [(len(e[0]), len(e[1])) for e in p["queries"][0]["steps"]] == [(16, 7), (16, 3)]
