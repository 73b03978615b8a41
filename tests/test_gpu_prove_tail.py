"""GPU parity for the proof tail: cp_prove_tail must produce byte-identical ProofWithPublicInputs to
the CPU oracle on the same polynomials (same smallest-nonce PoW rule), and the oracle's verifier side
must accept the GPU bytes."""
import numpy as np
import pytest

import oracle_lib as O
from proof_format import parse_proof

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def prover():
    import cityprover
    p = cityprover.Prover(0)
    yield p
    p.close()


def polys(shape, seed):
    n = 1 << shape.degree_bits
    k_cs = shape.num_constants + shape.num_routed_wires
    k_z = shape.num_challenges * (1 + shape.num_partial_products)
    k_q = shape.num_challenges * shape.quotient_degree_factor
    f = lambda k, s: O.splitmix64_felts(seed * 977 + s, k * n).reshape(k, n)
    return f(k_cs, 1), f(shape.num_wires, 2), f(k_z, 3), f(k_q, 4)


def oracle_shape(s):
    return O.standard_shape(degree_bits=s.degree_bits, num_wires=s.num_wires, num_routed=s.num_routed_wires,
                            num_constants=s.num_constants, num_challenges=s.num_challenges,
                            num_partial_products=s.num_partial_products,
                            quotient_degree_factor=s.quotient_degree_factor, rate_bits=s.rate_bits,
                            cap_height=s.cap_height, pow_bits=s.pow_bits, num_query_rounds=s.num_query_rounds,
                            arity_bits=tuple(s.arity_bits[i] for i in range(s.n_arity)))


SMALL = dict(num_constants=3, num_routed_wires=8, num_wires=12, num_challenges=2, num_partial_products=2,
             quotient_degree_factor=4, rate_bits=3, cap_height=2, pow_bits=6, num_query_rounds=5)


@pytest.mark.parametrize("db,arity,extra", [(6, (2, 2), {}), (8, (4,), {}), (9, (3, 2), {"cap_height": 3}),
                                            (5, (1, 1, 1), {"cap_height": 0, "pow_bits": 3}),
                                            (10, (4, 4), {"num_challenges": 3, "num_partial_products": 1})])
def test_small_shapes_byte_identical(prover, db, arity, extra):
    import cityprover as cp
    sh = cp.standard_recursion_shape(degree_bits=db, arity_bits=arity, num_public_inputs=3, **{**SMALL, **extra})
    cs, w, z, q = polys(sh, db)
    digest, pis = [11, 22, 33, 44], [5, 6, 7]
    circ = cp.Circuit(prover, sh, digest, cs)
    got = circ.prove_tail(pis, w, z, q)
    want, dbg = O.prove_tail(oracle_shape(sh), digest, pis, cs, w, z, q)
    assert parse_proof(got) == parse_proof(want)
    assert got == want
    rc, _ = O.verify_tail(oracle_shape(sh), digest, circ.cs_cap(), got)
    assert rc == 0
    # nonce injection: reproduce with the witness given
    nonce = parse_proof(want)["pow_witness"]
    assert circ.prove_tail(pis, w, z, q, pow_override=nonce) == want
    # no public inputs at all
    with pytest.raises(cp.CityProverError, match="public inputs"):   # the count is the circuit's
        circ.prove_tail([], w, z, q)
    sh0 = cp.standard_recursion_shape(degree_bits=db, arity_bits=arity, num_public_inputs=0, **{**SMALL, **extra})
    circ0 = cp.Circuit(prover, sh0, digest, cs)
    assert circ0.prove_tail([], w, z, q) == O.prove_tail(oracle_shape(sh), digest, [], cs, w, z, q)[0]
    circ0.close()
    circ.close()


def test_product_shape_byte_identical(prover):
    """standard_recursion_config: n = 2^12, 135 wires, 28 queries, 16-bit PoW — 130 360-byte proof."""
    import cityprover as cp
    sh = cp.standard_recursion_shape(num_public_inputs=8)
    cs, w, z, q = polys(sh, 42)
    digest, pis = [1, 2, 3, 4], list(range(100, 108))
    circ = cp.Circuit(prover, sh, digest, cs)
    got = circ.prove_tail(pis, w, z, q)
    O.lib().or_set_threads(8)
    want, dbg = O.prove_tail(oracle_shape(sh), digest, pis, cs, w, z, q)
    O.lib().or_set_threads(1)
    assert len(got) == 130360
    assert got == want
    assert O.verify_tail(oracle_shape(sh), digest, circ.cs_cap(), got)[0] == 0
    # determinism: two runs, same bytes
    assert circ.prove_tail(pis, w, z, q) == got
    circ.close()


def test_error_paths(prover):
    import cityprover as cp
    sh = cp.standard_recursion_shape(degree_bits=6, arity_bits=(2, 2), num_public_inputs=1, **SMALL)
    cs, w, z, q = polys(sh, 1)
    with pytest.raises(cp.CityProverError):
        cp.Circuit(prover, cp.standard_recursion_shape(degree_bits=6, arity_bits=(5, 5, 5), **SMALL), [0] * 4, cs)
    circ = cp.Circuit(prover, sh, [0, 0, 0, 0], cs)
    with pytest.raises(cp.CityProverError):
        circ.prove_tail([O.P], w, z, q)  # non-canonical public input
    assert len(circ.prove_tail([1], w, z, q)) > 0  # still usable
    circ.close()


def test_batched_proofs_of_different_circuits(prover):
    """Batch of 5 proofs over 2 circuits of the same shape: each must equal its single-proof oracle bytes."""
    import cityprover as cp
    sh = cp.standard_recursion_shape(degree_bits=7, arity_bits=(2, 2), **SMALL)
    osh = oracle_shape(sh)
    csA, _, _, _ = polys(sh, 100)
    csB, _, _, _ = polys(sh, 200)
    # proof i carries i public inputs: circuits that differ only in num_public_inputs share a batch
    def shape_with(npi):
        return cp.standard_recursion_shape(degree_bits=7, arity_bits=(2, 2), num_public_inputs=npi, **SMALL)
    is_a = [True, False, True, False, False]
    circs = [cp.Circuit(prover, shape_with(i), [1, 1, 1, 1] if is_a[i] else [2, 2, 2, 2], csA if is_a[i] else csB) for i in range(5)]
    ws, zs, qs, pis = [], [], [], []
    for i in range(5):
        _, w, z, q = polys(sh, 300 + i)
        ws.append(w); zs.append(z); qs.append(q); pis.append(list(range(i)))
    dw, dz, dq = prover.to_device(np.stack(ws)), prover.to_device(np.stack(zs)), prover.to_device(np.stack(qs))
    got = cp.prove_tail_batch_dev(prover, circs, pis, dw.ptr, dz.ptr, dq.ptr)
    for i in range(5):
        cs, dg = (csA, [1, 1, 1, 1]) if is_a[i] else (csB, [2, 2, 2, 2])
        want, _ = O.prove_tail(osh, dg, pis[i], cs, ws[i], zs[i], qs[i])
        assert got[i] == want, f"proof {i}"
    # nonce injection for a subset of the batch
    n2 = parse_proof(got[2])["pow_witness"]
    got2 = cp.prove_tail_batch_dev(prover, circs, pis, dw.ptr, dz.ptr, dq.ptr, pow_overrides=[None, None, n2, None, None])
    assert got2 == got
    with pytest.raises(cp.CityProverError):  # the number of public inputs is the circuit's
        cp.prove_tail_batch_dev(prover, circs, [p + [0] for p in pis], dw.ptr, dz.ptr, dq.ptr)
    for b in (dw, dz, dq):
        b.free()
    for c in circs:
        c.close()


def test_batch_rejects_mixed_shapes(prover):
    import cityprover as cp
    sh1 = cp.standard_recursion_shape(degree_bits=6, arity_bits=(2, 2), **SMALL)
    sh2 = cp.standard_recursion_shape(degree_bits=6, arity_bits=(2,), **SMALL)
    cs, w, z, q = polys(sh1, 1)
    c1, c2 = cp.Circuit(prover, sh1, [0] * 4, cs), cp.Circuit(prover, sh2, [0] * 4, cs)
    dw, dz, dq = prover.to_device(np.stack([w, w])), prover.to_device(np.stack([z, z])), prover.to_device(np.stack([q, q]))
    with pytest.raises(cp.CityProverError):
        cp.prove_tail_batch_dev(prover, [c1, c2], [[], []], dw.ptr, dz.ptr, dq.ptr)
    for b in (dw, dz, dq):
        b.free()
    c1.close(); c2.close()
