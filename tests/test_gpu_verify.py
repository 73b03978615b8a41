"""cp_verify (host-side verifier in the product library) accepts the GPU prover's proofs and the oracle's,
and rejects every kind of tampering the oracle's verifier rejects, naming the failing check."""
import numpy as np
import pytest

import oracle_lib as O
from proof_format import parse_proof, serialize_proof
from synth_circuit import build

pytestmark = pytest.mark.gpu
P = O.P


@pytest.fixture(scope="module")
def prover():
    import cityprover
    p = cityprover.Prover(0)
    yield p
    p.close()


def load(prover, c, digest):
    import cityprover as cp
    s = c["shape"]
    sh = cp.standard_recursion_shape(degree_bits=s.degree_bits, num_constants=s.num_constants,
                                     num_routed_wires=s.num_routed_wires, num_wires=s.num_wires,
                                     num_challenges=s.num_challenges, num_partial_products=s.num_partial_products,
                                     quotient_degree_factor=s.quotient_degree_factor, rate_bits=s.rate_bits,
                                     cap_height=s.cap_height, pow_bits=s.pow_bits, num_query_rounds=s.num_query_rounds,
                                     arity_bits=tuple(s.arity_bits[i] for i in range(s.n_arity)))
    circ = cp.Circuit(prover, sh, digest, c["cs_values"])
    cp.set_gates(circ, c["gate_list"], 1)
    return circ


@pytest.mark.parametrize("db,R,W,arity", [(5, 16, 20, (2,)), (7, 24, 30, (2, 2))])
def test_verify_accepts_and_rejects(prover, db, R, W, arity):
    import cityprover as cp
    c = build(db=db, num_routed=R, num_wires=W, chunk=8, rate_bits=3, arity_bits=arity, seed=db + 40)
    digest = [4, 3, 2, 1]
    circ = load(prover, c, digest)
    dw = prover.to_device(c["wires"][None])
    proof = cp.prove_batch_dev(prover, [circ], [c["public_inputs"]], dw.ptr)[0]
    cp.verify(circ, proof)                                    # accepted
    oracle_proof, _ = O.prove_full(c["shape"], c["gates"], digest, c["public_inputs"], c["cs_values"], c["wires"])
    cp.verify(circ, oracle_proof)
    muts = {
        "public input": lambda d: d["public_inputs"].__setitem__(0, (d["public_inputs"][0] + 1) % P),
        "opening": lambda d: d["openings"]["wires"][2].__setitem__(0, (d["openings"]["wires"][2][0] + 1) % P),
        "quotient opening": lambda d: d["openings"]["quotient_polys"][1].__setitem__(1, 7),
        "final poly": lambda d: d["final_poly"][0].__setitem__(0, (d["final_poly"][0][0] + 1) % P),
        "pow": lambda d: d.__setitem__("pow_witness", d["pow_witness"] + 1),
        "leaf value": lambda d: d["queries"][1]["initial"][1][0].__setitem__(3, 9),
        "sibling": lambda d: d["queries"][0]["initial"][2][1][0].__setitem__(0, 1),
        "fri step": lambda d: d["queries"][2]["steps"][0][0][1].__setitem__(0, 3),
        "cap": lambda d: d["wires_cap"][0].__setitem__(0, 5),
    }
    for name, m in muts.items():
        d = parse_proof(proof)
        m(d)
        bad = serialize_proof(d)
        with pytest.raises(cp.CityProverError):
            cp.verify(circ, bad)
        assert O.verify_full(c["shape"], c["gates"], digest, circ.cs_cap(), bad) != 0, name
    with pytest.raises(cp.CityProverError):
        cp.verify(circ, proof[:-8])                            # truncated
    with pytest.raises(cp.CityProverError):
        cp.verify(circ, proof + b"\0" * 8)                     # trailing bytes
    # a witness violating a gate proves, but does not verify; the message names the identity
    w = c["wires"].copy()
    row = c["gate_of_row"].index(3)
    w[3, row] = (int(w[3, row]) + 1) % P
    dw2 = prover.to_device(w[None])
    bad = cp.prove_batch_dev(prover, [circ], [c["public_inputs"]], dw2.ptr)[0]
    with pytest.raises(cp.CityProverError, match="vanishing identity"):
        cp.verify(circ, bad)
    # a proof for another circuit digest does not verify here
    other = load(prover, c, [9, 9, 9, 9])
    with pytest.raises(cp.CityProverError):
        cp.verify(other, proof)
    for b in (dw, dw2):
        b.free()
    circ.close(); other.close()


def test_verify_survives_malformed_bytes(prover):
    """cp_verify on mutated proofs — flipped bits, truncations, overwritten 8-byte words (length prefixes among them:
    huge and near-2^64 values), trailing bytes: every one is refused with an error code, none crashes or hangs (the ABI
    never aborts, include/cityprover.h). The oracle's parser sees the same mutations under ASan in DESIGN.md §3."""
    import cityprover as cp
    c = build(db=5, num_routed=16, num_wires=20, chunk=8, rate_bits=3, arity_bits=(2,), seed=3)
    circ = load(prover, c, [1, 2, 3, 4])
    proof = cp.prove(circ, c["wires"], c["public_inputs"])
    cp.verify(circ, proof)
    rng = np.random.default_rng(1)
    n = len(proof)
    for it in range(1500):
        b = bytearray(proof)
        kind = it % 5
        if kind == 0:
            for _ in range(int(rng.integers(1, 4))):
                b[int(rng.integers(0, n))] ^= 1 << int(rng.integers(0, 8))
        elif kind == 1:
            b = b[: int(rng.integers(0, n))]
        elif kind == 2:
            pos = int(rng.integers(0, n - 8)) & ~7
            b[pos:pos + 8] = int(rng.integers(0, 2**63)).to_bytes(8, "little")
        elif kind == 3:
            b += bytes(rng.integers(0, 256, int(rng.integers(1, 64)), dtype=np.uint8))
        else:
            pos = int(rng.integers(0, n - 8)) & ~7
            b[pos:pos + 8] = (2**64 - 1 - int(rng.integers(0, 4))).to_bytes(8, "little")
        with pytest.raises(cp.CityProverError):
            cp.verify(circ, bytes(b))
    cp.verify(circ, proof)        # the context is still usable
    circ.close()
