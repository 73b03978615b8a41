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
                                     arity_bits=tuple(s.arity_bits[i] for i in range(s.n_arity)),
                                     num_public_inputs=len(c["public_inputs"]))
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


def test_verify_rejects_noncanonical_elements_and_extended_public_inputs(prover):
    """(1) every field element of a proof must be < p — v + p in an opening acts as v where it is multiplied
    and as another value where it is added, which would break the binding between the FRI-checked evaluation and the
    constraint check; (2) the number of public inputs is the circuit's: hash_no_pad is an unpadded sponge, so
    [a.., 0] has the hash of [a..] whenever the appended element equals the state word it overwrites."""
    import cityprover as cp
    c = build(db=5, num_routed=16, num_wires=20, chunk=8, rate_bits=3, arity_bits=(2,), seed=11)
    circ = load(prover, c, [1, 2, 3, 4])
    proof = cp.prove(circ, c["wires"], c["public_inputs"])
    cp.verify(circ, proof)
    # a non-canonical encoding only exists for values below 2^32 - 1, which a real proof hardly ever holds: plant
    # one (P + 3 encodes 3) in every kind of element and require the canonicity check itself to fire
    plant = {
        "opening": lambda d: d["openings"]["wires"][1].__setitem__(1, P + 3),
        "cap": lambda d: d["quotient_cap"][0].__setitem__(2, P),
        "fri cap": lambda d: d["commit_caps"][0][0].__setitem__(0, 2**64 - 1),
        "leaf": lambda d: d["queries"][0]["initial"][1][0].__setitem__(0, P + 1),
        "sibling": lambda d: d["queries"][1]["initial"][0][1][0].__setitem__(3, P + 9),
        "fri eval": lambda d: d["queries"][0]["steps"][0][0][0].__setitem__(1, P + 2),
        "final poly": lambda d: d["final_poly"][0].__setitem__(0, P + 7),
        "public input": lambda d: d["public_inputs"].__setitem__(0, P + 1),
    }
    for name, m in plant.items():
        d = parse_proof(proof)
        m(d)
        with pytest.raises(cp.CityProverError, match="not canonical"):
            cp.verify(circ, serialize_proof(d))
    # zero-extended public inputs: same pi_hash when the sixth element equals the state word it would overwrite
    d = parse_proof(proof)
    st = np.zeros(12, np.uint64)
    st[:5] = np.asarray(c["public_inputs"], np.uint64)
    st = O.permute_many(st.reshape(1, 12)).reshape(12)
    d["public_inputs"] = list(d["public_inputs"]) + [int(st[5])]
    with pytest.raises(cp.CityProverError, match="public inputs"):
        cp.verify(circ, serialize_proof(d))
    # and the prover refuses a public-input vector of another length
    with pytest.raises(cp.CityProverError, match="public inputs"):
        cp.prove(circ, c["wires"], list(c["public_inputs"]) + [0])
    cp.verify(circ, proof)
    circ.close()


def test_power_on_self_test_guards_context_creation():
    """cp_ctx_create runs the device arithmetic self-test (field products through every carry path + one permutation against
    the host's portable code). A device that answers wrongly — simulated by cp_fault_inject — gets no context."""
    import cityprover as cp
    lib = cp.load_library()
    assert lib.cp_fault_inject(2, 0) == 0
    try:
        with pytest.raises(cp.CityProverError, match="self-test failed"):
            cp.Prover(0)
    finally:
        lib.cp_fault_inject(2, -1)
    p = cp.Prover(0)          # and the next context is fine
    assert (p.field_mul([3], [5]) == [15]).all()
    p.close()


def test_fault_injection_threads_and_allocations(prover):
    """include/cityprover.h: "NEVER abort or throw". A worker thread that cannot be created is done without (same
    bytes); a std::bad_alloc inside a proving / verifying call comes back as CP_ERR_OOM and the context stays usable."""
    import cityprover as cp
    lib = cp.load_library()
    cases = [build(db=5, num_routed=16, num_wires=20, chunk=8, rate_bits=3, arity_bits=(2,), seed=20 + i) for i in range(2)]
    circs = [load(prover, c, [i, 2, 3, 4]) for i, c in enumerate(cases)]
    pick = [i % 2 for i in range(16)]
    args = ([circs[i] for i in pick], [cases[i]["public_inputs"] for i in pick], [cases[i]["wires"] for i in pick])
    want = cp.prove_batch(prover, *args)
    try:
        for after in (0, 1, 2, 5):          # transcript helpers (host_for) at different points of the call
            assert lib.cp_fault_inject(0, after) == 0
            assert cp.prove_batch(prover, *args) == want
        prover.set_lanes(3)                 # lane threads: a share that finds no thread runs on the caller
        for after in (0, 1):
            assert lib.cp_fault_inject(0, after) == 0
            assert cp.prove_batch(prover, *args) == want
        prover.set_lanes(1)
        lib.cp_fault_inject(0, -1)
        for after in (0, 1, 3, 6):          # one allocation checkpoint per phase of the call
            assert lib.cp_fault_inject(1, after) == 0
            with pytest.raises(cp.CityProverError, match=r"\[-4\]"):
                cp.prove_batch(prover, *args)
            assert cp.prove_batch(prover, *args) == want      # arena rewound, context usable
        prover.set_lanes(2)
        assert lib.cp_fault_inject(1, 2) == 0
        with pytest.raises(cp.CityProverError, match=r"\[-4\]"):
            cp.prove_batch(prover, *args)
        assert cp.prove_batch(prover, *args) == want
        prover.set_lanes(1)
        assert lib.cp_fault_inject(1, 0) == 0
        with pytest.raises(cp.CityProverError, match=r"\[-4\]"):
            cp.verify(circs[0], want[0])
        cp.verify(circs[0], want[0])
        assert lib.cp_fault_inject(7, 0) != 0
    finally:
        lib.cp_fault_inject(0, -1)
        lib.cp_fault_inject(1, -1)
        prover.set_lanes(1)
    for c in circs:
        c.close()


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
