"""Oracle end-to-end: wires -> proof (A7 + A8 + tail) on a synthetic satisfiable circuit; the verifier side
(FRI + the vanishing identity at zeta) accepts it, and rejects a witness that breaks a gate or a copy
constraint. This is the self-consistency gate for the quotient restatement (parity unpinned by reference data)."""
import ctypes
import numpy as np
import pytest

import oracle_lib as O
from proof_format import parse_proof
from synth_circuit import build

P = O.P


def cs_cap(c):
    return O.commit_batch(c["cs_values"], c["shape"].rate_bits, c["shape"].cap_height, want=("cap",))["cap"]


@pytest.mark.parametrize("db,R,W,rb,arity", [(5, 16, 20, 3, (2,)), (6, 24, 30, 3, (2, 2)), (7, 8, 10, 3, (3,))])
def test_full_proof_verifies(db, R, W, rb, arity):
    c = build(db=db, num_routed=R, num_wires=W, chunk=1 << rb, rate_bits=rb, arity_bits=arity, seed=db)
    digest = [9, 8, 7, 6]
    proof, dbg = O.prove_full(c["shape"], c["gates"], digest, c["public_inputs"], c["cs_values"], c["wires"])
    assert O.verify_full(c["shape"], c["gates"], digest, cs_cap(c), proof) == 0
    p = parse_proof(proof)
    assert p["public_inputs"] == c["public_inputs"]
    # Z(1) = 1 shows up as L_0 term; the quotient chunks are genuine polynomials: re-proving is deterministic
    assert O.prove_full(c["shape"], c["gates"], digest, c["public_inputs"], c["cs_values"], c["wires"])[0] == proof


def test_broken_witness_is_rejected():
    c = build(db=5, num_routed=16, num_wires=20, chunk=8, rate_bits=3, seed=11)
    digest = [1, 1, 1, 1]
    cap = cs_cap(c)
    row = c["gate_of_row"].index(3)
    w = c["wires"].copy()
    w[3, row] = (int(w[3, row]) + 1) % P                 # arithmetic output wrong
    bad, _ = O.prove_full(c["shape"], c["gates"], digest, c["public_inputs"], c["cs_values"], w)
    assert O.verify_full(c["shape"], c["gates"], digest, cap, bad) <= -1000   # FRI fine, vanishing identity fails
    w = c["wires"].copy()
    w[0, 0] = (int(w[0, 0]) + 1) % P                     # public-input gate: wire != pi_hash
    bad, _ = O.prove_full(c["shape"], c["gates"], digest, c["public_inputs"], c["cs_values"], w)
    assert O.verify_full(c["shape"], c["gates"], digest, cap, bad) <= -1000
    # wrong public inputs at verification time
    good, _ = O.prove_full(c["shape"], c["gates"], digest, c["public_inputs"], c["cs_values"], c["wires"])
    d = parse_proof(good)
    from proof_format import serialize_proof
    d["public_inputs"][0] = (d["public_inputs"][0] + 1) % P
    assert O.verify_full(c["shape"], c["gates"], digest, cap, serialize_proof(d)) != 0


def test_copy_constraint_violation_is_rejected():
    c = build(db=5, num_routed=16, num_wires=20, chunk=8, rate_bits=3, seed=3, n_copies=10)
    ident_like = c["cs_values"][3:]
    # find a cell that sigma maps elsewhere, break the equality, and re-derive that row's outputs so that
    # only the permutation argument is violated
    omega = pow(7, (P - 1) >> 5, P)
    k_is = c["k_is"]
    moved = [(j, i) for j in range(16) for i in range(32) if int(ident_like[j, i]) != k_is[j] * pow(omega, i, P) % P]
    assert moved
    j, i = moved[0]
    w = c["wires"].copy()
    w[j, i] = (int(w[j, i]) + 5) % P
    op = j // 4
    m0, m1, ad = (int(w[4 * op + t, i]) for t in range(3))
    w[4 * op + 3, i] = (m0 * m1 % P * int(c["cs_values"][1, i]) + ad * int(c["cs_values"][2, i])) % P
    bad, _ = O.prove_full(c["shape"], c["gates"], [2, 2, 2, 2], c["public_inputs"], c["cs_values"], w)
    assert O.verify_full(c["shape"], c["gates"], [2, 2, 2, 2], cs_cap(c), bad) <= -1000


def test_poseidon_gate_rows_verify():
    """Circuit with PoseidonGate rows (two selector groups): valid witness verifies; corrupting one
    intermediate S-box wire or one output of a Poseidon row is rejected by the vanishing identity."""
    c = build(db=5, num_routed=80, num_wires=135, chunk=8, rate_bits=3, arity_bits=(2,), seed=21,
              poseidon_fraction=0.5)
    assert c["poseidon_gate_index"] in c["gate_of_row"]
    digest = [3, 1, 4, 1]
    proof, _ = O.prove_full(c["shape"], c["gates"], digest, c["public_inputs"], c["cs_values"], c["wires"])
    cap = cs_cap(c)
    assert O.verify_full(c["shape"], c["gates"], digest, cap, proof) == 0
    row = c["gate_of_row"].index(c["poseidon_gate_index"])
    for wire in (70, 15, 24, 100):   # partial-round S-box input, an output, the swap flag, a late full-round input
        w = c["wires"].copy()
        w[wire, row] = (int(w[wire, row]) + 1) % P
        bad, _ = O.prove_full(c["shape"], c["gates"], digest, c["public_inputs"], c["cs_values"], w)
        assert O.verify_full(c["shape"], c["gates"], digest, cap, bad) <= -1000, wire


def test_poseidon_gate_outputs_are_the_permutation():
    from synth_circuit import poseidon_gate_row
    x = [int(v) for v in O.splitmix64_felts(5, 12)]
    row = poseidon_gate_row(x, 0)
    assert row[12:24] == [int(v) for v in O.permute(x)]
    swapped = x[4:8] + x[0:4] + x[8:]
    assert poseidon_gate_row(x, 1)[12:24] == [int(v) for v in O.permute(swapped)]


# wire -> what it is, for the three in-tree u32 gates at the synthetic circuit's parameters
U32_CORRUPTIONS = {
    "cmp": [(0, "first input"), (2, "result bool"), (3, "most significant diff"), (5, "a chunk"), (4 + 32 + 3, "equality dummy"),
            (4 + 48 + 1, "chunk-equal flag"), (4 + 64 + 7, "intermediate value"), (4 + 80 + 1, "msd bit")],
    "arith": [(1, "multiplicand"), (3, "low half"), (4, "high half"), (18 + 40, "a 2-bit limb"), (6 + 5, "inverse")],
    "range": [(2, "input limb"), (7 + 20, "aux limb")],
}


def test_u32_gates_verify_and_reject():
    """In-tree gates (ComparisonGate(32,16), U32ArithmeticGate(3 ops), U32RangeCheckGate(7)) in their own selector
    group: the witness built by the reference's generator logic verifies; every class of wire they constrain is
    caught by the vanishing identity when corrupted."""
    c = build(db=6, num_routed=80, num_wires=135, chunk=8, rate_bits=3, arity_bits=(2,), seed=33, u32_gates=True)
    ids = c["u32_gate_ids"]
    for name, gid in ids.items():
        assert gid in c["gate_of_row"], name
    digest = [9, 9, 9, 1]
    proof, _ = O.prove_full(c["shape"], c["gates"], digest, c["public_inputs"], c["cs_values"], c["wires"])
    cap = cs_cap(c)
    assert O.verify_full(c["shape"], c["gates"], digest, cap, proof) == 0
    for name, gid in ids.items():
        row = c["gate_of_row"].index(gid)
        for wire, what in U32_CORRUPTIONS[name]:
            w = c["wires"].copy()
            if what == "equality dummy":   # only constrained where the two chunks differ
                wire = next(4 + 32 + i for i in range(16) if w[4 + i, row] != w[4 + 16 + i, row])
            w[wire, row] = (int(w[wire, row]) + 1) % P
            bad, _ = O.prove_full(c["shape"], c["gates"], digest, c["public_inputs"], c["cs_values"], w)
            assert O.verify_full(c["shape"], c["gates"], digest, cap, bad) <= -1000, (name, what)


def test_u32_gate_constraint_counts():
    """num_constraints of the in-tree gates (comparison.rs:310, arithmetic_u32.rs:269, range_check_u32.rs:158)."""
    g = O.make_gates([(O.GATE_COMPARISON, 0, 0, 1, 32, 16)], 1, [1])
    assert O.lib().or_gates_num_constraints(ctypes.byref(g)) == 6 + 5 * 16 + 2
    g = O.make_gates([(O.GATE_U32_ARITHMETIC, 0, 0, 1, 3, 0)], 1, [1])
    assert O.lib().or_gates_num_constraints(ctypes.byref(g)) == 108
    g = O.make_gates([(O.GATE_U32_RANGE_CHECK, 0, 0, 1, 8, 0)], 1, [1])
    assert O.lib().or_gates_num_constraints(ctypes.byref(g)) == 136


def test_comparison_witness_semantics():
    from synth_circuit import comparison_row
    for a, b in [(5, 9), (9, 5), (7, 7), (0, 2**32 - 1), (2**32 - 1, 0), (0x12345678, 0x12355678)]:
        r = comparison_row(a, b)
        assert r[2] == int(a <= b)
        assert len(r) == 4 + 5 * 16 + 3


def zk_case(seed=61, db=6):
    """A circuit in zero-knowledge mode: FRI `hiding` salts on the wires / Z / quotient leaves (the blinding ROWS plonky2
    adds at build time are part of the circuit + witness and need nothing from the prover)."""
    c = build(db=db, num_routed=16, num_wires=24, chunk=8, rate_bits=3, arity_bits=(2, 1), seed=seed, cap_height=2)
    c["shape"].zero_knowledge = 1
    N = 1 << (db + 3)
    rng = np.random.default_rng(seed)
    c["salts"] = rng.integers(0, P, (3, O.SALT_SIZE, N), dtype=np.uint64)
    return c


def test_zero_knowledge_salted_leaves():
    """standard_recursion_zk_config path (city_common_circuit/src/circuits/zk_signature/inner.rs:50): the three blinded
    oracles' leaves carry 4 salt elements — present in every query opening, covered by the Merkle paths, ignored by
    fri_combine_initial. Accepts; rejects a touched salt; refuses salts without the flag and the flag without salts."""
    c = zk_case()
    digest = [5, 5, 5, 5]
    proof, _ = O.prove_full_zk(c["shape"], c["gates"], digest, c["public_inputs"], c["cs_values"], c["wires"], c["salts"])
    cap = cs_cap(c)
    assert O.verify_full(c["shape"], c["gates"], digest, cap, proof) == 0
    pr = parse_proof(proof)
    q0 = pr["queries"][0]["initial"]
    assert [len(e) for e, _ in q0] == [16 + 3, 24 + 4, 2 * (1 + 1) + 4, 16 + 4]   # constants+sigmas unsalted
    x = None
    N = c["salts"].shape[2]
    for i in range(N):   # the wires leaf of query 0 ends with that leaf's salt
        if [int(v) for v in c["salts"][0, :, i]] == [int(v) for v in q0[1][0][-4:]]:
            x = i
    assert x is not None
    # different salts -> different caps and proof, same openings
    s2 = c["salts"].copy()
    s2[0, 0, x] = (int(s2[0, 0, x]) + 1) % P
    proof2, _ = O.prove_full_zk(c["shape"], c["gates"], digest, c["public_inputs"], c["cs_values"], c["wires"], s2)
    assert proof2 != proof and O.verify_full(c["shape"], c["gates"], digest, cap, proof2) == 0
    # a salt altered inside the proof breaks that leaf's Merkle path
    bad = bytearray(proof)
    pos = proof.index(np.array(q0[1][0][-4:], dtype=np.uint64).tobytes())
    bad[pos] ^= 1
    assert O.verify_full(c["shape"], c["gates"], digest, cap, bytes(bad)) != 0
    # the verifier must know the circuit is zero-knowledge
    c["shape"].zero_knowledge = 0
    assert O.verify_full(c["shape"], c["gates"], digest, cap, proof) != 0
    with pytest.raises(AssertionError):
        O.prove_full_zk(c["shape"], c["gates"], digest, c["public_inputs"], c["cs_values"], c["wires"], c["salts"])
    c["shape"].zero_knowledge = 1
    with pytest.raises(AssertionError):
        O.prove_full(c["shape"], c["gates"], digest, c["public_inputs"], c["cs_values"], c["wires"])
