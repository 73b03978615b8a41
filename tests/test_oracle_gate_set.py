"""A8 gate-set coverage on the CPU oracle: circuits that contain every gate of the city-common set
(city_common_circuit/src/builder/pad_circuit.rs:31-55) and every in-tree u32 gate, laid out with plonky2's selector
grouping; satisfying witnesses must verify, a corrupted first / last wire of each gate must be rejected by the
vanishing identity (which is evaluated over F_p^2 at zeta: an independent path from the prover's base-field one)."""
import ctypes

import numpy as np
import pytest

import oracle_lib as O
import synth_gates as SG

P = O.P


def cs_cap(c):
    sh = c["shape"]
    k, n = c["cs_values"].shape
    cap = np.zeros((1 << sh.cap_height, 4), np.uint64)
    O.lib().or_commit_batch(O.ptr(c["cs_values"]), k, sh.degree_bits, sh.rate_bits, sh.cap_height, None, None, None, O.ptr(cap))
    return cap


def test_selector_grouping_matches_plonky2_rule():
    gates = sorted(SG.CITY_COMMON, key=lambda g: (SG.gate_degree(g), SG._ID[g[0]]))
    assert SG.selector_groups(gates, 9) == [(0, 7), (7, 11), (11, 13), (13, 14)]
    # the recursion gate set without ComparisonGate needs 3 selectors + 2 constants = the 5 "constants" openings of the
    # reference proofs in qbench_data/example.bin
    rec = [g for g in gates if g[0] != SG.COMPARISON]
    assert len(SG.selector_groups(rec, 9)) == 3
    allg = sorted(SG.ALL_GATES, key=lambda g: (SG.gate_degree(g), SG._ID[g[0]]))
    assert SG.selector_groups(allg, 9) == [(0, 7), (7, 13), (13, 18), (18, 21), (21, 22)]


def test_constraint_counts():
    want = {SG.COMPARISON: 88, SG.RANDOM_ACCESS: 26, SG.POSEIDON: 123, SG.POSEIDON_MDS: 24, SG.REDUCING: 86,
            SG.REDUCING_EXT: 64, SG.ARITHMETIC: 20, SG.ARITHMETIC_EXT: 20, SG.MUL_EXT: 26, SG.BASE_SUM: 64,
            SG.COSET_INTERPOLATION: 12, SG.U32_ARITHMETIC: 108, SG.U32_RANGE_CHECK: 119, SG.U32_ADD_MANY: 105,
            SG.U32_SUBTRACTION: 114, SG.U32_INTERLEAVE: 102, SG.UNINTERLEAVE_TO_U32: 134, SG.UNINTERLEAVE_TO_B32: 134,
            SG.EXPONENTIATION: 67}
    for g in SG.ALL_GATES:
        if g[0] in want:
            og = O.make_gates([(g[0], 0, 0, 1, g[1], g[2], g[3])], 1, [1])
            assert O.lib().or_gates_num_constraints(ctypes.byref(og)) == want[g[0]], g


@pytest.mark.parametrize("name,gate_set,db", [("city_common", SG.CITY_COMMON, 6), ("all", SG.ALL_GATES, 6)])
def test_gate_set_verifies_and_rejects(name, gate_set, db):
    c = SG.build_gate_set(gate_set, db=db, seed=5 + db, arity_bits=(2,))
    assert c["copies"], "no copy constraints were placed"
    digest = [4, 2, 4, 2]
    O.lib().or_set_threads(8)
    try:
        proof, _ = O.prove_full(c["shape"], c["gates"], digest, c["public_inputs"], c["cs_values"], c["wires"])
        cap = cs_cap(c)
        assert O.verify_full(c["shape"], c["gates"], digest, cap, proof) == 0
        params = {g[0]: g for g in c["sorted_gates"]}
        for t in sorted(set(c["row_types"])):
            nw = SG.gate_num_wires(params[t])
            if nw == 0:
                continue
            row = c["row_types"].index(t)
            for wire in {0, nw - 1}:
                w = c["wires"].copy()
                w[wire, row] = (int(w[wire, row]) + 1) % P
                bad, _ = O.prove_full(c["shape"], c["gates"], digest, c["public_inputs"], c["cs_values"], w)
                assert O.verify_full(c["shape"], c["gates"], digest, cap, bad) <= -1000, (SG._ID[t], wire)
        # a broken copy constraint (gates still satisfied) is caught by the permutation argument
        (ja, ra), _ = c["copies"][0]
        w = c["wires"].copy()
        w[ja, ra] = (int(w[ja, ra]) + 1) % P
        bad, _ = O.prove_full(c["shape"], c["gates"], digest, c["public_inputs"], c["cs_values"], w)
        assert O.verify_full(c["shape"], c["gates"], digest, cap, bad) <= -1000
    finally:
        O.lib().or_set_threads(1)
