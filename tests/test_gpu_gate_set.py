"""GPU A8 gate-set parity: circuits containing every gate of the city-common set (pad_circuit.rs:31-55) and every
in-tree u32 gate, with plonky2's selector grouping. cp_prove_batch bytes == oracle bytes; cp_verify (vanishing identity
over F_p^2 through the same gates.h code, host side) and the oracle verifier accept; corrupted wires are rejected."""
import pytest

import oracle_lib as O
import synth_gates as SG
from test_gpu_prove_full import cp_shape_of

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def prover():
    import cityprover
    p = cityprover.Prover(0)
    yield p
    p.close()


@pytest.mark.parametrize("name,gate_set,db,arity", [("city_common", SG.CITY_COMMON, 6, (2,)), ("all", SG.ALL_GATES, 6, (2,)),
                                                    ("all_2^8", SG.ALL_GATES, 8, (2, 2))])
def test_gate_set_proofs_byte_identical(prover, name, gate_set, db, arity):
    import cityprover as cp
    c = SG.build_gate_set(gate_set, db=db, seed=11 + db, arity_bits=arity)
    sh = cp_shape_of(cp, c["shape"])
    digest = [6, 6, 6, db]
    circ = cp.Circuit(prover, sh, digest, c["cs_values"])
    cp.set_gates(circ, c["gate_list"], c["num_selectors"])
    dw = prover.to_device(c["wires"][None])
    got = cp.prove_batch_dev(prover, [circ], [c["public_inputs"]], dw.ptr)[0]
    O.lib().or_set_threads(8)
    try:
        want, _ = O.prove_full(c["shape"], c["gates"], digest, c["public_inputs"], c["cs_values"], c["wires"])
    finally:
        O.lib().or_set_threads(1)
    assert got == want
    cp.verify(circ, got)
    assert O.verify_full(c["shape"], c["gates"], digest, circ.cs_cap(), got) == 0
    params = {g[0]: g for g in c["sorted_gates"]}
    for t in sorted(set(c["row_types"])):
        nw = SG.gate_num_wires(params[t])
        if nw == 0:
            continue
        row = c["row_types"].index(t)
        w = c["wires"].copy()
        w[nw - 1, row] = (int(w[nw - 1, row]) + 1) % O.P
        dw2 = prover.to_device(w[None])
        bad = cp.prove_batch_dev(prover, [circ], [c["public_inputs"]], dw2.ptr)[0]
        dw2.free()
        with pytest.raises(cp.CityProverError, match="vanishing identity"):
            cp.verify(circ, bad)
    dw.free(); circ.close()


@pytest.mark.parametrize("options", [dict(QUOT_ALL_MAX=0), dict(QUOT_ALL_MAX=0, QUOT_GROUP=0), dict(QUOT_ALL_MAX=0, QUOT_TILE=1),
                                     dict(QUOT_ALL_MAX=0, QUOT_FLIP=0, QUOT_TILE=1, QUOT_GROUP=0), dict(QUOT_ALL_MAX=100)])
def test_every_form_of_the_quotient_gives_the_same_bytes(options):
    """The quotient has four forms, chosen by batch size and switches: every piece a slice of one grid (small batches), a launch per
    gate, the arithmetic family grouped into one launch (round 4, default for batches that fill the chip), and a workgroup per
    64-point tile with the wires staged in LDS once (round 4, off by default). cp_ctx_set_option forces each on a context of its
    own; a circuit with all 22 gate types and one with the city-common set must prove to the oracle's bytes under every one."""
    import cityprover as cp
    p = cp.Prover(0)
    try:
        for name, v in options.items():
            p.set_option(name, v)
        for gate_set, db in ((SG.ALL_GATES, 6), (SG.CITY_COMMON, 7)):
            c = SG.build_gate_set(gate_set, db=db, seed=31 + db, arity_bits=(2,))
            sh = cp_shape_of(cp, c["shape"])
            digest = [5, 5, 5, db]
            circ = cp.Circuit(p, sh, digest, c["cs_values"])
            cp.set_gates(circ, c["gate_list"], c["num_selectors"])
            O.lib().or_set_threads(8)
            try:
                want, _ = O.prove_full(c["shape"], c["gates"], digest, c["public_inputs"], c["cs_values"], c["wires"])
            finally:
                O.lib().or_set_threads(1)
            for B in (1, 3):
                got = cp.prove_batch(p, [circ] * B, [c["public_inputs"]] * B, [c["wires"]] * B)
                assert all(g == want for g in got), (options, B)
            circ.close()
    finally:
        p.close()


def test_set_gates_parameter_validation(prover):
    import cityprover as cp
    c = SG.build_gate_set(SG.CITY_COMMON, db=6, seed=1, arity_bits=(2,))
    sh = cp_shape_of(cp, c["shape"])
    circ = cp.Circuit(prover, sh, [1, 1, 1, 1], c["cs_values"])
    for bad in [(cp.GATE_RANDOM_ACCESS, 0, 0, 1, 5, 1, 0),        # bits > 4
                (cp.GATE_RANDOM_ACCESS, 0, 0, 1, 4, 8, 0),        # 8 copies need 176 wires
                (cp.GATE_COSET_INTERPOLATION, 0, 0, 1, 4, 1, 0),  # degree < 2
                (cp.GATE_COSET_INTERPOLATION, 0, 0, 1, 6, 6, 0),  # 64-point coset
                (cp.GATE_BASE_SUM, 0, 0, 1, 63, 1, 0),            # base 1
                (cp.GATE_REDUCING, 0, 0, 1, 44, 0, 0),            # 136 wires
                (cp.GATE_U32_ADD_MANY, 0, 0, 1, 5, 0, 0),         # no addends
                (cp.GATE_UNINTERLEAVE_TO_U32, 0, 0, 1, 3, 0, 0),  # 201 wires
                (cp.GATE_EXPONENTIATION, 0, 0, 1, 67, 0, 0),      # 136 wires
                (22, 0, 0, 1, 0, 0, 0)]:                          # unknown type
        with pytest.raises(cp.CityProverError):
            cp.set_gates(circ, [bad], 1)
    circ.close()


@pytest.mark.parametrize("db,arity", [(13, (4, 4)), (14, (4, 4, 1))])
def test_larger_circuits(prover, db, arity):
    """Degrees above the 4096-point single-pass NTT tile (op circuits before minification are 2^13..2^15 rows): the
    multi-pass iNTT / LDE paths, deeper trees and a third FRI layer must give the oracle's bytes too."""
    import cityprover as cp
    c = SG.build_gate_set(SG.CITY_COMMON, db=db, seed=40 + db, arity_bits=arity, cap_height=4, num_query_rounds=8,
                          pow_bits=8, noop_fraction=0.3)
    sh = cp_shape_of(cp, c["shape"])
    digest = [9, 9, db, 1]
    circ = cp.Circuit(prover, sh, digest, c["cs_values"])
    cp.set_gates(circ, c["gate_list"], c["num_selectors"])
    got = cp.prove(circ, c["wires"], c["public_inputs"])
    O.lib().or_set_threads(16)
    try:
        want, _ = O.prove_full(c["shape"], c["gates"], digest, c["public_inputs"], c["cs_values"], c["wires"])
    finally:
        O.lib().or_set_threads(1)
    assert got == want
    cp.verify(circ, got)
    circ.close()
