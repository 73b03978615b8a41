"""GPU whole-proof parity (A7 + A8 + tail): wires -> ProofWithPublicInputs bytes must equal the oracle's,
and the oracle's verifier (FRI + vanishing identity) must accept them."""
import numpy as np
import pytest

import oracle_lib as O
from synth_circuit import build

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def prover():
    import cityprover
    p = cityprover.Prover(0)
    yield p
    p.close()


def cp_shape_of(cp, s, num_public_inputs=5):  # the synthetic circuits carry five public inputs
    return cp.standard_recursion_shape(degree_bits=s.degree_bits, num_constants=s.num_constants,
                                       num_routed_wires=s.num_routed_wires, num_wires=s.num_wires,
                                       num_challenges=s.num_challenges, num_partial_products=s.num_partial_products,
                                       quotient_degree_factor=s.quotient_degree_factor, rate_bits=s.rate_bits,
                                       cap_height=s.cap_height, pow_bits=s.pow_bits,
                                       num_query_rounds=s.num_query_rounds,
                                       arity_bits=tuple(s.arity_bits[i] for i in range(s.n_arity)),
                                       zero_knowledge=s.zero_knowledge, num_public_inputs=num_public_inputs)


@pytest.mark.parametrize("db,R,W,arity,B", [(5, 16, 20, (2,), 2), (8, 24, 30, (2, 2), 3), (12, 80, 135, (4, 4), 2)])
def test_full_proofs_byte_identical(prover, db, R, W, arity, B):
    import cityprover as cp
    kw = dict(cap_height=4, pow_bits=16, num_query_rounds=28) if db == 12 else {}
    cases = [build(db=db, num_routed=R, num_wires=W, chunk=8, rate_bits=3, arity_bits=arity, seed=100 + i, **kw)
             for i in range(B)]
    sh = cp_shape_of(cp, cases[0]["shape"])
    circs = []
    for i, c in enumerate(cases):
        circ = cp.Circuit(prover, sh, [i, 2, 3, 4], c["cs_values"])
        cp.set_gates(circ, c["gate_list"], 1)
        circs.append(circ)
    dw = prover.to_device(np.stack([c["wires"] for c in cases]))
    got = cp.prove_batch_dev(prover, circs, [c["public_inputs"] for c in cases], dw.ptr)
    O.lib().or_set_threads(8)
    for i, c in enumerate(cases):
        want, _ = O.prove_full(c["shape"], c["gates"], [i, 2, 3, 4], c["public_inputs"], c["cs_values"], c["wires"])
        assert got[i] == want, f"proof {i}"
        assert O.verify_full(c["shape"], c["gates"], [i, 2, 3, 4], circs[i].cs_cap(), got[i]) == 0
    O.lib().or_set_threads(1)
    dw.free()
    for c in circs:
        c.close()


def test_prove_requires_gates_and_supported_types(prover):
    import cityprover as cp
    c = build(db=5, num_routed=16, num_wires=20, chunk=8, rate_bits=3, seed=1)
    sh = cp_shape_of(cp, c["shape"])
    circ = cp.Circuit(prover, sh, [0] * 4, c["cs_values"])
    dw = prover.to_device(c["wires"][None])
    with pytest.raises(cp.CityProverError):
        cp.prove_batch_dev(prover, [circ], [c["public_inputs"]], dw.ptr)      # no gate set yet
    with pytest.raises(cp.CityProverError):
        cp.set_gates(circ, [(99, 0, 0, 1, 0)], 1)                              # unknown gate type
    cp.set_gates(circ, c["gate_list"], 1)
    assert len(cp.prove_batch_dev(prover, [circ], [c["public_inputs"]], dw.ptr)[0]) > 0
    dw.free(); circ.close()


@pytest.mark.parametrize("db,frac", [(5, 0.5), (8, 0.6)])
def test_poseidon_gate_circuits(prover, db, frac):
    """Circuits with PoseidonGate rows (two selector groups, 135 wires): GPU proof == oracle proof; cp_verify
    and the oracle verifier accept; a corrupted Poseidon intermediate wire is caught by both."""
    import cityprover as cp
    c = build(db=db, num_routed=80, num_wires=135, chunk=8, rate_bits=3, arity_bits=(2, 2), seed=70 + db,
              poseidon_fraction=frac)
    assert c["poseidon_gate_index"] in c["gate_of_row"]
    sh = cp_shape_of(cp, c["shape"])
    digest = [7, 7, 7, db]
    circ = cp.Circuit(prover, sh, digest, c["cs_values"])
    cp.set_gates(circ, c["gate_list"], c["num_selectors"])
    dw = prover.to_device(c["wires"][None])
    got = cp.prove_batch_dev(prover, [circ], [c["public_inputs"]], dw.ptr)[0]
    O.lib().or_set_threads(8)
    want, _ = O.prove_full(c["shape"], c["gates"], digest, c["public_inputs"], c["cs_values"], c["wires"])
    O.lib().or_set_threads(1)
    assert got == want
    cp.verify(circ, got)
    assert O.verify_full(c["shape"], c["gates"], digest, circ.cs_cap(), got) == 0
    row = c["gate_of_row"].index(c["poseidon_gate_index"])
    w = c["wires"].copy()
    w[77, row] = (int(w[77, row]) + 1) % O.P
    dw2 = prover.to_device(w[None])
    bad = cp.prove_batch_dev(prover, [circ], [c["public_inputs"]], dw2.ptr)[0]
    with pytest.raises(cp.CityProverError, match="vanishing identity"):
        cp.verify(circ, bad)
    dw.free(); dw2.free(); circ.close()


@pytest.mark.parametrize("db,frac", [(6, 0.0), (8, 0.3)])
def test_u32_gate_circuits(prover, db, frac):
    """In-tree city-rollup gates (ComparisonGate(32,16), U32ArithmeticGate(3), U32RangeCheckGate(7)) in their own
    selector group, optionally beside PoseidonGate rows: GPU proof bytes == oracle proof bytes, both verifiers accept,
    and a corrupted witness of each gate is rejected by cp_verify's vanishing identity."""
    import cityprover as cp
    c = build(db=db, num_routed=80, num_wires=135, chunk=8, rate_bits=3, arity_bits=(2, 2), seed=90 + db,
              poseidon_fraction=frac, u32_gates=True)
    ids = c["u32_gate_ids"]
    for gid in ids.values():
        assert gid in c["gate_of_row"]
    sh = cp_shape_of(cp, c["shape"])
    digest = [8, 8, 8, db]
    circ = cp.Circuit(prover, sh, digest, c["cs_values"])
    cp.set_gates(circ, c["gate_list"], c["num_selectors"])
    dw = prover.to_device(c["wires"][None])
    got = cp.prove_batch_dev(prover, [circ], [c["public_inputs"]], dw.ptr)[0]
    O.lib().or_set_threads(8)
    want, _ = O.prove_full(c["shape"], c["gates"], digest, c["public_inputs"], c["cs_values"], c["wires"])
    O.lib().or_set_threads(1)
    assert got == want
    cp.verify(circ, got)
    assert O.verify_full(c["shape"], c["gates"], digest, circ.cs_cap(), got) == 0
    for name, wire in (("cmp", 2), ("arith", 4), ("range", 7 + 20)):
        row = c["gate_of_row"].index(ids[name])
        w = c["wires"].copy()
        w[wire, row] = (int(w[wire, row]) + 1) % O.P
        dw2 = prover.to_device(w[None])
        bad = cp.prove_batch_dev(prover, [circ], [c["public_inputs"]], dw2.ptr)[0]
        with pytest.raises(cp.CityProverError, match="vanishing identity"):
            cp.verify(circ, bad)
        dw2.free()
    dw.free(); circ.close()


def test_set_gates_rejects_oversized_gate(prover):
    """A gate whose wires do not fit the circuit's wire count must be refused at load time, not read out of bounds."""
    import cityprover as cp
    c = build(db=5, num_routed=16, num_wires=24, chunk=8, rate_bits=3, arity_bits=(2,), seed=3)
    sh = cp_shape_of(cp, c["shape"])
    circ = cp.Circuit(prover, sh, [1, 1, 1, 1], c["cs_values"])
    for bad in [(cp.GATE_U32_RANGE_CHECK, 0, 0, 1, 8, 0), (cp.GATE_COMPARISON, 0, 0, 1, 32, 16),
                (cp.GATE_POSEIDON, 0, 0, 1, 0, 0), (cp.GATE_COMPARISON, 0, 0, 1, 32, 0)]:
        with pytest.raises(cp.CityProverError):
            cp.set_gates(circ, [bad], 1)
    circ.close()


def test_host_memory_entry_points(prover):
    """cp_prove / cp_prove_batch_host (wires in host memory, the Rust shim's call) == the device-pointer path."""
    import cityprover as cp
    cases = [build(db=8, num_routed=24, num_wires=30, chunk=8, rate_bits=3, arity_bits=(2, 2), seed=300 + i) for i in range(3)]
    sh = cp_shape_of(cp, cases[0]["shape"])
    circs = []
    for i, c in enumerate(cases):
        circ = cp.Circuit(prover, sh, [i, 5, 5, 5], c["cs_values"])
        cp.set_gates(circ, c["gate_list"], 1)
        circs.append(circ)
    dw = prover.to_device(np.stack([c["wires"] for c in cases]))
    pis = [c["public_inputs"] for c in cases]
    want = cp.prove_batch_dev(prover, circs, pis, dw.ptr)
    assert cp.prove_batch(prover, circs, pis, [c["wires"] for c in cases]) == want
    assert cp.prove(circs[1], cases[1]["wires"], pis[1]) == want[1]
    assert cp.prove_batch(prover, circs[:1], pis[:1], [cases[0]["wires"]]) == want[:1]   # smaller batch after a larger one
    with pytest.raises(ValueError):
        cp.prove_batch(prover, circs, pis, [cases[0]["wires"], cases[1]["wires"], np.zeros(7, np.uint64)])
    dw.free()
    for c in circs:
        c.close()


def test_contexts_in_parallel_are_deterministic():
    """Three contexts on one GPU proving concurrently from three host threads (the bench / worker set-up): every proof
    must equal the one a single context produces for the same job — no cross-context interference through the
    page-locked staging ring, the arena or the cached tables."""
    import threading
    import cityprover as cp
    cases = [build(db=9, num_routed=24, num_wires=30, chunk=8, rate_bits=3, arity_bits=(2, 2), seed=500 + i, pow_bits=6)
             for i in range(4)]
    B, iters, T = 8, 6, 3
    pick = [i % len(cases) for i in range(B)]

    def make(p):
        sh = cp_shape_of(cp, cases[0]["shape"])
        circs = []
        for i, c in enumerate(cases):
            circ = cp.Circuit(p, sh, [i, 4, 4, 4], c["cs_values"])
            cp.set_gates(circ, c["gate_list"], 1)
            circs.append(circ)
        return circs

    def run(p, circs):
        return cp.prove_batch(p, [circs[i] for i in pick], [cases[i]["public_inputs"] for i in pick],
                              [cases[i]["wires"] for i in pick])

    p0 = cp.Prover(0)
    c0 = make(p0)
    want = run(p0, c0)
    for i in range(len(cases)):
        assert want[i] == want[i + len(cases)]          # same job twice in one batch
    for c in c0:
        c.close()
    p0.close()
    errors = []

    def worker(t):
        try:
            p = cp.Prover(0)
            circs = make(p)
            for _ in range(iters):
                if run(p, circs) != want:
                    errors.append("thread %d: proof bytes differ" % t)
            for c in circs:
                c.close()
            p.close()
        except Exception as e:   # noqa: BLE001
            errors.append("thread %d: %r" % (t, e))

    ths = [threading.Thread(target=worker, args=(t,)) for t in range(T)]
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    assert not errors, errors


def test_zero_knowledge_circuits(prover):
    """standard_recursion_zk_config path: salted leaves on the wires / Z / quotient oracles (FRI hiding). GPU bytes ==
    oracle bytes for the same salts (three different circuits in one batch), cp_verify and the oracle verifier accept,
    and the plain / zk entry points refuse each other's circuits."""
    import cityprover as cp
    from test_oracle_full import zk_case
    cases = [zk_case(seed=70 + i, db=7) for i in range(3)]
    sh = cp_shape_of(cp, cases[0]["shape"])
    assert sh.zero_knowledge == 1
    circs = []
    for i, c in enumerate(cases):
        circ = cp.Circuit(prover, sh, [i, 6, 6, 6], c["cs_values"])
        cp.set_gates(circ, c["gate_list"], 1)
        circs.append(circ)
    pis = [c["public_inputs"] for c in cases]
    got = cp.prove_batch_zk(prover, circs, pis, [c["wires"] for c in cases], [c["salts"] for c in cases])
    for i, c in enumerate(cases):
        want, _ = O.prove_full_zk(c["shape"], c["gates"], [i, 6, 6, 6], c["public_inputs"], c["cs_values"], c["wires"], c["salts"])
        assert got[i] == want
        cp.verify(circs[i], got[i])
        assert O.verify_full(c["shape"], c["gates"], [i, 6, 6, 6], circs[i].cs_cap(), got[i]) == 0
    bad = bytearray(got[0])
    bad[-200] ^= 1    # inside the final polynomial
    with pytest.raises(cp.CityProverError):
        cp.verify(circs[0], bytes(bad))
    with pytest.raises(cp.CityProverError, match="zero-knowledge"):
        cp.prove_batch(prover, circs, pis, [c["wires"] for c in cases])
    plain = build(db=7, num_routed=16, num_wires=24, chunk=8, rate_bits=3, arity_bits=(2, 1), seed=70, cap_height=2)
    pc = cp.Circuit(prover, cp_shape_of(cp, plain["shape"]), [9, 9, 9, 9], plain["cs_values"])
    cp.set_gates(pc, plain["gate_list"], 1)
    with pytest.raises(cp.CityProverError, match="not zero-knowledge"):
        cp.prove_batch_zk(prover, [pc], [plain["public_inputs"]], [plain["wires"]], [cases[0]["salts"]])
    pc.close()
    for c in circs:
        c.close()


def test_internal_lanes_give_the_same_bytes(prover):
    """cp_ctx_set_lanes: one cp_prove_batch_host call split among internal contexts returns the bytes of the unsplit call
    (plain and zero-knowledge entry points); a failing part fails the whole call and leaves no outputs behind."""
    import cityprover as cp
    p2 = cp.Prover(0)
    try:
        cases = [build(db=6, num_routed=16, num_wires=20, chunk=8, rate_bits=3, arity_bits=(2, 2), seed=300 + i) for i in range(3)]
        sh = cp_shape_of(cp, cases[0]["shape"])
        circs = []
        for i, c in enumerate(cases):
            circ = cp.Circuit(p2, sh, [i, 5, 5, 5], c["cs_values"])
            cp.set_gates(circ, c["gate_list"], c["num_selectors"])
            circs.append(circ)
        pick = [i % 3 for i in range(13)]
        args = ([circs[i] for i in pick], [cases[i]["public_inputs"] for i in pick], [cases[i]["wires"] for i in pick])
        want = cp.prove_batch(p2, *args)
        for lanes in (2, 3):
            p2.set_lanes(lanes)
            assert cp.prove_batch(p2, *args) == want
        bad = list(args[1])
        bad[12] = np.full_like(np.asarray(bad[12], np.uint64), 2**64 - 1)      # not canonical: the last part fails
        with pytest.raises(cp.CityProverError, match="canonical"):
            cp.prove_batch(p2, args[0], bad, args[2])
        p2.set_lanes(1)
        assert cp.prove_batch(p2, *args) == want
        with pytest.raises(cp.CityProverError):
            p2.set_lanes(9)
        for c in circs:
            c.close()
    finally:
        p2.close()


def test_internal_lanes_zero_knowledge(prover):
    """The split path of cp_prove_batch_zk_host (salts travel with their part of the batch): same bytes as unsplit."""
    import cityprover as cp
    from test_oracle_full import zk_case
    p2 = cp.Prover(0)
    try:
        cases = [zk_case(seed=170 + i, db=6) for i in range(3)]
        sh = cp_shape_of(cp, cases[0]["shape"])
        circs = []
        for i, c in enumerate(cases):
            circ = cp.Circuit(p2, sh, [i, 8, 8, 8], c["cs_values"])
            cp.set_gates(circ, c["gate_list"], 1)
            circs.append(circ)
        pick = [i % 3 for i in range(9)]
        args = ([circs[i] for i in pick], [cases[i]["public_inputs"] for i in pick], [cases[i]["wires"] for i in pick],
                [cases[i]["salts"] for i in pick])
        want = cp.prove_batch_zk(p2, *args)
        p2.set_lanes(2)
        assert cp.prove_batch_zk(p2, *args) == want
        for c in circs:
            c.close()
    finally:
        p2.close()


@pytest.mark.parametrize("n_proofs", [1, 2, 5])
def test_host_and_device_transcripts_give_the_same_bytes(n_proofs):
    """The Fiat-Shamir transcripts are hashed on the device for a batch and on the host for one or two proofs
    (cp_ctx_set_device_transcript; csrc/fri_engine.inc `Transcript`): forced either way, for every batch size, the proofs are the
    oracle's bytes."""
    import cityprover as cp
    from synth_circuit import build as build_circuit
    prover = cp.Prover(0)
    c = build_circuit(db=7, num_routed=16, num_wires=20, chunk=8, rate_bits=3, arity_bits=(2, 2), seed=40 + n_proofs)
    s = c["shape"]
    sh = cp.standard_recursion_shape(
        degree_bits=s.degree_bits, num_constants=s.num_constants, num_routed_wires=s.num_routed_wires, num_wires=s.num_wires,
        num_challenges=s.num_challenges, num_partial_products=s.num_partial_products, quotient_degree_factor=s.quotient_degree_factor,
        rate_bits=s.rate_bits, cap_height=s.cap_height, pow_bits=s.pow_bits, num_query_rounds=s.num_query_rounds,
        arity_bits=tuple(s.arity_bits[i] for i in range(s.n_arity)), num_public_inputs=len(c["public_inputs"]))
    circ = cp.Circuit(prover, sh, [9, 8, 7, 6], c["cs_values"])
    cp.set_gates(circ, c["gate_list"], 1)
    want, _ = O.prove_full(c["shape"], c["gates"], [9, 8, 7, 6], c["public_inputs"], c["cs_values"], c["wires"])
    dw = prover.to_device(np.stack([c["wires"]] * n_proofs))
    for mode in (0, 1, -1):
        prover.set_device_transcript(mode)
        got = cp.prove_batch_dev(prover, [circ] * n_proofs, [c["public_inputs"]] * n_proofs, dw.ptr)
        assert all(g == want for g in got), f"mode {mode}"
    with pytest.raises(cp.CityProverError):
        prover.set_device_transcript(2)
    dw.free()
    circ.close()
    prover.close()
