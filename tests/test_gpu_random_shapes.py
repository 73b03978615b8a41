"""Randomised shape sweep: whole proofs (cp_prove_batch) on circuits with random degree, wire counts, number of
challenges, blow-up, cap height, FRI reduction schedule, query count and PoW bits must equal the oracle's bytes and
pass cp_verify. Seeds are fixed, so a failure names a reproducible configuration.
CITY_RANDOM_SHAPES / CITY_RANDOM_ZK / CITY_RANDOM_GATE_SETS widen the sweep for a soak run (defaults 24 / 8 / 8)."""
import os

import numpy as np
import pytest

import oracle_lib as O
from synth_circuit import build
from test_gpu_prove_full import cp_shape_of

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def prover():
    import cityprover
    p = cityprover.Prover(0)
    yield p
    p.close()


def random_config(seed):
    rng = np.random.default_rng(1000 + seed)
    rb = int(rng.choice([3, 3, 4]))
    db = int(rng.integers(3, 10))
    routed = int(rng.integers(2, 11)) * 4            # Arithmetic ops need 4 routed wires each
    wires = routed + int(rng.integers(0, 12))
    nc = int(rng.integers(1, 4))
    bits = db + rb
    arity, left = [], bits
    while left - rb > 1 and len(arity) < 4 and rng.random() < 0.8:
        a = int(rng.integers(1, min(4, left - rb) + 1))
        arity.append(a)
        left -= a
    cap = int(rng.integers(0, left + 1))             # the last layer's tree must still reach the cap
    return dict(db=db, num_routed=routed, num_wires=wires, chunk=1 << rb, rate_bits=rb, nc=nc, arity_bits=tuple(arity),
                cap_height=cap, num_query_rounds=int(rng.integers(1, 9)), pow_bits=int(rng.integers(0, 9)), seed=seed,
                n_copies=int(rng.integers(0, 8)))


@pytest.mark.parametrize("seed", range(int(os.environ.get("CITY_RANDOM_SHAPES", "24"))))
def test_random_shape(prover, seed):
    import cityprover as cp
    cfg = random_config(seed)
    c = build(**cfg)
    sh = cp_shape_of(cp, c["shape"])
    digest = [seed, 1, 2, 3]
    circ = cp.Circuit(prover, sh, digest, c["cs_values"])
    cp.set_gates(circ, c["gate_list"], c["num_selectors"])
    got = cp.prove(circ, c["wires"], c["public_inputs"])
    want, _ = O.prove_full(c["shape"], c["gates"], digest, c["public_inputs"], c["cs_values"], c["wires"])
    assert got == want, cfg
    cp.verify(circ, got)
    assert O.verify_full(c["shape"], c["gates"], digest, circ.cs_cap(), got) == 0, cfg
    circ.close()


@pytest.mark.parametrize("seed", range(int(os.environ.get("CITY_RANDOM_ZK", "8"))))
def test_random_shape_zero_knowledge(prover, seed):
    """The same sweep in zero-knowledge mode (FRI hiding: 4 salt elements on every wires / Z / quotient leaf)."""
    import cityprover as cp
    cfg = random_config(7000 + seed)
    c = build(**cfg)
    c["shape"].zero_knowledge = 1
    N = 1 << (cfg["db"] + cfg["rate_bits"])
    salts = np.random.default_rng(seed).integers(0, O.P, (3, O.SALT_SIZE, N), dtype=np.uint64)
    sh = cp_shape_of(cp, c["shape"])
    digest = [seed, 9, 9, 9]
    circ = cp.Circuit(prover, sh, digest, c["cs_values"])
    cp.set_gates(circ, c["gate_list"], c["num_selectors"])
    got = cp.prove_batch_zk(prover, [circ], [c["public_inputs"]], [c["wires"]], [salts])[0]
    want, _ = O.prove_full_zk(c["shape"], c["gates"], digest, c["public_inputs"], c["cs_values"], c["wires"], salts)
    assert got == want, cfg
    cp.verify(circ, got)
    assert O.verify_full(c["shape"], c["gates"], digest, circ.cs_cap(), got) == 0, cfg
    circ.close()


@pytest.mark.parametrize("seed", range(int(os.environ.get("CITY_RANDOM_GATE_SETS", "8"))))
def test_random_gate_subsets_in_one_batch(prover, seed):
    """Random subsets of the 22 gate types (plonky2 selector grouping recomputed per subset), three different circuits of
    one shape proved in ONE batch call."""
    import cityprover as cp
    import synth_gates as SG
    rng = np.random.default_rng(5000 + seed)
    always = [g for g in SG.ALL_GATES if g[0] in (SG.NOOP, SG.CONSTANT, SG.PUBLIC_INPUT)]
    rest = [g for g in SG.ALL_GATES if g not in always]
    pick = [rest[i] for i in sorted(rng.choice(len(rest), size=int(rng.integers(1, 9)), replace=False))]
    db = int(rng.integers(6, 9))
    arity = [(2,), (1, 2), (3, 1), (2, 2, 1), ()][int(rng.integers(0, 5))]
    kw = dict(db=db, arity_bits=arity, cap_height=int(rng.integers(0, 4)), num_query_rounds=int(rng.integers(1, 6)),
              pow_bits=int(rng.integers(0, 6)))
    cases = [SG.build_gate_set(always + pick, seed=100 * seed + i, **kw) for i in range(3)]
    sh = cp_shape_of(cp, cases[0]["shape"])
    circs = []
    for i, c in enumerate(cases):
        circ = cp.Circuit(prover, sh, [seed, i, 7, 7], c["cs_values"])
        cp.set_gates(circ, c["gate_list"], c["num_selectors"])
        circs.append(circ)
    got = cp.prove_batch(prover, circs, [c["public_inputs"] for c in cases], [c["wires"] for c in cases])
    O.lib().or_set_threads(8)
    try:
        for i, c in enumerate(cases):
            want, _ = O.prove_full(c["shape"], c["gates"], [seed, i, 7, 7], c["public_inputs"], c["cs_values"], c["wires"])
            assert got[i] == want, (seed, i, [SG._ID[g[0]] for g in pick], kw)
            cp.verify(circs[i], got[i])
    finally:
        O.lib().or_set_threads(1)
    for c in circs:
        c.close()
