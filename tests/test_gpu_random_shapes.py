"""Randomised shape sweep: whole proofs (cp_prove_batch) on circuits with random degree, wire counts, number of
challenges, blow-up, cap height, FRI reduction schedule, query count and PoW bits must equal the oracle's bytes and
pass cp_verify. Seeds are fixed, so a failure names a reproducible configuration."""
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


@pytest.mark.parametrize("seed", range(24))
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
