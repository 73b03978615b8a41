"""cp_batcher (include/cityprover.h): concurrent one-proof callers — the reference's worker loops,
city_rollup_core_worker/src/actors/simple.rs:32-56, as threads of one process — merged into cp_prove_batch_host launches.
The bytes must be those of cp_prove (and so the oracle's), whatever got batched with whatever; a failing request must
fail alone."""
import threading

import numpy as np
import pytest

import oracle_lib as O
from synth_circuit import build
from test_gpu_prove_full import cp_shape_of

pytestmark = pytest.mark.gpu

P = 0xFFFFFFFF00000001


def _circuits(cp, prover, specs):
    """[(case, circuit)]: synthetic circuits of the given (db, R, W, arity, seed)"""
    out = []
    for i, (db, R, W, arity, seed) in enumerate(specs):
        c = build(db=db, num_routed=R, num_wires=W, chunk=8, rate_bits=3, arity_bits=arity, seed=seed)
        circ = cp.Circuit(prover, cp_shape_of(cp, c["shape"]), [i, 5, 6, 7], c["cs_values"])
        cp.set_gates(circ, c["gate_list"], 1)
        out.append((c, circ))
    return out


def _run_callers(batcher, jobs, n_threads):
    """jobs: [(circuit, wires, public_inputs)]; thread t proves jobs t, t + n_threads, ... one call at a time.
    Returns results in job order: bytes or the exception raised."""
    results = [None] * len(jobs)
    start = threading.Barrier(n_threads)

    def caller(t):
        start.wait()
        for j in range(t, len(jobs), n_threads):
            circ, wires, pis = jobs[j]
            try:
                results[j] = batcher.prove(circ, wires, pis)
            except Exception as e:  # noqa: BLE001 - the test inspects it
                results[j] = e

    threads = [threading.Thread(target=caller, args=(t,)) for t in range(n_threads)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    return results


@pytest.mark.parametrize("lanes,linger_us", [(1, 0), (1, 2000), (3, 0)])
def test_concurrent_callers_get_cp_prove_bytes(lanes, linger_us):
    """16 threads, two shapes (three circuits of one, one of the other) interleaved: every caller receives the bytes
    cp_prove gives for its job; the calls were merged (fewer launches than calls); shapes never share a batch
    (a mixed batch would have been refused and shown up in retried_singly)."""
    import cityprover as cp
    prover = cp.Prover(0)
    prover.set_lanes(lanes)
    cc = _circuits(cp, prover, [(6, 16, 20, (2, 2), 11), (6, 16, 20, (2, 2), 12), (6, 16, 20, (2, 2), 13), (7, 24, 30, (2, 2), 14)])
    want = [cp.prove(circ, c["wires"], c["public_inputs"]) for c, circ in cc]
    O.lib().or_set_threads(8)
    c0, circ0 = cc[0]
    ref, _ = O.prove_full(c0["shape"], c0["gates"], [0, 5, 6, 7], c0["public_inputs"], c0["cs_values"], c0["wires"])
    O.lib().or_set_threads(1)
    assert want[0] == ref                                   # cp_prove itself is pinned to the oracle
    batcher = cp.Batcher(prover, max_batch=8, linger_us=linger_us)
    jobs = [(cc[j % 4][1], cc[j % 4][0]["wires"], cc[j % 4][0]["public_inputs"]) for j in range(96)]
    got = _run_callers(batcher, jobs, 16)
    for j, g in enumerate(got):
        assert not isinstance(g, Exception), f"job {j}: {g}"
        assert g == want[j % 4], f"job {j}"
    st = batcher.stats()
    assert st["calls"] == st["proofs"] == 96
    assert st["batches"] < 96 and 2 <= st["largest_batch"] <= 8
    assert st["retried_singly"] == 0
    batcher.close()
    for _, circ in cc:
        circ.close()
    prover.close()


def test_a_bad_request_is_refused_before_it_is_queued():
    """One caller hands in a non-canonical public input (cp_prove: CP_ERR_INVALID_ARG). It is refused before it reaches the queue
    (inside a merged batch it would fail the batch and have every neighbour proved again singly); the failing caller
    gets the status and the message in its own thread, the others their proofs in ONE pass."""
    import cityprover as cp
    prover = cp.Prover(0)
    (c, circ), = _circuits(cp, prover, [(6, 16, 20, (2, 2), 21)])
    want = cp.prove(circ, c["wires"], c["public_inputs"])
    bad = np.array(c["public_inputs"], dtype=np.uint64).copy()
    bad[0] = np.uint64(P)
    batcher = cp.Batcher(prover, max_batch=16, linger_us=20000)   # linger: the calls below share one batch
    jobs = [(circ, c["wires"], bad if j == 3 else c["public_inputs"]) for j in range(8)]
    got = _run_callers(batcher, jobs, 8)
    for j, g in enumerate(got):
        if j == 3:
            assert isinstance(g, cp.CityProverError) and "not canonical" in str(g), g
        else:
            assert g == want, f"job {j}: {g!r:.80}"
    st = batcher.stats()
    assert st["calls"] == 7 and st["proofs"] == 7 and st["retried_singly"] == 0
    batcher.close()
    circ.close()
    prover.close()


def test_a_failing_batch_is_retried_request_by_request():
    """A failure that only shows inside the batch (here: an injected allocation failure in one of its phases, as an
    out-of-memory would) fails the merged call as a whole; every request of it is then proved singly and gets its bytes."""
    import cityprover as cp
    prover = cp.Prover(0)
    (c, circ), = _circuits(cp, prover, [(6, 16, 20, (2, 2), 22)])
    want = cp.prove(circ, c["wires"], c["public_inputs"])
    batcher = cp.Batcher(prover, max_batch=16, linger_us=20000)
    jobs = [(circ, c["wires"], c["public_inputs"]) for _ in range(6)]
    assert prover.lib.cp_fault_inject(1, 2) == 0        # CP_FAULT_ALLOC: the third checkpoint from now fails, once
    try:
        got = _run_callers(batcher, jobs, 6)
    finally:
        prover.lib.cp_fault_inject(1, -1)
    for j, g in enumerate(got):
        assert g == want, f"job {j}: {g!r:.80}"
    st = batcher.stats()
    assert st["calls"] == 6 and st["retried_singly"] >= 1
    batcher.close()
    circ.close()
    prover.close()


def test_batcher_argument_checks():
    import cityprover as cp
    a, b = cp.Prover(0), cp.Prover(0)
    (c, circ), = _circuits(cp, a, [(5, 16, 20, (2,), 31)])
    for kw in (dict(max_batch=0), dict(max_batch=5000), dict(linger_us=2_000_000)):
        with pytest.raises(cp.CityProverError):
            cp.Batcher(a, **kw)
    foreign = cp.Batcher(b)
    with pytest.raises(cp.CityProverError, match="another context"):
        foreign.prove(circ, c["wires"], c["public_inputs"])
    assert foreign.stats()["calls"] == 0
    foreign.close()
    own = cp.Batcher(a)
    assert own.prove(circ, c["wires"], c["public_inputs"]) == cp.prove(circ, c["wires"], c["public_inputs"])
    assert own.stats() == dict(calls=1, batches=1, proofs=1, largest_batch=1, retried_singly=0)
    own.close()
    circ.close()
    a.close(); b.close()
