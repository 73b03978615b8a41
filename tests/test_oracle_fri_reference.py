"""P7(v): FRI fold semantics pinned against the REFERENCE proofs (qbench_data/example.bin).

The transcript of those proofs cannot be replayed (the circuit digest is not in the fixture), so the
fold challenges are RECOVERED: for a query, the folded value is a degree-15 polynomial P_q(beta) of
the unknown challenge (Lagrange interpolation through the 16 opened coset values) and must equal
the value opened in the next layer; gcd(P_q1 - t_q1, P_q2 - t_q2) over F_p^2[X] isolates beta from two
queries, and the other 26 queries — and the final polynomial — must then agree. This pins, on real
plonky2 output: omega_N = 7^((p-1)/N), the coset shift 7, bit-reversed leaf order, the index -> point
map, the position of a value inside its FRI leaf, `compute_evaluation`, and the final-poly check."""
import json
import os

import numpy as np
import pytest

import oracle_lib as O
from proof_format import parse_proof
from reference_challenges import *  # noqa: F401,F403 (the F_p^2 / polynomial helpers other tests import from here)
from reference_challenges import LOG_N, P, recover_betas


@pytest.mark.parametrize("which", range(10))
def test_reference_proof_fri_folds(golden_dir, which):
    from proof_format import reference_proofs
    pf = parse_proof(reference_proofs(golden_dir)[which][1])
    idxs, beta0, beta1, polys0, polys1 = recover_betas(pf)    # asserts: all 28 queries + the final polynomial agree
    qs = list(zip(idxs, pf["queries"]))
    # ---- the C oracle's compute_evaluation / query point agree with the reference data
    L = O.lib()
    for (idx, q), (_, t0, x), (_, t1, x1) in list(zip(qs, polys0, polys1))[:6]:
        assert L.or_fri_query_point(idx, LOG_N) == x
        out = np.zeros(2, np.uint64)
        ev = O.arr([c for e in q["steps"][0][0] for c in e])
        L.or_fri_compute_evaluation(x, idx & 15, 4, O.ptr(ev), O.ptr(O.arr(beta0)), O.ptr(out))
        assert tuple(int(v) for v in out) == t0
        ev = O.arr([c for e in q["steps"][1][0] for c in e])
        L.or_fri_compute_evaluation(x1, (idx >> 4) & 15, 4, O.ptr(ev), O.ptr(O.arr(beta1)), O.ptr(out))
        assert tuple(int(v) for v in out) == t1
    # the proof-of-work witness is a canonical field element
    assert pf["pow_witness"] < P
