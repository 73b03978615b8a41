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
from proof_format import find_leaf_index, parse_proof

P = O.P
W = 7


# ---- F_p^2 = F_p[X]/(X^2-7) on Python ints -------------------------------------------------------
def eadd(a, b): return ((a[0] + b[0]) % P, (a[1] + b[1]) % P)
def esub(a, b): return ((a[0] - b[0]) % P, (a[1] - b[1]) % P)
def emul(a, b): return ((a[0] * b[0] + W * a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)
def einv(a):
    n = pow((a[0] * a[0] - W * a[1] * a[1]) % P, P - 2, P)
    return (a[0] * n % P, (-a[1]) * n % P)
ZERO, ONE = (0, 0), (1, 0)


def ptrim(p):
    while p and p[-1] == ZERO:
        p = p[:-1]
    return p
def pmul(a, b):
    out = [ZERO] * (len(a) + len(b) - 1)
    for i, x in enumerate(a):
        for j, y in enumerate(b):
            out[i + j] = eadd(out[i + j], emul(x, y))
    return out
def pmod(a, b):
    a = ptrim(list(a)); b = ptrim(list(b))
    inv = einv(b[-1])
    while len(a) >= len(b):
        f = emul(a[-1], inv)
        s = len(a) - len(b)
        for i, y in enumerate(b):
            a[s + i] = esub(a[s + i], emul(f, y))
        a = ptrim(a)
    return a
def pgcd(a, b):
    a, b = ptrim(list(a)), ptrim(list(b))
    while b:
        a, b = b, pmod(a, b)
    inv = einv(a[-1])
    return [emul(c, inv) for c in a]
def peval(p, x):
    acc = ZERO
    for c in reversed(p):
        acc = eadd(emul(acc, x), c)
    return acc


def lagrange_coeffs(pts, vals):
    """coefficients (in beta) of the interpolant through (pts[i] in F_p, vals[i] in F_p^2)"""
    n = len(pts)
    total = [ZERO] * n
    for i in range(n):
        num = [ONE]
        den = 1
        for j in range(n):
            if j != i:
                num = pmul(num, [((-pts[j]) % P, 0), ONE])
                den = den * (pts[i] - pts[j]) % P
        s = emul(vals[i], (pow(den, P - 2, P), 0))
        for k, c in enumerate(num):
            total[k] = eadd(total[k], emul(c, s))
    return total


def rev(x, bits):
    return int(format(x, "0%db" % bits)[::-1], 2) if bits else 0


def fold_poly(x, within, arity_bits, evals):
    arity = 1 << arity_bits
    g = pow(7, (P - 1) >> arity_bits, P)
    ev = [tuple(evals[rev(i, arity_bits)]) for i in range(arity)]
    start = x * pow(g, arity - rev(within, arity_bits), P) % P
    pts = [start * pow(g, i, P) % P for i in range(arity)]
    return lagrange_coeffs(pts, ev)


@pytest.mark.parametrize("which", range(10))
def test_reference_proof_fri_folds(golden_dir, which):
    from proof_format import reference_proofs
    pf = parse_proof(reference_proofs(golden_dir)[which][1])
    LOG_N = 15
    omega = pow(7, (P - 1) >> LOG_N, P)
    qs = []
    for q in pf["queries"]:
        leaf, sib = q["initial"][1]
        idx = find_leaf_index(leaf, sib, pf["wires_cap"], O)
        assert idx is not None
        qs.append((idx, q))
    # ---- layer 0: recover beta0 from the first two distinct queries
    polys0 = []
    for idx, q in qs:
        x = 7 * pow(omega, rev(idx, LOG_N), P) % P
        ev0, ev1 = q["steps"][0][0], q["steps"][1][0]
        target = tuple(ev1[(idx >> 4) & 15])
        polys0.append((fold_poly(x, idx & 15, 4, ev0), target, x))
    def shifted(i):
        p, t, _ = polys0[i]
        return [esub(p[0], t)] + p[1:]
    distinct = [i for i in range(len(qs)) if qs[i][0] >> 4 != qs[0][0] >> 4]
    g0 = pgcd(shifted(0), shifted(distinct[0]))
    assert len(g0) == 2, "two queries should isolate a single common root"
    beta0 = ((-g0[0][0]) % P, (-g0[0][1]) % P)
    for p, t, _ in polys0:                                   # all 28 queries agree with that beta
        assert peval(p, beta0) == t
    # ---- layer 1 -> final polynomial
    polys1 = []
    for (idx, q), (_, _, x) in zip(qs, polys0):
        x1 = pow(x, 16, P)
        ev1 = q["steps"][1][0]
        x2 = pow(x1, 16, P)
        target = peval([tuple(c) for c in pf["final_poly"]], (x2, 0))
        polys1.append((fold_poly(x1, (idx >> 4) & 15, 4, ev1), target, x1))
    def shifted1(i):
        p, t, _ = polys1[i]
        return [esub(p[0], t)] + p[1:]
    distinct1 = [i for i in range(len(qs)) if qs[i][0] >> 8 != qs[0][0] >> 8]
    g1 = pgcd(shifted1(0), shifted1(distinct1[0]))
    assert len(g1) == 2
    beta1 = ((-g1[0][0]) % P, (-g1[0][1]) % P)
    for p, t, _ in polys1:
        assert peval(p, beta1) == t
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
