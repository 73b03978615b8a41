"""P7(vi): `fri_combine_initial` pinned against a REFERENCE proof by recovering alpha and zeta.

For query q with point x_q the value opened in the first FRI layer must equal
    alpha^2 * (R0_q(alpha) - O0(alpha)) / (x_q - zeta)  +  (R1_q(alpha) - O1(alpha)) / (x_q - g*zeta)
where R0_q / O0 are the alpha-reductions of the 256 opened leaf values / 256 openings in the order
[constants, sigmas, wires, zs, partial products, quotient chunks] and R1_q / O1 those of the 2 Z
polynomials / their `zs_next` openings. Clearing denominators and treating (zeta, zeta^2) as two
independent unknowns gives one linear equation per query whose coefficients are polynomials in
alpha; any three queries force a 3x3 determinant D(alpha) = 0, and gcd(D_123, D_124) isolates alpha.
Zeta then follows from a 2x2 linear solve, must satisfy w == zeta^2, and all 28 queries must agree.
This pins on real plonky2 output: the batch order of the opened polynomials, the alpha-power and
shift-by-count conventions of ReducingFactor, and g = omega_n for the `next` opening point."""
import json
import os

import pytest

import oracle_lib as O
from proof_format import find_leaf_index, parse_proof
from test_oracle_fri_reference import (ONE, P, ZERO, eadd, einv, emul, esub, peval, pgcd, pmul, ptrim, rev)


def padd(a, b):
    n = max(len(a), len(b))
    a = a + [ZERO] * (n - len(a)); b = b + [ZERO] * (n - len(b))
    return [eadd(x, y) for x, y in zip(a, b)]
def psub(a, b):
    n = max(len(a), len(b))
    a = a + [ZERO] * (n - len(a)); b = b + [ZERO] * (n - len(b))
    return [esub(x, y) for x, y in zip(a, b)]
def pscale(a, s): return [emul(x, s) for x in a]
def base(v): return (v % P, 0)


@pytest.mark.parametrize("which", range(10))
def test_reference_proof_fri_combine_initial(golden_dir, which):
    from proof_format import reference_proofs
    pf = parse_proof(reference_proofs(golden_dir)[which][1])
    LOG_N, LOG_DEG = 15, 12
    omega = pow(7, (P - 1) >> LOG_N, P)
    g = pow(7, (P - 1) >> LOG_DEG, P)
    o = pf["openings"]
    O0 = [tuple(e) for k in ("constants", "plonk_sigmas", "wires", "plonk_zs", "partial_products",
                             "quotient_polys") for e in o[k]]
    O1 = [tuple(e) for e in o["plonk_zs_next"]]
    assert len(O0) == 256 and len(O1) == 2
    rows = []
    for q in pf["queries"]:
        leaf, sib = q["initial"][1]
        idx = find_leaf_index(leaf, sib, pf["wires_cap"], O)
        x = 7 * pow(omega, rev(idx, LOG_N), P) % P
        vals = [v for e in q["initial"] for v in e[0]]          # 85 + 135 + 20 + 16, batch order
        A = [esub(base(vals[j]), O0[j]) for j in range(256)]     # A_q(alpha), degree 255
        zs = q["initial"][2][0][:2]
        B = [esub(base(zs[j]), O1[j]) for j in range(2)]         # B_q(alpha), degree 1
        v = tuple(q["steps"][0][0][idx & 15])
        a2A = [ZERO, ZERO] + A                                   # alpha^2 * A
        # c_u*u + c_w*w + c_1 = 0 with u = zeta, w = zeta^2
        c_u = padd(padd(pscale(a2A, base(g)), B), [emul(v, base(-(1 + g) * x))])
        c_w = emul(v, base(g))
        c_1 = psub([emul(v, base(x * x))], pscale(padd(a2A, B), base(x)))
        rows.append((idx, x, v, c_u, c_w, c_1, A, B))

    def det3(i, j, k):
        r = [rows[i], rows[j], rows[k]]
        cu, cw, c1 = [t[3] for t in r], [t[4] for t in r], [t[5] for t in r]
        # expand along the constant column c_w
        def minor(a, b):  # cu[a]*c1[b] - cu[b]*c1[a]
            return psub(pmul(cu[a], c1[b]), pmul(cu[b], c1[a]))
        d = pscale(minor(1, 2), cw[0])
        d = psub(d, pscale(minor(0, 2), cw[1]))
        d = padd(d, pscale(minor(0, 1), cw[2]))
        return ptrim(d)

    # three queries with pairwise distinct points
    seen, pick = set(), []
    for i, r in enumerate(rows):
        if r[1] not in seen:
            seen.add(r[1]); pick.append(i)
    assert len(pick) >= 4
    G = pgcd(det3(pick[0], pick[1], pick[2]), det3(pick[0], pick[1], pick[3]))
    # strip the trivial common root alpha = 0 if present (both determinants have alpha^2 * ... terms)
    while len(G) > 1 and G[0] == ZERO:
        G = G[1:]
    assert len(G) == 2, f"expected a single common root, got degree {len(G) - 1}"
    alpha = ((-G[0][0]) % P, (-G[0][1]) % P)

    # zeta from two queries: [cu_i cw_i; cu_j cw_j] (u, w)^T = -(c1_i, c1_j)^T
    def at(poly): return peval(poly, alpha)
    i, j = pick[0], pick[1]
    a, b, e = at(rows[i][3]), rows[i][4], at(rows[i][5])
    c, d, f = at(rows[j][3]), rows[j][4], at(rows[j][5])
    det = esub(emul(a, d), emul(b, c))
    u = emul(esub(emul(b, f), emul(e, d)), einv(det))
    w = emul(esub(emul(e, c), emul(a, f)), einv(det))
    assert w == emul(u, u), "zeta^2 consistency"
    zeta = u
    zeta_next = emul(zeta, base(g))
    # every query satisfies the ORIGINAL (rational) relation with these alpha, zeta
    for idx, x, v, _, _, _, A, B in rows:
        t0 = emul(emul(emul(alpha, alpha), peval(A, alpha)), einv(esub(base(x), zeta)))
        t1 = emul(peval(B, alpha), einv(esub(base(x), zeta_next)))
        assert eadd(t0, t1) == v
    # zeta is not in the trace subgroup (the prover asserts this)
    z = zeta
    for _ in range(LOG_DEG):
        z = emul(z, z)
    assert z != ONE
