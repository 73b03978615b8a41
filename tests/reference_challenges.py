"""FRI challenges of the REFERENCE proofs (qbench_data/example.bin), recovered by algebra from the proof bytes alone.

The transcript of those proofs cannot be replayed: their circuits (digest, constants/sigmas cap) are not in the dump.
But every challenge the query phase needs is over-determined by the proof:
  * fold challenges beta_l: for a query, the folded value is a degree-15 polynomial P_q(beta) of the unknown challenge
    (Lagrange interpolation through the 16 opened coset values) and must equal the value opened in the next layer;
    gcd(P_q1 - t_q1, P_q2 - t_q2) over F_p^2[X] isolates beta from two queries (the other 26 must then agree);
  * alpha and zeta: clearing the denominators of fri_combine_initial and treating (zeta, zeta^2) as two unknowns gives one
    linear equation per query whose coefficients are polynomials in alpha; any three queries force a 3x3 determinant
    D(alpha) = 0, gcd(D_123, D_124) isolates alpha, zeta follows from a 2x2 solve and must satisfy w == zeta^2.
Used by the oracle pinning tests (P7(v), P7(vi)) and by the tests that hold PRODUCT code against the reference proofs
(cp_verify's query phase, the HIP fold / combine kernels)."""
import functools

import oracle_lib as O
from proof_format import find_leaf_index, parse_proof, reference_proofs

P = O.P
W = 7
LOG_N, LOG_DEG = 15, 12


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




def query_indices(pf):
    """leaf index of every query round (searched: the transcript is not replayable)"""
    out = []
    for q in pf["queries"]:
        leaf, sib = q["initial"][1]
        idx = find_leaf_index(leaf, sib, pf["wires_cap"], O)
        assert idx is not None
        out.append(idx)
    return out


def recover_betas(pf, idxs=None):
    """-> (query indices, beta0, beta1, polys0, polys1); asserts that all 28 queries and the final polynomial agree"""
    omega = pow(7, (P - 1) >> LOG_N, P)
    idxs = idxs or query_indices(pf)
    qs = list(zip(idxs, pf["queries"]))
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
    return idxs, beta0, beta1, polys0, polys1


def recover_alpha_zeta(pf, idxs=None):
    """-> (alpha, zeta); asserts that all 28 queries satisfy fri_combine_initial with them"""
    idxs = idxs or query_indices(pf)
    omega = pow(7, (P - 1) >> LOG_N, P)
    g = pow(7, (P - 1) >> LOG_DEG, P)
    o = pf["openings"]
    O0 = [tuple(e) for k in ("constants", "plonk_sigmas", "wires", "plonk_zs", "partial_products",
                             "quotient_polys") for e in o[k]]
    O1 = [tuple(e) for e in o["plonk_zs_next"]]
    assert len(O0) == 256 and len(O1) == 2
    rows = []
    for idx, q in zip(idxs, pf["queries"]):
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
    return alpha, zeta


@functools.lru_cache(maxsize=None)
def challenges(golden_dir, which):
    """(proof bytes, parsed proof, dict(x_indices, alpha, zeta, betas)) of reference proof `which`"""
    meta, blob = reference_proofs(golden_dir)[which]
    pf = parse_proof(blob)
    idxs, beta0, beta1, _, _ = recover_betas(pf)
    alpha, zeta = recover_alpha_zeta(pf, idxs)
    return blob, pf, dict(x_indices=idxs, alpha=alpha, zeta=zeta, betas=[beta0, beta1], meta=meta)
