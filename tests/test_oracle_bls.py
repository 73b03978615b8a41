"""Pins of the BLS12-381 G1 oracle (oracle/bls12_381.c): the constants satisfy the published parameterisation
p = (x-1)^2 r / 3 + x, r = x^4 - x^2 + 1 at x = -0xd201000000010000, the generator is on y^2 = x^3 + 4, r*G = infinity,
and the group law agrees with an independent affine implementation in Python integers."""
import numpy as np

import oracle_lib as O

X = -0xd201000000010000


def affine_add(P, Q, p):
    if P is None:
        return Q
    if Q is None:
        return P
    x1, y1 = P
    x2, y2 = Q
    if x1 == x2:
        if (y1 + y2) % p == 0:
            return None
        lam = 3 * x1 * x1 * pow(2 * y1, p - 2, p) % p
    else:
        lam = (y2 - y1) * pow(x2 - x1, p - 2, p) % p
    x3 = (lam * lam - x1 - x2) % p
    return x3, (lam * (x1 - x3) - y1) % p


def affine_mul(k, P, p):
    R = None
    while k:
        if k & 1:
            R = affine_add(R, P, p)
        P = affine_add(P, P, p)
        k >>= 1
    return R


def test_constants_satisfy_the_curve_parameterisation():
    p, r, G = O.bls_constants()
    assert r == X**4 - X**2 + 1
    assert p == (X - 1)**2 * r // 3 + X and ((X - 1)**2 * r) % 3 == 0
    assert (G[1] * G[1] - G[0]**3 - 4) % p == 0 and O.bls_g1_on_curve(G)
    assert not O.bls_g1_on_curve((G[0], G[1] + 1))
    assert O.bls_g1_mul(G, r) is None
    assert O.bls_g1_mul(G, r + 5) == O.bls_g1_mul(G, 5)


def test_group_law_against_python_integers():
    p, r, G = O.bls_constants()
    rng = np.random.default_rng(1)
    for _ in range(6):
        a, b = (int.from_bytes(rng.bytes(32), "little") % r for _ in range(2))
        A, B = O.bls_g1_mul(G, a), O.bls_g1_mul(G, b)
        assert A == affine_mul(a, G, p) and B == affine_mul(b, G, p)
        assert O.bls_g1_add(A, B) == affine_add(A, B, p) == O.bls_g1_mul(G, (a + b) % r)
    two = O.bls_g1_add(G, G)
    assert two == O.bls_g1_mul(G, 2) == affine_add(G, G, p)
    assert O.bls_g1_add(G, None) == G and O.bls_g1_add(None, None) is None
    neg = (G[0], p - G[1])
    assert O.bls_g1_add(G, neg) is None
    assert O.bls_g1_mul(G, 0) is None and O.bls_g1_mul(None, 12345) is None


def test_msm_is_the_sum_of_the_scalar_multiples():
    p, r, G = O.bls_constants()
    rng = np.random.default_rng(2)
    ks = [int.from_bytes(rng.bytes(32), "little") % r for _ in range(9)]
    pts = [O.bls_g1_mul(G, 3 + i) for i in range(9)]
    scal = np.array([[(k >> (64 * j)) & (2**64 - 1) for j in range(4)] for k in ks], dtype=np.uint64)
    xy = np.array([O.bls_point(P)[0] for P in pts], dtype=np.uint64)
    want = None
    for k, P in zip(ks, pts):
        want = affine_add(want, affine_mul(k, P, p), p)
    for threads in (1, 4):
        O.lib().or_set_threads(threads)
        assert O.bls_g1_msm(scal, xy) == want
    O.lib().or_set_threads(1)
    assert want == O.bls_g1_mul(G, sum(k * (3 + i) for i, k in enumerate(ks)) % r)


# ---- G2 ------------------------------------------------------------------------------------------------------
def f2_mul(a, b, p):
    return ((a[0] * b[0] - a[1] * b[1]) % p, (a[0] * b[1] + a[1] * b[0]) % p)


def f2_inv(a, p):
    d = pow((a[0] * a[0] + a[1] * a[1]) % p, p - 2, p)
    return (a[0] * d % p, -a[1] * d % p)


def affine2_add(P, Q, p):
    if P is None:
        return Q
    if Q is None:
        return P
    (x1, y1), (x2, y2) = P, Q
    if x1 == x2:
        if ((y1[0] + y2[0]) % p, (y1[1] + y2[1]) % p) == (0, 0):
            return None
        lam = f2_mul(f2_mul((3, 0), f2_mul(x1, x1, p), p), f2_inv(((2 * y1[0]) % p, (2 * y1[1]) % p), p), p)
    else:
        lam = f2_mul(((y2[0] - y1[0]) % p, (y2[1] - y1[1]) % p), f2_inv(((x2[0] - x1[0]) % p, (x2[1] - x1[1]) % p), p), p)
    l2 = f2_mul(lam, lam, p)
    x3 = ((l2[0] - x1[0] - x2[0]) % p, (l2[1] - x1[1] - x2[1]) % p)
    t = f2_mul(lam, ((x1[0] - x3[0]) % p, (x1[1] - x3[1]) % p), p)
    return x3, ((t[0] - y1[0]) % p, (t[1] - y1[1]) % p)


def affine2_mul(k, P, p):
    R = None
    while k:
        if k & 1:
            R = affine2_add(R, P, p)
        P = affine2_add(P, P, p)
        k >>= 1
    return R


def test_g2_generator_and_group_law():
    p, r, _ = O.bls_constants()
    G2 = O.bls_g2_generator()
    x, y = G2
    x3 = f2_mul(f2_mul(x, x, p), x, p)
    assert f2_mul(y, y, p) == ((x3[0] + 4) % p, (x3[1] + 4) % p) and O.bls_g2_on_curve(G2)     # y^2 = x^3 + 4(1 + u)
    assert not O.bls_g2_on_curve((x, ((y[0] + 1) % p, y[1])))
    assert O.bls_g2_mul(G2, r) is None
    rng = np.random.default_rng(8)
    for _ in range(3):
        a, b = (int.from_bytes(rng.bytes(32), "little") % r for _ in range(2))
        A, B = O.bls_g2_mul(G2, a), O.bls_g2_mul(G2, b)
        assert A == affine2_mul(a, G2, p)
        assert O.bls_g2_add(A, B) == affine2_add(A, B, p) == O.bls_g2_mul(G2, (a + b) % r)
    assert O.bls_g2_add(G2, G2) == O.bls_g2_mul(G2, 2) == affine2_add(G2, G2, p)
    neg = (x, ((-y[0]) % p, (-y[1]) % p))
    assert O.bls_g2_add(G2, neg) is None and O.bls_g2_add(G2, None) == G2
    ks = [int.from_bytes(rng.bytes(32), "little") % r for _ in range(5)]
    pts = [O.bls_g2_mul(G2, 2 + i) for i in range(5)]
    scal = np.array([[(k >> (64 * j)) & (2**64 - 1) for j in range(4)] for k in ks], dtype=np.uint64)
    xy = np.array([O.bls_point2(P)[0] for P in pts], dtype=np.uint64)
    assert O.bls_g2_msm(scal, xy) == O.bls_g2_mul(G2, sum(k * (2 + i) for i, k in enumerate(ks)) % r)


# ---- F_r NTT ---------------------------------------------------------------------------------------------------
def fr_rand(rng, n, r):
    vals = [int.from_bytes(rng.bytes(32), "little") % r for _ in range(n)]
    return vals, np.array([[(v >> (64 * j)) & (2**64 - 1) for j in range(4)] for v in vals], dtype=np.uint64)


def fr_ints(a):
    return [sum(int(a[i, j]) << (64 * j) for j in range(4)) for i in range(a.shape[0])]


def test_fr_roots_and_ntt_definition():
    _, r, _ = O.bls_constants()
    assert (r - 1) % (1 << 32) == 0
    for k in (1, 2, 5, 32):
        w = O.fr_root_of_unity(k)
        assert w == pow(7, (r - 1) >> k, r) and pow(w, 1 << k, r) == 1 and pow(w, 1 << (k - 1), r) == r - 1
    rng = np.random.default_rng(3)
    for log_n in (0, 1, 3, 6):
        n = 1 << log_n
        vals, a = fr_rand(rng, n, r)
        w = pow(7, (r - 1) >> log_n, r)
        want = [sum(vals[j] * pow(w, j * k, r) for j in range(n)) % r for k in range(n)]
        assert fr_ints(O.fr_ntt(a)) == want == fr_ints(O.fr_dft_naive(a))
        assert (O.fr_ntt(O.fr_ntt(a), inverse=True) == a).all()
        s = 7
        cos = [sum(vals[j] * pow(s * pow(w, k, r), j, r) for j in range(n)) % r for k in range(n)]   # evaluations on s*<w>
        assert fr_ints(O.fr_ntt(a, shift=s)) == cos
        assert (O.fr_ntt(O.fr_ntt(a, shift=s), inverse=True, shift=s) == a).all()


# ---- Groth16 quotient ------------------------------------------------------------------------------------------
def to_limbs(vals):
    return np.array([[(v >> (64 * j)) & (2**64 - 1) for j in range(4)] for v in vals], dtype=np.uint64)


def interp_eval(evals, z, w, r):
    """value at z of the polynomial of degree < n with the given evaluations on <w> (barycentric form)"""
    n = len(evals)
    zn = (pow(z, n, r) - 1) * pow(n, -1, r) % r
    return zn * sum(e * pow(w, i, r) * pow(z - pow(w, i, r), -1, r) for i, e in enumerate(evals)) % r


def test_groth16_quotient_identity():
    """h = (a b - c) / (x^n - 1): for a satisfied R1CS (c = a o b on the domain) the division is exact, h has degree
    <= n - 2 and a(z) b(z) - c(z) = h(z) (z^n - 1) at any z; for arbitrary c the output is the degree < n polynomial
    that agrees with the quotient on the coset 7<w> (what the coset transform computes). Python integers throughout."""
    _, r, _ = O.bls_constants()
    rng = np.random.default_rng(17)
    for log_n in (1, 3, 6):
        n = 1 << log_n
        w = pow(7, (r - 1) >> log_n, r)
        av, a = fr_rand(rng, n, r)
        bv, b = fr_rand(rng, n, r)
        cv = [x * y % r for x, y in zip(av, bv)]
        h = fr_ints(O.groth16_quotient(a, b, to_limbs(cv)))
        assert h[n - 1] == 0
        for z in (5, int.from_bytes(rng.bytes(31), "little")):
            lhs = (interp_eval(av, z, w, r) * interp_eval(bv, z, w, r) - interp_eval(cv, z, w, r)) % r
            assert lhs == sum(hk * pow(z, k, r) for k, hk in enumerate(h)) * (pow(z, n, r) - 1) % r
        # unsatisfied: agreement on the coset only
        cv2, c2 = fr_rand(rng, n, r)
        h2 = fr_ints(O.groth16_quotient(a, b, c2))
        for i in (0, n // 2, n - 1):
            x = 7 * pow(w, i, r) % r
            lhs = (interp_eval(av, x, w, r) * interp_eval(bv, x, w, r) - interp_eval(cv2, x, w, r)) % r
            assert lhs == sum(hk * pow(x, k, r) for k, hk in enumerate(h2)) * (pow(7, n, r) - 1) % r
