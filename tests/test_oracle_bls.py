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
