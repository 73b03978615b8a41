"""The device arithmetic (city-rollup_amd/csrc/*.h, __host__ __device__) instantiated on the host
and checked against the oracle: lazy reductions, limb-plane MDS, full permutation. CPU only."""
import ctypes
import os
import sys

import numpy as np
import pytest

import oracle_lib as O

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "hostsim"))
P = O.P


@pytest.fixture(scope="module")
def hs():
    import build as hb
    lib = ctypes.CDLL(hb.build())
    u64 = ctypes.c_uint64
    for f in ("hs_mul", "hs_mul_lazy", "hs_add", "hs_sub"):
        getattr(lib, f).restype = u64
        getattr(lib, f).argtypes = [u64, u64]
    lib.hs_poseidon_permute.argtypes = [ctypes.POINTER(u64), ctypes.c_size_t]
    lib.hs_mds_limb.argtypes = [ctypes.POINTER(ctypes.c_uint32)] * 2
    return lib


EDGE = [0, 1, 2, P - 1, P - 2, P, P + 1, 2**64 - 1, 2**64 - 2**32, 0xFFFFFFFF, 0x100000000,
        0xFFFFFFFF00000000, 1 << 63, 0xFFFFFFFEFFFFFFFF]


def test_field_ops(hs):
    rng = np.random.default_rng(0)
    vals = [int(x) for x in rng.integers(0, 2**64, 300, dtype=np.uint64)] + EDGE
    for a in vals[:80] + EDGE:
        for b in vals[-40:]:
            assert hs.hs_mul_lazy(a, b) % P == (a * b) % P  # lazy: any u64 in, congruent u64 out
            ac, bc = a % P, b % P
            assert hs.hs_mul(ac, bc) == (ac * bc) % P
            assert hs.hs_add(ac, bc) == (ac + bc) % P
            assert hs.hs_sub(ac, bc) == (ac - bc) % P


def test_mds_limb_plane(hs):
    C = [17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20]
    rng = np.random.default_rng(1)
    for t in range(200):
        s = rng.integers(0, 1 << 22, 12, dtype=np.uint32)
        if t == 0:
            s[:] = (1 << 22) - 1
        y = np.zeros(12, np.uint32)
        hs.hs_mds_limb(s.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)),
                       y.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)))
        want = [sum(C[i] * int(s[(i + r) % 12]) for i in range(12)) + (8 * int(s[0]) if r == 0 else 0)
                for r in range(12)]
        assert [int(v) for v in y] == want


def test_permutation_matches_oracle(hs):
    st = O.splitmix64_felts(42, 12 * 300).reshape(-1, 12)
    st[0] = 0
    st[1] = P - 1
    st[2] = np.array([0, 1, 2, P - 1, P - 2, 0xFFFFFFFF, 0x100000000, 0xFFFFFFFF00000000,
                      0xFFFFFFFE00000001, 1 << 63, (1 << 63) + 1, 0x7FFFFFFF80000000], np.uint64)
    got = st.copy()
    hs.hs_poseidon_permute(got.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), got.shape[0])
    assert (got == O.permute_many(st).reshape(-1, 12)).all()


def test_mul_pow2_shifts(hs):
    hs.hs_mul_pow2.restype = ctypes.c_uint64
    hs.hs_mul_pow2.argtypes = [ctypes.c_uint64, ctypes.c_int]
    rng = np.random.default_rng(2)
    vals = [int(x) % P for x in rng.integers(0, 2**64, 200, dtype=np.uint64)] + [0, 1, P - 1, P - 2, 0xFFFFFFFF,
                                                                                 0xFFFFFFFF00000000, 1 << 63]
    for k in (0, 1, 5, 12, 24, 31, 32, 33, 36, 48, 60, 63, 64, 65, 72, 84, 95):
        for x in vals:
            assert hs.hs_mul_pow2(x, k) == (x << k) % P, (x, k)


def test_radix16_register_butterfly_is_a_16_point_dif_ntt(hs):
    hs.hs_round16.argtypes = [ctypes.POINTER(ctypes.c_uint64), ctypes.c_int]
    x = O.splitmix64_felts(5, 16)
    x[0], x[1], x[2] = P - 1, 0, 1
    got = x.copy()
    hs.hs_round16(got.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), 0)
    assert (got == O.bit_reverse(O.ntt(x))).all()
    # inverse direction: same network with omega^-1 (no 1/n scaling)
    got = x.copy()
    hs.hs_round16(got.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), 1)
    inv16 = pow(16, P - 2, P)
    want = O.bit_reverse(O.intt(x))
    assert [int(v) * inv16 % P for v in got] == [int(v) for v in want]


def test_plane_resident_permute_equals_textbook(hs):
    """`permute` (partial rounds resident in limb planes, constants pushed forward) == `permute_textbook` == oracle, on
    random states and on states full of boundary words."""
    hs.hs_poseidon_permute_textbook.argtypes = [ctypes.POINTER(ctypes.c_uint64), ctypes.c_size_t]
    rng = np.random.default_rng(9)
    st = O.splitmix64_felts(7, 12 * 400).reshape(-1, 12)
    edge = np.array([0, 1, 2, P - 1, P - 2, 0xFFFFFFFF, 0x100000000, 0xFFFFFFFF00000000, 0xFFFFFFFE00000001,
                     1 << 63, 0x3FFFFF, 0x400000, 0xFFFFF00000000000, 0xFFFFEFFFFFFFFFFF], np.uint64)
    for i in range(200):   # rows built only from boundary words
        st[i] = rng.choice(edge, 12)
    a, b = st.copy(), st.copy()
    p64 = ctypes.POINTER(ctypes.c_uint64)
    hs.hs_poseidon_permute(a.ctypes.data_as(p64), a.shape[0])
    hs.hs_poseidon_permute_textbook(b.ctypes.data_as(p64), b.shape[0])
    assert (a == b).all()
    assert (a == O.permute_many(st).reshape(-1, 12)).all()


def test_double_precision_carry_normalisation(hs):
    """renorm_d (poseidon.h): a value l + 2^32 h on two double-precision limbs — integers or quarter-integers up to 2^51 —
    keeps its residue mod p exactly and comes back with both limbs within 2^31 + 2^20 of zero."""
    from fractions import Fraction
    pd = ctypes.POINTER(ctypes.c_double)
    hs.hs_renorm_d.argtypes = [pd, pd]
    rng = np.random.default_rng(3)
    B = 1 << 51
    cases = [(0, 0), (B - 1, B - 1), (-B + 1, -B + 1), (B - 1, -B + 1), (1, 0), (0, 1), (-1, 0), (0, -1), ((1 << 32), 0), (0, 1 << 32),
             ((1 << 31), (1 << 31)), (-(1 << 31), -(1 << 31))]
    cases += [tuple(int(v) for v in rng.integers(-B, B, 2)) for _ in range(3000)]
    for l, h in cases:
        for q in (0, 1, 2, 3):                      # the fractional part the low limb carries: quarters
            lf = Fraction(4 * l + q, 4)
            if abs(lf) >= B:
                continue
            arr = (ctypes.c_double * 2)(float(lf), float(h))
            out = (ctypes.c_double * 2)()
            hs.hs_renorm_d(arr, out)
            ol, oh = Fraction(out[0]), Fraction(out[1])
            assert abs(ol) <= (1 << 31) + (1 << 20) and abs(oh) <= (1 << 31) + (1 << 20), (l, h, q, out[0], out[1])
            d = (ol + (1 << 32) * oh) - (lf + (1 << 32) * h)
            assert d.denominator == 1 and d.numerator % P == 0, (l, h, q)


def test_double_precision_layer_identities(hs):
    """T^-1' . K . T is the MDS without its diagonal, T . T^-1' scales by (4, 4, 2) and the scaled products take the state
    divided by (4, 4, 2): what the transformed-domain partial rounds rest on. Then one whole layer: the double-precision
    planes against the integer planes on lazy u64 states."""
    pd = ctypes.POINTER(ctypes.c_double)
    hs.hs_dom_d.argtypes = [ctypes.c_int, pd, pd]
    C = [17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20]
    rng = np.random.default_rng(4)

    def dom(op, v):
        a = (ctypes.c_double * 12)(*[float(x) for x in v])
        b = (ctypes.c_double * 12)()
        hs.hs_dom_d(op, a, b)
        return [x for x in b]
    for _ in range(200):
        s = [int(x) for x in rng.integers(-(1 << 40), 1 << 40, 12)]
        u = dom(0, s)
        o = dom(1, u)
        y = dom(3, o)
        assert y == [float(sum(C[i] * s[(i + r) % 12] for i in range(12))) for r in range(12)]
        assert dom(0, y) == [4.0 * v for v in o[:6]] + [2.0 * v for v in o[6:]]
        assert y[0] == o[0] + o[3] + o[6]
        w = [v / 4.0 for v in u[:6]] + [v / 2.0 for v in u[6:]]           # the stored form: quarter / half integers, exact
        assert dom(2, w) == o
    p64 = ctypes.POINTER(ctypes.c_uint64)
    hs.hs_mds_layer.argtypes = [p64, ctypes.c_int]
    M = (1 << 64) - 1
    states = [[0] * 12, [M] * 12, [P - 1] * 12, [M, 0] * 6] + [[int(x) for x in rng.integers(0, 1 << 64, 12, dtype=np.uint64)] for _ in range(300)]
    for st in states:
        a, b = (ctypes.c_uint64 * 12)(*st), (ctypes.c_uint64 * 12)(*st)
        hs.hs_mds_layer(a, 0)
        hs.hs_mds_layer(b, 1)
        want = [(sum(C[i] * st[(i + r) % 12] for i in range(12)) + (8 * st[0] if r == 0 else 0)) % P for r in range(12)]
        assert list(a) == want and list(b) == want


def test_double_precision_magnitudes(hs):
    """Worst-case magnitudes of the double-precision layers (poseidon.h `permute_until`, `mds_layer_d`) by interval
    arithmetic: between two carry normalisations every limb — with its two fractional bits — stays inside the 53 bits a
    double holds exactly, and inside the 2^51 the limb -> integer conversion of `recombine_d` is good for."""
    hs.hs_recombine_d.restype = ctypes.c_uint64
    hs.hs_recombine_d.argtypes = [ctypes.c_double, ctypes.c_double, ctypes.c_uint64]
    lim = (1 << 31) + (1 << 20)                    # a normalised limb (test above)
    n = 1 << 32                                    # a limb of an S-box output
    # The partial rounds go two per trip (poseidon.h partial_rounds): W = (E, F, v) at the top of a trip is the output of the first
    # layer (entry from the full rounds: limbs < 2^32, T sums four, the unscaled products multiply by at most 64) or of a SQUARED scaled
    # layer; element 0 is read off it (three entries), the change goes in with a factor <= 1/2, W is normalised, element 0 of the next
    # layer is a 12-term dot product with l K (sum of |coefficients| 350), and the squared layer (rows sum to at most 2^16: aa =
    # 4096 (5 J + P^2)) plus the second change times K t (entries <= 32) gives the next W.
    cz = 64 + 64 + 128 + 4 + 8 + 32 + 4 + 2 + 2 + 2 + 32 + 8
    k2 = max(4096 * 16, (4 + 8 + 32) ** 2, (2 * (2 + 4 + 16 + 1 + 1 + 1)) ** 2)
    kt = 32
    assert cz == 350 and k2 == 1 << 16
    x = 64 * 4 * n
    for trip in range(11):                         # x bounds the limbs of W at the top of the trip (integers)
        z = 3 * x + 8 * n                          # element 0 + the diagonal
        assert z < (1 << 51) - (1 << 32)           # recombine_d
        w = x + (n + z)                            # + (new - z) / 4, / 2 (bounded by |new - z|)
        assert w * 4 < 1 << 53                     # two fractional bits
        z2 = cz * lim + 8 * n                      # one layer ahead, from the normalised W
        assert z2 < (1 << 51) - (1 << 32)
        x = k2 * lim + kt * (n + z2)               # the squared layer + the second change one layer on: integers again
        assert x < 1 << 53
    assert 4 * x + 8 * n < (1 << 51) - (1 << 32)   # leaving: natural limbs (T^-1' adds four products) + diagonal
    # the conversion itself at its limits, against exact integers
    for l, h, c in ((0, 0, 0), ((1 << 51) - (1 << 32) - 1, (1 << 51) - (1 << 32) - 1, P - 1), (-(1 << 51) + 1, -(1 << 51) + 1, 5), (-1, 0, 0), (0, -1, 0),
                    (123456789012345, -98765432109876, 0xFFFFFFFF00000000)):
        got = hs.hs_recombine_d(float(l), float(h), c)
        assert got % P == (l + (h << 32) + c + ((0x433 << 52) + (1 << 51)) * (1 + (1 << 32))) % P, (l, h, c)   # c is stored minus that offset


# ---- BLS12-381 device formulas (csrc/bls12_381.h) on the host ------------------------------------------------
def _w32(v, n=12):
    return (ctypes.c_uint32 * n)(*[(int(v) >> (32 * i)) & 0xFFFFFFFF for i in range(n)])


def _from_w32(w):
    return sum(int(x) << (32 * i) for i, x in enumerate(w))


def test_bls_field_ops_match_python_integers(hs):
    p, r, G = O.bls_constants()
    rng = np.random.default_rng(4)
    vals = [0, 1, 2, p - 1, p - 2, (1 << 380) + 12345, G[0], G[1]] + [int.from_bytes(rng.bytes(48), "little") % p for _ in range(40)]
    out = (ctypes.c_uint32 * 12)()
    for i in range(len(vals) - 1):
        a, b = vals[i], vals[i + 1]
        for op, want in ((0, a * b % p), (1, (a + b) % p), (2, (a - b) % p)):
            hs.hs_bls_fp_op(op, _w32(a), _w32(b), out)
            assert _from_w32(out) == want, (op, hex(a), hex(b))
    for a in vals[1:12]:
        hs.hs_bls_fp_op(3, _w32(a), _w32(0), out)
        assert _from_w32(out) * a % p == 1
    for a in vals:      # the dedicated Montgomery square (doubled cross products)
        hs.hs_bls_fp_op(4, _w32(a), _w32(0), out)
        assert _from_w32(out) == a * a % p, hex(a)


def _limbs28(v):
    return (ctypes.c_uint32 * 14)(*[(int(v) >> (28 * i)) & ((1 << 28) - 1) for i in range(14)])


def _from_limbs28(w):
    return sum(int(x) << (28 * i) for i, x in enumerate(w))


def test_bls_loose_field_primitives(hs):
    """The bounded-but-unreduced arithmetic of the bucket accumulation (bls12_381.h "loose arithmetic"): each primitive
    returns normalised limbs, the value it promises (an exact integer for add / sub / weak, a residue below 2p for the
    product), for operands anywhere below 32 p — and for the product also with limbs up to 2^29."""
    p, r, G = O.bls_constants()
    R = 1 << 392
    rng = np.random.default_rng(14)
    out = (ctypes.c_uint32 * 14)()
    M28 = (1 << 28) - 1

    def call(op, a, b=0):
        hs.hs_bls_lz_op(op, _limbs28(a) if isinstance(a, int) else a, _limbs28(b) if isinstance(b, int) else b, out)
        assert all(x <= M28 for x in out), "limbs not normalised"
        return _from_limbs28(out)
    vals = [0, 1, p - 1, p, p + 1, 2 * p - 1, 8 * p, 16 * p - 1, 16 * p, 22 * p - 1, 31 * p, 32 * p - 1]
    vals += [int.from_bytes(rng.bytes(49), "little") % (32 * p) for _ in range(40)]
    Rinv = pow(R, -1, p)
    for a in vals:
        for b in vals[::3]:
            if a * b < R * p:
                got = call(0, a, b)
                assert got < 2 * p and got % p == a * b * Rinv % p, (hex(a), hex(b))
            if a + b < R:
                assert call(1, a, b) == a + b
            if b <= 8 * p:
                assert call(2, a, b) == a + 8 * p - b
        assert call(3, a) == (a - 16 * p if a >= 16 * p else a)
        assert call(4, a) == a % p                                  # a R / R: the canonical value of a
        if a % p == 0:
            assert hs.hs_bls_lz_maybe_zero(_limbs28(a)) == 1
    # the zero filter: every multiple of p below 32 p passes; random values pass with probability 2^-23
    assert all(hs.hs_bls_lz_maybe_zero(_limbs28(k * p)) == 1 for k in range(32))
    assert sum(hs.hs_bls_lz_maybe_zero(_limbs28(int.from_bytes(rng.bytes(49), "little") % (32 * p))) for _ in range(2000)) <= 1
    # product operands with unnormalised limbs (< 2^29: a carry-free sum of two normalised values)
    for _ in range(50):
        x = [int(v) for v in rng.integers(0, 1 << 28, 14)]
        y = [int(v) for v in rng.integers(0, 1 << 28, 14)]
        z = [int(v) for v in rng.integers(0, 1 << 28, 14)]
        x[13] = y[13] = z[13] = 0x1000            # keep the values below a few p
        s = (ctypes.c_uint32 * 14)(*[u + v for u, v in zip(x, y)])
        zz = (ctypes.c_uint32 * 14)(*z)
        sv, zv = _from_limbs28(s), _from_limbs28(zz)
        if sv * zv < R * p:
            got = call(0, s, zz)
            assert got < 2 * p and got % p == sv * zv * Rinv % p


def test_bls_loose_bucket_accumulation(hs):
    """xyzz_add_mixed_loose chained like k_bucket_sum does, G1 and G2: sums of random points, runs that hit the doubling
    branch (the same point twice in a row, and a point equal to the running sum), and cancellation to infinity and on."""
    p, r, G = O.bls_constants()
    G2 = O.bls_g2_generator()
    rng = np.random.default_rng(15)

    def flat1(P):
        return list(_w32(P[0])) + list(_w32(P[1]))

    def flat2(P):
        return list(_w32(P[0][0])) + list(_w32(P[0][1])) + list(_w32(P[1][0])) + list(_w32(P[1][1]))

    def chain1(pts):
        xy = (ctypes.c_uint32 * (24 * len(pts)))(*[w for P in pts for w in flat1(P)])
        out = (ctypes.c_uint32 * 24)()
        inf = hs.hs_bls_g1_chain(xy, len(pts), out)
        return None if inf else (_from_w32(out[:12]), _from_w32(out[12:]))

    def chain2(pts):
        xy = (ctypes.c_uint32 * (48 * len(pts)))(*[w for P in pts for w in flat2(P)])
        out = (ctypes.c_uint32 * 48)()
        inf = hs.hs_bls_g2_chain(xy, len(pts), out)
        return None if inf else ((_from_w32(out[:12]), _from_w32(out[12:24])), (_from_w32(out[24:36]), _from_w32(out[36:])))

    def total(add, pts):
        acc = None
        for P in pts:
            acc = P if acc is None else add(acc, P)
        return acc
    ks = [int.from_bytes(rng.bytes(32), "little") % r for _ in range(40)]
    g1 = [O.bls_g1_mul(G, k) for k in ks]
    assert chain1(g1) == total(O.bls_g1_add, g1)
    assert chain1(g1[:1]) == g1[0]
    A, B = g1[0], g1[1]
    assert chain1([A, A]) == O.bls_g1_mul(A, 2)                                  # doubling as the second addition
    assert chain1([A, B, O.bls_g1_add(A, B), g1[2]]) == O.bls_g1_add(O.bls_g1_mul(O.bls_g1_add(A, B), 2), g1[2])   # x equal to the running sum's
    negA = (A[0], p - A[1])
    assert chain1([A, negA]) is None                                             # cancellation
    assert chain1([A, negA, B, g1[3]]) == O.bls_g1_add(B, g1[3])                 # ... and on from infinity
    g2 = [O.bls_g2_mul(G2, k) for k in ks[:24]]
    assert chain2(g2) == total(O.bls_g2_add, g2)
    C, D = g2[0], g2[1]
    assert chain2([C, C]) == O.bls_g2_mul(C, 2)
    assert chain2([C, D, O.bls_g2_add(C, D), g2[2]]) == O.bls_g2_add(O.bls_g2_mul(O.bls_g2_add(C, D), 2), g2[2])
    negC = (C[0], ((p - C[1][0]) % p, (p - C[1][1]) % p))
    assert chain2([C, negC]) is None
    assert chain2([C, negC, D, g2[3]]) == O.bls_g2_add(D, g2[3])


def test_bls_group_law_matches_oracle(hs):
    p, r, G = O.bls_constants()
    rng = np.random.default_rng(5)

    def run(op, P, Q, k=0):
        pxy = (ctypes.c_uint32 * 24)(*(list(_w32(P[0])) + list(_w32(P[1])))) if P else (ctypes.c_uint32 * 24)()
        qxy = (ctypes.c_uint32 * 24)(*(list(_w32(Q[0])) + list(_w32(Q[1])))) if Q else (ctypes.c_uint32 * 24)()
        out = (ctypes.c_uint32 * 24)()
        inf = hs.hs_bls_g1_op(op, pxy, 0 if P else 1, qxy, 0 if Q else 1, k, out)
        return None if inf else (_from_w32(out[:12]), _from_w32(out[12:]))

    pts = [O.bls_g1_mul(G, int.from_bytes(rng.bytes(32), "little") % r) for _ in range(5)] + [G]
    neg = lambda P: (P[0], p - P[1])
    for A in pts:
        for B in pts[:3]:
            assert run(0, A, B) == O.bls_g1_add(A, B)
            assert run(1, A, B) == O.bls_g1_add(A, B)
        assert run(0, A, A) == run(1, A, A) == run(2, A, None) == O.bls_g1_mul(A, 2)     # the doubling branch
        assert run(0, A, neg(A)) is None and run(1, A, neg(A)) is None                    # the cancelling branch
        assert run(0, A, None) == A and run(0, None, A) == A and run(1, None, A) == A
        for k in (0, 1, 2, 3, 255, 65535, 40000):
            assert run(3, A, None, k) == O.bls_g1_mul(A, k)
    assert run(2, None, None) is None and run(3, None, None, 7) is None


def test_bls_fp2_and_g2_formulas(hs):
    """The G2 instantiation of the templated group law (coordinates in F_p^2) against the oracle, on the host."""
    p, r, _ = O.bls_constants()
    G2 = O.bls_g2_generator()
    rng = np.random.default_rng(6)

    def w2(c):   # (c0, c1) -> 24 words
        return (ctypes.c_uint32 * 24)(*(list(_w32(c[0])) + list(_w32(c[1]))))

    out = (ctypes.c_uint32 * 24)()
    for _ in range(20):
        a = (int.from_bytes(rng.bytes(48), "little") % p, int.from_bytes(rng.bytes(48), "little") % p)
        b = (int.from_bytes(rng.bytes(48), "little") % p, int.from_bytes(rng.bytes(48), "little") % p)
        hs.hs_bls_fp2_op(0, w2(a), w2(b), out)
        assert (_from_w32(out[:12]), _from_w32(out[12:])) == ((a[0] * b[0] - a[1] * b[1]) % p, (a[0] * b[1] + a[1] * b[0]) % p)
        hs.hs_bls_fp2_op(1, w2(a), w2(a), out)
        assert (_from_w32(out[:12]), _from_w32(out[12:])) == ((a[0] * a[0] - a[1] * a[1]) % p, 2 * a[0] * a[1] % p)
        hs.hs_bls_fp2_op(2, w2(a), w2(a), out)
        inv = (_from_w32(out[:12]), _from_w32(out[12:]))
        assert ((a[0] * inv[0] - a[1] * inv[1]) % p, (a[0] * inv[1] + a[1] * inv[0]) % p) == (1, 0)

    def pt_words(P):
        if P is None:
            return (ctypes.c_uint32 * 48)()
        (x0, x1), (y0, y1) = P
        return (ctypes.c_uint32 * 48)(*(list(_w32(x0)) + list(_w32(x1)) + list(_w32(y0)) + list(_w32(y1))))

    def run(op, P, Q, k=0):
        o = (ctypes.c_uint32 * 48)()
        inf = hs.hs_bls_g2_op(op, pt_words(P), 0 if P else 1, pt_words(Q), 0 if Q else 1, k, o)
        return None if inf else ((_from_w32(o[0:12]), _from_w32(o[12:24])), (_from_w32(o[24:36]), _from_w32(o[36:48])))

    pts = [O.bls_g2_mul(G2, int.from_bytes(rng.bytes(32), "little") % r) for _ in range(3)] + [G2]
    neg = lambda P: (P[0], ((-P[1][0]) % p, (-P[1][1]) % p))
    for A in pts:
        for B in pts[:2]:
            assert run(0, A, B) == run(1, A, B) == O.bls_g2_add(A, B)
        assert run(0, A, A) == run(1, A, A) == run(2, A, None) == O.bls_g2_mul(A, 2)
        assert run(0, A, neg(A)) is None and run(1, A, neg(A)) is None
        assert run(0, A, None) == A and run(0, None, A) == A
        for k in (0, 1, 5, 65535):
            assert run(3, A, None, k) == O.bls_g2_mul(A, k)


def test_bls_scalar_field_ops(hs):
    _, r, _ = O.bls_constants()
    rng = np.random.default_rng(12)
    vals = [0, 1, 2, r - 1, r - 2, (1 << 254) + 99, 7] + [int.from_bytes(rng.bytes(32), "little") % r for _ in range(40)]
    out = (ctypes.c_uint32 * 8)()
    for i in range(len(vals) - 1):
        a, b = vals[i], vals[i + 1]
        for op, want in ((0, a * b % r), (1, (a + b) % r), (2, (a - b) % r)):
            hs.hs_bls_fr_op(op, _w32(a, 8), _w32(b, 8), out)
            assert _from_w32(out) == want, (op, hex(a), hex(b))
    for a in vals[1:10]:
        hs.hs_bls_fr_op(3, _w32(a, 8), _w32(0, 8), out)
        assert _from_w32(out) * a % r == 1
        e = int(rng.integers(0, 2**62))
        hs.hs_bls_fr_op(4, _w32(a, 8), _w32(e, 8), out)
        assert _from_w32(out) == pow(a, e, r)


def test_parallel_for_survives_thread_creation_failures(hs):
    """host_util.h (the transcript helpers, cp_ctx_set_lanes and the Groth16 side chain are built on it): every index
    is done exactly once however many helper threads could be started, and an exception inside a helper reaches the
    caller only after every helper has been joined (never std::terminate)."""
    hs.hs_parallel_for.restype = ctypes.c_long
    hs.hs_parallel_for.argtypes = [ctypes.c_size_t, ctypes.c_size_t, ctypes.c_long, ctypes.POINTER(ctypes.c_uint64)]
    hs.hs_parallel_for_throw.argtypes = [ctypes.c_size_t, ctypes.c_size_t, ctypes.c_size_t]
    for n, threads in ((0, 4), (1, 4), (7, 1), (64, 8), (1000, 16), (33, 40)):
        for fail_after in (-1, 0, 1, 3):
            out = np.zeros(max(n, 1), np.uint64)
            ran = hs.hs_parallel_for(n, threads, fail_after, out.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)))
            assert ran == n
            assert (out[:n] == np.arange(1, n + 1, dtype=np.uint64)).all(), (n, threads, fail_after)
    for bad in (0, 17, 63):
        assert hs.hs_parallel_for_throw(64, 8, bad) == 1
    assert hs.hs_parallel_for_throw(64, 8, 1000) == 0


def _pool_script(hs, n_devices, pool_cap, capacity, ops):
    hs.hs_pool_script.argtypes = [ctypes.c_size_t] * 3 + [ctypes.POINTER(ctypes.c_int64), ctypes.c_size_t, ctypes.POINTER(ctypes.c_int64),
                                                          ctypes.POINTER(ctypes.c_uint64)]
    a = np.array(ops, dtype=np.int64).reshape(-1, 4)
    res = np.zeros(len(a), np.int64)
    cnt = np.zeros(10, np.uint64)
    rc = hs.hs_pool_script(n_devices, pool_cap, capacity, a.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)), len(a),
                           res.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)), cnt.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)))
    assert rc == 0, "the pool handed one buffer out twice"
    keys = ("in_use", "n_malloc", "n_free", "n_oom", "live", "pooled0", "pooled1", "hits0", "misses0", "trims0")
    return [int(v) for v in res], dict(zip(keys, (int(v) for v in cnt)))


def test_batch_buffer_pool_logic(hs):
    """city-rollup_amd/csrc/dev_pool.h over a counting allocator: reuse by exact size, the cap, exported (non-reusable)
    buffers bypass the pool, devices never share buffers, a device index outside the table has no pool, out-of-memory
    empties the pool and tries again (the OOM-flush path), for pool allocations and for every other allocation alike."""
    ALLOC, REL, TRIM, MALLOC = 0, 1, 2, 3
    MB = 1 << 20
    # reuse: the second commitment of a shape gets the first one's buffers back, nothing is freed in between
    res, c = _pool_script(hs, 2, 64 * MB, 1 << 40, [(ALLOC, 0, 8 * MB, 0), (ALLOC, 0, MB, 0), (REL, 0, 0, 1), (REL, 0, 1, 1),
                                                      (ALLOC, 0, 8 * MB, 0), (ALLOC, 0, MB, 0), (ALLOC, 0, 2 * MB, 0)])
    assert res[4] >= 0 and c["n_malloc"] == 3 and c["n_free"] == 0 and c["hits0"] == 2 and c["misses0"] == 3 and c["pooled0"] == 0
    # the cap: what does not fit goes back to the runtime
    res, c = _pool_script(hs, 1, 10 * MB, 1 << 40, [(ALLOC, 0, 8 * MB, 0), (ALLOC, 0, 4 * MB, 0), (REL, 0, 0, 1), (REL, 0, 1, 1)])
    assert c["pooled0"] == 8 * MB and c["n_free"] == 1 and c["in_use"] == 8 * MB
    # cap 0 = pool off
    res, c = _pool_script(hs, 1, 0, 1 << 40, [(ALLOC, 0, MB, 0), (REL, 0, 0, 1), (ALLOC, 0, MB, 0)])
    assert c["hits0"] == 0 and c["n_free"] == 1 and c["n_malloc"] == 2
    # a handle whose pointers were exported is freed through the runtime, never re-issued
    res, c = _pool_script(hs, 1, 64 * MB, 1 << 40, [(ALLOC, 0, MB, 0), (REL, 0, 0, 0), (ALLOC, 0, MB, 0)])
    assert c["pooled0"] == 0 and c["n_free"] == 1 and c["hits0"] == 0
    # two devices: a buffer of device 0 is not handed to device 1; a device outside the table (7) bypasses the pool
    res, c = _pool_script(hs, 2, 64 * MB, 1 << 40, [(ALLOC, 0, MB, 0), (REL, 0, 0, 1), (ALLOC, 1, MB, 0), (ALLOC, 7, MB, 0), (REL, 0, 2, 1),
                                                      (ALLOC, 7, MB, 0)])
    assert c["pooled0"] == MB and c["pooled1"] == 0 and c["n_malloc"] == 4 and c["n_free"] == 1
    # out of memory with buffers parked: the pool is emptied and the allocation succeeds (both entry points); with
    # nothing parked it fails
    script = [(ALLOC, 0, 6 * MB, 0), (ALLOC, 0, 3 * MB, 0), (REL, 0, 0, 1), (REL, 0, 1, 1),   # 9 MB parked, 10 MB capacity
              (ALLOC, 0, 5 * MB, 0),                                                             # miss -> OOM -> trim -> ok
              (MALLOC, 0, 4 * MB, 0),                                                            # fits
              (MALLOC, 0, 4 * MB, 0)]                                                            # OOM, nothing to trim
    res, c = _pool_script(hs, 1, 64 * MB, 10 * MB, script)
    assert res[4] >= 0 and res[5] >= 0 and res[6] == -2
    assert c["trims0"] == 1 and c["n_free"] == 2 and c["pooled0"] == 0 and c["in_use"] == 9 * MB and c["n_oom"] == 2
    res, c = _pool_script(hs, 1, 64 * MB, 10 * MB, [(ALLOC, 0, 6 * MB, 0), (REL, 0, 0, 1), (MALLOC, 0, 8 * MB, 0), (TRIM, 0, 0, 0)])
    assert res[2] >= 0 and res[3] == 0 and c["trims0"] == 1
    # random scripts: the allocator's books always balance (no leak, no double free, no buffer out twice)
    rng = np.random.default_rng(5)
    for _ in range(50):
        ops, live = [], []
        n_alloc = 0
        for _ in range(200):
            r = rng.random()
            if r < 0.5 or not live:
                ops.append((ALLOC if rng.random() < 0.8 else MALLOC, int(rng.integers(0, 3)), int(rng.choice([1, 2, 4, 8])) * MB, 0))
                live.append(n_alloc)
                n_alloc += 1
            elif r < 0.95:
                h = live.pop(int(rng.integers(0, len(live))))
                ops.append((REL, 0, h, int(rng.random() < 0.8)))
            else:
                ops.append((TRIM, int(rng.integers(0, 2)), 0, 0))
        res, c = _pool_script(hs, 2, 16 * MB, 1 << 40, ops)   # capacity ample: every allocation succeeds, handle indices line up
        held = sum(o[2] for o, r_ in zip(ops, res) if o[0] in (ALLOC, MALLOC)) - sum(ops[[i for i, o in enumerate(ops) if o[0] in (ALLOC, MALLOC)][o[2]]][2]
                                                                                 for o in ops if o[0] == REL)
        assert c["in_use"] == held + c["pooled0"] + c["pooled1"]
        assert c["n_free"] != 2**64 - 1
