"""GPU parity of the BLS12-381 G1 MSM (cp_msm_bls12381_g1) against the oracle's by-definition
sum of scalar multiples, plus size-independent properties at sizes the oracle does not reach."""
import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def prover():
    import cityprover
    p = cityprover.Prover(0)
    yield p
    p.close()


def limbs(v, n):
    return [(int(v) >> (64 * i)) & (2**64 - 1) for i in range(n)]


def make_points(count, seed):
    """count distinct multiples of the generator, via the oracle (affine canonical limbs)"""
    _, r, G = O.bls_constants()
    rng = np.random.default_rng(seed)
    ks = [int.from_bytes(rng.bytes(32), "little") % r for _ in range(count)]
    pts = [O.bls_g1_mul(G, k) for k in ks]
    return ks, pts, np.array([limbs(P[0], 6) + limbs(P[1], 6) for P in pts], dtype=np.uint64)


@pytest.mark.parametrize("n", [1, 2, 3, 31, 100, 1000])
def test_msm_matches_oracle(prover, n):
    import cityprover as cp
    _, r, G = O.bls_constants()
    rng = np.random.default_rng(n)
    base_k, pts, xy = make_points(min(n, 24), n)
    idx = rng.integers(0, len(pts), n)
    xy_n = xy[idx]
    scal = [int.from_bytes(rng.bytes(32), "little") for _ in range(n)]     # full 256-bit scalars (not reduced mod r)
    edge = [0, 1, 2, r - 1, r, r + 1, 2**256 - 1, 2**255, 65535, 65536]
    for i, e in enumerate(edge[:n]):
        scal[i] = e
    sc = np.array([limbs(k, 4) for k in scal], dtype=np.uint64)
    O.lib().or_set_threads(8)
    try:
        want = O.bls_g1_msm(sc, xy_n)
    finally:
        O.lib().or_set_threads(1)
    assert cp.msm_g1(prover, sc, xy_n) == want
    # the answer is also (sum k_i * a_i) * G for points a_i * G
    assert want == O.bls_g1_mul(G, sum(k * base_k[j] for k, j in zip(scal, idx)) % r)


def test_msm_special_inputs(prover):
    import cityprover as cp
    p, r, G = O.bls_constants()
    g = np.array([limbs(G[0], 6) + limbs(G[1], 6)], dtype=np.uint64)
    neg = np.array([limbs(G[0], 6) + limbs(p - G[1], 6)], dtype=np.uint64)
    one = np.array([limbs(1, 4)], dtype=np.uint64)
    assert cp.msm_g1(prover, np.zeros((0, 4), np.uint64), np.zeros((0, 12), np.uint64)) is None
    assert cp.msm_g1(prover, one, g) == G
    assert cp.msm_g1(prover, np.array([limbs(0, 4)], dtype=np.uint64), g) is None
    assert cp.msm_g1(prover, np.array([limbs(r, 4)], dtype=np.uint64), g) is None
    # identical points land in the same bucket (doubling branch); a point and its negative cancel
    n = 300
    assert cp.msm_g1(prover, np.repeat(one, n, 0), np.repeat(g, n, 0)) == O.bls_g1_mul(G, n)
    both = np.concatenate([np.repeat(g, 5, 0), np.repeat(neg, 5, 0)])
    assert cp.msm_g1(prover, np.repeat(one, 10, 0), both) is None
    # infinity flags
    flags = np.zeros(10, np.uint8)
    flags[5:] = 1
    assert cp.msm_g1(prover, np.repeat(one, 10, 0), both, flags) == O.bls_g1_mul(G, 5)
    # a coordinate >= p is refused
    bad = g.copy()
    bad[0, :6] = limbs(p, 6)
    with pytest.raises(cp.CityProverError, match="canonical"):
        cp.msm_g1(prover, one, bad)


def test_msm_large_properties(prover):
    """2^16 points (beyond the oracle's reach): points drawn from 64 known multiples a_j * G, so that
    MSM = (sum k_i a_j(i)) * G; and linearity MSM(k + k') = MSM(k) + MSM(k')."""
    import cityprover as cp
    _, r, G = O.bls_constants()
    n = 1 << 16
    rng = np.random.default_rng(99)
    base_k, pts, xy = make_points(64, 7)
    idx = rng.integers(0, 64, n)
    xy_n = np.ascontiguousarray(xy[idx])
    k1 = rng.integers(0, 2**63, (n, 4), dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, (n, 4), dtype=np.uint64)
    k1[:, 3] >>= np.uint64(2)     # keep k1 + k2 below 2^256
    k2 = rng.integers(0, 2**62, (n, 4), dtype=np.uint64)
    to_int = lambda a: [sum(int(a[i, j]) << (64 * j) for j in range(4)) for i in range(a.shape[0])]
    i1, i2 = to_int(k1), to_int(k2)
    base = np.array(base_k, dtype=object)
    m1 = cp.msm_g1(prover, k1, xy_n)
    assert m1 == O.bls_g1_mul(G, int(sum(k * base_k[j] for k, j in zip(i1, idx)) % r))
    ksum = np.array([limbs(a + b, 4) for a, b in zip(i1, i2)], dtype=np.uint64)
    m2 = cp.msm_g1(prover, k2, xy_n)
    assert cp.msm_g1(prover, ksum, xy_n) == O.bls_g1_add(m1, m2)
    # device-resident point set
    P = cp.G1Points(prover, xy_n)
    ds = prover.to_device(k1)
    assert P.msm_dev(ds.ptr) == m1
    ds.free()
    P.free()


@pytest.mark.parametrize("log_n,c", [(9, 5), (10, 8), (13, 13), (16, 16)])
def test_msm_signed_digit_boundaries(prover, log_n, c):
    """The windows are recoded to signed digits (|d| <= 2^(c-1), carry into the next window): scalars whose c-bit digits
    sit on the boundary (2^(c-1) - 1, 2^(c-1), 2^(c-1) + 1, 2^c - 1, 0) in every window, full 256-bit scalars
    included — long carry chains up to the extra top window. Closed form: points (a i + b) G."""
    import cityprover as cp
    _, r, G = O.bls_constants()
    n = 1 << log_n
    half = 1 << (c - 1)
    rng = np.random.default_rng(1000 + c)
    choices = [0, 1, half - 1, half, half + 1, (1 << c) - 1, (1 << c) - 2]
    windows = (256 + c - 1) // c
    ks = []
    for i in range(n):
        digs = rng.choice(choices, windows)
        if i % 7 == 0:
            digs[:] = (1 << c) - 1          # 2^256 - 1 after truncation: carries through every window
        if i % 7 == 1:
            digs[:] = half
        ks.append(sum(int(d) << (c * w) for w, d in enumerate(digs)) % (1 << 256))
    k = np.array([limbs(v, 4) for v in ks], dtype=np.uint64)
    P = cp.G1Points.synthetic(prover, G, 3, 1, n)
    ds = prover.to_device(k)
    got = P.msm_dev(ds.ptr)
    ds.free()
    P.free()
    assert got == O.bls_g1_mul(G, sum(v * (3 * i + 1) for i, v in enumerate(ks)) % r)


@pytest.mark.parametrize("n", [511, 513, 1023, 1025, 4097, 8191, 8193, 65535, 65537, 100003, 262143, 262144, 300007])
def test_msm_sizes_around_window_changes(prover, n):
    """Point counts that are not powers of two, on both sides of every change of window width (c = 5 / 8 / 13 / 16),
    and of the switch to the LDS-privatised digit sort (n >= 2^18); full 256-bit scalars; closed form
    (sum k_i (a i + b)) G for points (a i + b) G."""
    import cityprover as cp
    _, r, G = O.bls_constants()
    rng = np.random.default_rng(n)
    k = rng.integers(0, 2**64, (n, 4), dtype=np.uint64)
    k[: n // 2, 3] >>= np.uint64(1)                      # half below 2^255, half anywhere below 2^256
    P = cp.G1Points.synthetic(prover, G, 11, 5, n)
    ds = prover.to_device(k)
    got = P.msm_dev(ds.ptr)
    ds.free()
    P.free()
    ks = [sum(int(k[i, j]) << (64 * j) for j in range(4)) for i in range(n)]
    assert got == O.bls_g1_mul(G, sum(v * (11 * i + 5) for i, v in enumerate(ks)) % r)


def test_msm_large_point_set_tiled_sort(prover):
    """The LDS-privatised digit sort (tiles of scalars per window) with many tiles: 2^21 + 777 points, a third of the
    scalars witness-like (0 / 1 / small), the rest full 256-bit; closed form as above."""
    import cityprover as cp
    _, r, G = O.bls_constants()
    n = (1 << 21) + 777
    rng = np.random.default_rng(21)
    k = rng.integers(0, 2**64, (n, 4), dtype=np.uint64)
    small = rng.random(n) < 0.33
    k[small, 1:] = 0
    k[small, 0] = rng.choice(np.array([0, 1, 1, 2, 65535, 65536], dtype=np.uint64), int(small.sum()))
    P = cp.G1Points.synthetic(prover, G, 3, 7, n)
    ds = prover.to_device(k)
    got = P.msm_dev(ds.ptr)
    ds.free()
    P.free()
    idx = np.arange(n, dtype=object) * 3 + 7
    total = 0
    for j in range(4):
        total += int((k[:, j].astype(object) * idx).sum()) << (64 * j)
    assert got == O.bls_g1_mul(G, total % r)


def test_msm_skewed_scalars(prover):
    """Witness-like scalars: half are 0 or 1, the rest tiny — one bucket receives a large share of the points (the
    workgroup path for heavy buckets), the upper windows are empty."""
    import cityprover as cp
    _, r, G = O.bls_constants()
    n = 1 << 14
    P = cp.G1Points.synthetic(prover, G, 5, 2, n)
    rng = np.random.default_rng(11)
    vals = rng.choice([0, 1, 1, 1, 2, 3, 255, 65537], n)
    k = np.zeros((n, 4), np.uint64)
    k[:, 0] = vals.astype(np.uint64)
    ds = prover.to_device(k)
    got = P.msm_dev(ds.ptr)
    ds.free()
    P.free()
    assert got == O.bls_g1_mul(G, int(sum(int(v) * (5 * i + 2) for i, v in enumerate(vals)) % r))


# ---- G2 ------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n", [1, 5, 200])
def test_g2_msm_matches_oracle(prover, n):
    import cityprover as cp
    _, r, _ = O.bls_constants()
    G2 = O.bls_g2_generator()
    rng = np.random.default_rng(100 + n)
    base_k = [int.from_bytes(rng.bytes(32), "little") % r for _ in range(min(n, 12))]
    pts = [O.bls_g2_mul(G2, k) for k in base_k]
    xy = np.array([O.bls_point2(P)[0] for P in pts], dtype=np.uint64)
    idx = rng.integers(0, len(pts), n)
    scal = [int.from_bytes(rng.bytes(32), "little") for _ in range(n)]
    for i, e in enumerate([0, 1, r - 1, r, 2**256 - 1][:n]):
        scal[i] = e
    sc = np.array([limbs(k, 4) for k in scal], dtype=np.uint64)
    O.lib().or_set_threads(8)
    try:
        want = O.bls_g2_msm(sc, xy[idx])
    finally:
        O.lib().or_set_threads(1)
    assert cp.msm_g2(prover, sc, xy[idx]) == want
    assert want == O.bls_g2_mul(G2, sum(k * base_k[j] for k, j in zip(scal, idx)) % r)


def test_g2_msm_large_closed_form(prover):
    import cityprover as cp
    _, r, _ = O.bls_constants()
    G2 = O.bls_g2_generator()
    n = 1 << 14
    P = cp.G2Points.synthetic(prover, G2, 3, 11, n)
    rng = np.random.default_rng(21)
    k = rng.integers(0, 2**64, (n, 4), dtype=np.uint64)
    k[:, 3] >>= np.uint64(1)
    k[:100] = 0
    k[:100, 0] = 1          # a heavy bucket
    ds = prover.to_device(k)
    got = P.msm_dev(ds.ptr)
    ds.free()
    P.free()
    ks = [sum(int(k[i, j]) << (64 * j) for j in range(4)) for i in range(n)]
    assert got == O.bls_g2_mul(G2, sum(kk * (3 * i + 11) for i, kk in enumerate(ks)) % r)
    g = np.array([O.bls_point2(G2)[0]], dtype=np.uint64)
    neg = g.copy()
    p = O.bls_constants()[0]
    neg[0, 12:18] = limbs(p - G2[1][0], 6)
    neg[0, 18:24] = limbs(p - G2[1][1], 6)
    one = np.array([limbs(1, 4)], dtype=np.uint64)
    assert cp.msm_g2(prover, np.repeat(one, 2, 0), np.concatenate([g, neg])) is None
    assert cp.msm_g2(prover, np.repeat(one, 9, 0), np.repeat(g, 9, 0)) == O.bls_g2_mul(G2, 9)
