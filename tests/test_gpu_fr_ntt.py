"""GPU parity of the BLS12-381 scalar-field NTT (cp_ntt_bls12381_fr) against the oracle, plus
properties at the full size."""
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


def fr_rand(rng, n, r):
    raw = rng.integers(0, 2**64, (n, 4), dtype=np.uint64)
    raw[:, 3] &= np.uint64((1 << 62) - 1)      # < 2^254 < r
    return raw


@pytest.mark.parametrize("log_n", [0, 1, 2, 5, 9, 10, 11, 13, 16])
def test_fr_ntt_matches_oracle(prover, log_n):
    import cityprover as cp
    _, r, _ = O.bls_constants()
    rng = np.random.default_rng(log_n)
    a = fr_rand(rng, 1 << log_n, r)
    if log_n >= 2:
        a[0] = [(r - 1) >> (64 * j) & (2**64 - 1) for j in range(4)]
        a[1] = 0
    f = cp.fr_ntt(prover, a)
    assert (f == O.fr_ntt(a)).all()
    assert (cp.fr_ntt(prover, f, inverse=True) == a).all()
    assert (cp.fr_ntt(prover, a, inverse=True) == O.fr_ntt(a, inverse=True)).all()
    c = cp.fr_ntt(prover, a, shift=7)
    assert (c == O.fr_ntt(a, shift=7)).all()
    assert (cp.fr_ntt(prover, c, inverse=True, shift=7) == a).all()


def test_fr_ntt_full_size_properties(prover):
    import cityprover as cp
    _, r, _ = O.bls_constants()
    n = 1 << 20
    rng = np.random.default_rng(5)
    a, b = fr_rand(rng, n, r), fr_rand(rng, n, r)
    fa = cp.fr_ntt(prover, a)
    assert (cp.fr_ntt(prover, fa, inverse=True) == a).all()
    assert (fa == O.fr_ntt(a)).all()              # the oracle's O(n log n) transform, same size
    # linearity: NTT(a + b) = NTT(a) + NTT(b) (mod r), checked on a slice of outputs with Python integers
    to_int = lambda m, i: sum(int(m[i, j]) << (64 * j) for j in range(4))
    s = np.array([[((to_int(a, i) + to_int(b, i)) % r >> (64 * j)) & (2**64 - 1) for j in range(4)] for i in range(n)],
                 dtype=np.uint64)
    fs, fb = cp.fr_ntt(prover, s), cp.fr_ntt(prover, b)
    for i in list(range(0, n, n // 64)) + [n - 1]:
        assert to_int(fs, i) == (to_int(fa, i) + to_int(fb, i)) % r


def test_fr_ntt_rejects_bad_input(prover):
    import cityprover as cp
    _, r, _ = O.bls_constants()
    bad = np.array([[(r >> (64 * j)) & (2**64 - 1) for j in range(4)]] * 4, dtype=np.uint64)
    with pytest.raises(cp.CityProverError, match="canonical"):
        cp.fr_ntt(prover, bad)
    ok = np.zeros((4, 4), np.uint64)
    with pytest.raises(cp.CityProverError):
        cp.fr_ntt(prover, ok, shift=0)


# ---- Groth16 quotient (cp_groth16_quotient_bls12381) -------------------------------------------------------------
def fr_mul_rows(a, b, r):
    ints = lambda m: [sum(int(m[i, j]) << (64 * j) for j in range(4)) for i in range(m.shape[0])]
    prod = [x * y % r for x, y in zip(ints(a), ints(b))]
    return np.array([[(v >> (64 * j)) & (2**64 - 1) for j in range(4)] for v in prod], dtype=np.uint64)


@pytest.mark.parametrize("log_n", [1, 4, 10, 13])
def test_groth16_quotient_matches_oracle(prover, log_n):
    import cityprover as cp
    _, r, _ = O.bls_constants()
    rng = np.random.default_rng(100 + log_n)
    n = 1 << log_n
    a, b = fr_rand(rng, n, r), fr_rand(rng, n, r)
    c_sat, c_any = fr_mul_rows(a, b, r), fr_rand(rng, n, r)
    for c in (c_sat, c_any):
        assert (cp.groth16_quotient(prover, a, b, c) == O.groth16_quotient(a, b, c)).all()
    assert not cp.groth16_quotient(prover, a, b, c_sat)[n - 1].any()      # exact division: degree <= n - 2


def test_groth16_quotient_large_identity(prover):
    """2^18 constraints, device-resident: a(z) b(z) - c(z) = h(z) (z^n - 1) at a random z, with the three interpolants
    taken from the (separately tested) inverse NTT and evaluated with Python integers."""
    import cityprover as cp
    _, r, _ = O.bls_constants()
    log_n = 18
    n = 1 << log_n
    rng = np.random.default_rng(77)
    a, b = fr_rand(rng, n, r), fr_rand(rng, n, r)
    # c = a o b on the domain without a Python loop over 2^18 products: square-free trick - take b = 1 on the domain
    # except a sparse set, so that c differs from a only there
    b[:] = 0
    b[:, 0] = 1
    special = rng.integers(0, n, 50)
    b[special] = fr_rand(rng, len(special), r)
    c = a.copy()
    c[special] = fr_mul_rows(a[special], b[special], r)
    da, db, dc = prover.to_device(a), prover.to_device(b), prover.to_device(c)
    cp.groth16_quotient_dev(prover, da.ptr, db.ptr, dc.ptr, log_n)
    h = da.download().reshape(n, 4)
    for d in (da, db, dc):
        d.free()
    assert not h[n - 1].any()
    z = int.from_bytes(rng.bytes(31), "little")
    zs = [1]
    for _ in range(n - 1):
        zs.append(zs[-1] * z % r)
    ints = lambda m: [sum(int(m[i, j]) << (64 * j) for j in range(4)) for i in range(m.shape[0])]
    ev = lambda coef: sum(x * y for x, y in zip(ints(coef), zs)) % r
    ca, cb, cc = (cp.fr_ntt(prover, v, inverse=True) for v in (a, b, c))
    assert (ev(ca) * ev(cb) - ev(cc)) % r == ev(h) * (pow(z, n, r) - 1) % r


def test_groth16_quotient_rejects_bad_input(prover):
    import cityprover as cp
    _, r, _ = O.bls_constants()
    ok = np.zeros((4, 4), np.uint64)
    bad = np.array([[(r >> (64 * j)) & (2**64 - 1) for j in range(4)]] * 4, dtype=np.uint64)
    with pytest.raises(cp.CityProverError, match="canonical"):
        cp.groth16_quotient(prover, ok, bad, ok)
    d = prover.to_device(ok)
    with pytest.raises(cp.CityProverError, match="distinct"):
        cp.groth16_quotient_dev(prover, d.ptr, d.ptr, d.ptr, 2)
    d.free()
