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
