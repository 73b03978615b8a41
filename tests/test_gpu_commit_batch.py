"""GPU parity for the product-shape fast paths: 4096-point natural-order transforms, the LDE as
2^rate coset NTTs, and batched commitments (several proofs' oracles in one call)."""
import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu
SEED = 0x243F6A8885A308D3


@pytest.fixture(scope="module")
def prover():
    import cityprover
    p = cityprover.Prover(0)
    yield p
    p.close()


def felts(n, seed=0):
    return O.splitmix64_felts(SEED + seed, n)


@pytest.mark.parametrize("k,trees,rate,cap_h", [(135, 1, 3, 4), (20, 3, 3, 4), (16, 5, 3, 4), (7, 2, 1, 0),
                                                (3, 2, 3, 15)])
def test_commit_batch_matches_oracle_per_tree(prover, k, trees, rate, cap_h):
    log_n, n = 12, 4096
    N = n << rate
    vals = felts(k * trees * n, k + trees).reshape(trees * k, n)
    dv = prover.to_device(vals)
    dl, dco = prover.alloc(trees * k * N), prover.alloc(trees * k * n)
    per_tree = (2 * N - (2 << cap_h)) if N > (1 << cap_h) else N
    dd, dcap = prover.alloc(trees * per_tree * 4), prover.alloc(trees * (4 << cap_h))
    prover.commit_batch_dev(dv.ptr, k, trees, log_n, rate, cap_h, dl.ptr, dcap.ptr, dco.ptr, dd.ptr)
    lde = dl.download().reshape(trees, k, N)
    co = dco.download().reshape(trees, k, n)
    dig = dd.download().reshape(trees, per_tree, 4)
    caps = dcap.download().reshape(trees, 1 << cap_h, 4)
    for b in (dv, dl, dco, dd, dcap):
        b.free()
    O.lib().or_set_threads(8)
    for t in range(trees):
        want = O.commit_batch(vals[t * k:(t + 1) * k], rate, cap_h, want=("coeffs", "lde", "cap", "digests"))
        assert (co[t] == want["coeffs"]).all()
        assert (lde[t] == want["lde"]).all()
        assert (caps[t] == want["cap"]).all()
        if want["digests"].shape[0]:
            assert (dig[t] == want["digests"]).all()
    O.lib().or_set_threads(1)


def test_4096_natural_order_paths(prover):
    import cityprover as cp
    x = felts(4096 * 3, 9).reshape(3, 4096)
    f = prover.ntt(x)
    i = prover.intt(x)
    c = prover.ntt(x, flags=cp.NTT_COSET, shift=7)
    for b in range(3):
        assert (f[b] == O.ntt(x[b])).all()
        assert (i[b] == O.intt(x[b])).all()
        assert (c[b] == O.coset_lde(x[b], 0, 7)).all()


@pytest.mark.parametrize("rate", [0, 1, 3, 4])
def test_lde_4096_as_coset_ntts(prover, rate):
    c = felts(4096 * 2, 30 + rate).reshape(2, 4096)
    got = prover.lde(c, rate, shift=7, bitrev=True)
    for b in range(2):
        assert (got[b] == O.bit_reverse(O.coset_lde(c[b], rate, 7))).all()
    got3 = prover.lde(c[0], rate, shift=3, bitrev=True)
    assert (got3 == O.bit_reverse(O.coset_lde(c[0], rate, 3))).all()
