"""Oracle NTT / LDE / commit self-consistency (no vectors exist in the reference tree for these:
they are exact integer arithmetic, checked here against the O(n^2) definition)."""
import numpy as np

import oracle_lib as O

P = O.P


def rnd(n, seed=0):
    return O.splitmix64_felts(0x243F6A8885A308D3 + seed, n)


def test_ntt_matches_naive_dft():
    for log_n in range(0, 9):
        x = rnd(1 << log_n, log_n)
        want = np.zeros_like(x)
        O.lib().or_dft_naive(O.ptr(x), O.ptr(want), log_n)
        assert (O.ntt(x) == want).all()
        assert (O.intt(want) == x).all()


def test_ntt_definition_spot_check_python_ints():
    log_n = 6
    x = rnd(1 << log_n, 99)
    w = pow(7, (P - 1) >> log_n, P)
    got = O.ntt(x)
    for i in (0, 1, 5, 63):
        assert int(got[i]) == sum(int(x[j]) * pow(w, i * j, P) for j in range(64)) % P


def test_ntt_roundtrip_and_linearity_large():
    n = 1 << 14
    a, b = rnd(n, 1), rnd(n, 2)
    assert (O.intt(O.ntt(a)) == a).all()
    s = (a.astype(object) + b.astype(object)) % P
    s = np.array(s, dtype=np.uint64)
    fs = (O.ntt(a).astype(object) + O.ntt(b).astype(object)) % P
    assert (O.ntt(s) == np.array(fs, dtype=np.uint64)).all()


def test_coset_lde_evaluates_polynomial_on_coset():
    log_n, rate = 4, 3
    c = rnd(1 << log_n, 5)
    out = O.coset_lde(c, rate, 7)
    N = 1 << (log_n + rate)
    w = pow(7, (P - 1) // N, P)
    for i in (0, 1, 17, N - 1):
        x = 7 * pow(w, i, P) % P
        assert int(out[i]) == sum(int(c[j]) * pow(x, j, P) for j in range(1 << log_n)) % P
    # the LDE restricted to multiples of 2^rate-th cosets reproduces... values: intt(values)=c
    vals = O.ntt(c)
    assert (O.intt(vals) == c).all()


def test_commit_batch_consistency():
    k, log_n, rate, cap_h = 5, 5, 3, 2
    vals = rnd(k << log_n, 7).reshape(k, -1)
    r = O.commit_batch(vals, rate, cap_h, want=("coeffs", "lde", "cap", "digests"))
    N = 1 << (log_n + rate)
    for p in range(k):
        assert (r["coeffs"][p] == O.intt(vals[p])).all()
        assert (O.bit_reverse(r["lde"][p]) == O.coset_lde(r["coeffs"][p], rate, 7)).all()
    # leaf i = row i of the bit-reversed LDE columns
    leaves = np.ascontiguousarray(r["lde"].T)
    cap2, dig2 = O.merkle_tree(leaves, cap_h, want_digests=True)
    assert (cap2 == r["cap"]).all() and (dig2 == r["digests"]).all()
    assert (r["digests"][0] == O.hash_no_pad(leaves[0])).all()
    assert dig2.shape[0] == 2 * N - (2 << cap_h)


def test_merkle_small_leaves_are_not_hashed():
    leaves = rnd(16 * 3, 11).reshape(16, 3)
    cap, dig = O.merkle_tree(leaves, 0, want_digests=True)
    assert (dig[0][:3] == leaves[0]).all() and dig[0][3] == 0
    assert (dig[16] == O.two_to_one(dig[0], dig[1])).all()
