"""GPU parity: the HIP path (through the C ABI) vs the CPU oracle on the same seeded inputs.
Bit-exact is the bar (integer arithmetic mod p); run with `-m gpu` on an MI355X."""
import json
import os

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu

P = O.P
SEED = 0x243F6A8885A308D3


@pytest.fixture(scope="module")
def prover():
    import cityprover
    p = cityprover.Prover(0)
    yield p
    p.close()


def felts(n, seed=0):
    return O.splitmix64_felts(SEED + seed, n)


EDGE = np.array([0, 1, 2, P - 1, P - 2, 0xFFFFFFFF, 0x100000000, 0xFFFFFFFF00000000,
                 0xFFFFFFFE00000001, 1 << 63, (1 << 63) + 1, 0x7FFFFFFF80000000], dtype=np.uint64)


# ---- the field multiplication every kernel shares (csrc/gl.h: carries taken in hand-written multiply-adds) ------------
def test_field_mul_every_carry_and_borrow_corner(prover):
    """gl::mul on the device against Python integers: lazy (non-canonical) operands, every pair of edge values, and operands
    constructed so that each rare path is taken — the carry out of the third multiply-add of the product, the borrow of
    `lo - w3` (needs lo < w3 < 2^32: a 2^-32 event for random inputs, repaired behind a branch), and the wrap of the
    final `+ w2 (2^32 - 1)`."""
    rng = np.random.default_rng(11)
    M = (1 << 64) - 1
    edge = [0, 1, 2, 3, P - 1, P - 2, P, P + 1, M, M - 1, 0xFFFFFFFF, 0x100000000, 0x100000001, 0xFFFFFFFF00000000, 0xFFFFFFFE00000001,
            1 << 63, (1 << 63) + 1, 0x7FFFFFFF80000000, 0x8000000080000000, 0xFFFFFFFFFFFF0000, 1 << 48, 3 << 48, 1 << 32, 1 << 33]
    a = [x for x in edge for _ in edge]
    b = [y for _ in edge for y in edge]
    # borrow of lo - w3 (a b = w3 2^96 + w2 2^64 + lo with lo < w3): lo = 0 from two multiples of 2^32 ...
    for _ in range(2000):
        u, k = int(rng.integers(1 << 16, 1 << 32)), int(rng.integers(1 << 16, 1 << 32))
        a.append(u << 32)
        b.append(k << 32)
    # ... and a small non-zero lo: a odd, b = t / a mod 2^64 makes the low 64 bits of the product equal t
    for _ in range(2000):
        aa = int(rng.integers(1 << 62, 1 << 64, dtype=np.uint64)) | 1
        t = int(rng.integers(1, 1 << 20))
        a.append(aa)
        b.append((t * pow(aa, -1, 1 << 64)) & M)
    # carry out of the cross terms (a1 b0 + a0 b1 + carry >= 2^64) and wrap of the final multiply-add: large halves
    hi_vals = [0xFFFFFFFF, 0xFFFFFFFE, 0x80000000, 0xFFFF0000, 1]
    for a1 in hi_vals:
        for a0 in hi_vals:
            for b1 in hi_vals:
                for b0 in hi_vals:
                    a.append((a1 << 32) | a0)
                    b.append((b1 << 32) | b0)
    ra = rng.integers(0, 1 << 64, 200000, dtype=np.uint64)
    rb = rng.integers(0, 1 << 64, 200000, dtype=np.uint64)
    # random operands with a zero or tiny low product: 2-adic structure makes lo small far more often than 2^-32
    sh = rng.integers(0, 64, 200000)
    rc = (rng.integers(0, 1 << 64, 200000, dtype=np.uint64) >> sh.astype(np.uint64)) << sh.astype(np.uint64)
    A = np.concatenate([np.array(a, dtype=np.uint64), ra, rc])
    B = np.concatenate([np.array(b, dtype=np.uint64), rb, rc[::-1]])
    got = prover.field_mul(A, B)
    want = np.array([(int(x) * int(y)) % P for x, y in zip(A, B)], dtype=np.uint64)
    bad = np.nonzero(got != want)[0]
    assert bad.size == 0, [(hex(int(A[i])), hex(int(B[i])), int(got[i]), int(want[i])) for i in bad[:5]]
    borrows = sum(1 for x, y in zip(A, B) if ((int(x) * int(y)) & M) < ((int(x) * int(y)) >> 96))
    assert borrows > 1000            # the rare path was exercised


# ---- Poseidon ------------------------------------------------------------------------------
def test_poseidon_permute_matches_oracle(prover):
    states = felts(12 * 5000, 1).reshape(-1, 12)
    states[0] = 0
    states[1] = EDGE
    states[2] = P - 1
    got = prover.poseidon_permute(states)
    assert (got == O.permute_many(states).reshape(-1, 12)).all()


def test_poseidon_golden_zero_hashes(prover, golden_dir):
    z = json.load(open(os.path.join(golden_dir, "poseidon_zero_hashes.json")))["two_to_one"]
    cur = np.zeros(4, np.uint64)
    for i in range(1, 40):
        cur = prover.two_to_one(cur, cur)
        assert cur.tolist() == z[i]
    m = json.load(open(os.path.join(golden_dir, "poseidon_zero_hashes.json")))["marked_leaf"]
    assert prover.hash_no_pad(np.array([0] * 8 + [1], np.uint64)).tolist() == m[1]


def test_golden_fingerprint_roots(prover, golden_dir):
    fps = json.load(open(os.path.join(golden_dir, "circuit_fingerprints.json")))
    l = np.array([f["leaf"] for f in fps], np.uint64)
    r = np.array([f["aggregator"] for f in fps], np.uint64)
    want = np.array([f["root"] for f in fps], np.uint64)
    assert (prover.two_to_one(l, r) == want).all()


@pytest.mark.parametrize("length", [0, 1, 3, 4, 5, 7, 8, 9, 15, 16, 17, 20, 85, 135])
def test_hash_no_pad_lengths(prover, length):
    x = felts(64 * max(length, 1), length).reshape(64, -1)[:, :length]
    x = np.ascontiguousarray(x)
    got = prover.hash_no_pad(x) if length else prover.hash_no_pad(np.zeros((64, 0), np.uint64))
    for i in range(0, 64, 7):
        assert (got[i] == O.hash_no_pad(x[i])).all()


# ---- Merkle --------------------------------------------------------------------------------
@pytest.mark.parametrize("n_log,leaf_len,cap_h", [(4, 3, 0), (4, 4, 4), (6, 5, 2), (10, 135, 4),
                                                  (12, 20, 4), (9, 16, 0), (5, 8, 5), (13, 85, 4)])
def test_merkle_cols_matches_oracle(prover, n_log, leaf_len, cap_h):
    n = 1 << n_log
    cols = felts(n * leaf_len, n_log * 100 + leaf_len).reshape(leaf_len, n)
    cap, dig = prover.merkle_cols(cols, cap_h, want_digests=True)
    ocap, odig = O.merkle_tree_cols(cols, cap_h, want_digests=True)
    assert (cap == ocap).all()
    if odig.shape[0]:
        assert (dig == odig).all()


def test_merkle_cap_rows_matches_oracle(prover):
    rows = felts(1024 * 135, 77).reshape(1024, 135)
    assert (prover.merkle_cap(rows, 4) == O.merkle_tree(rows, 4)).all()
    rows3 = felts(64 * 3, 78).reshape(64, 3)
    assert (prover.merkle_cap(rows3, 1) == O.merkle_tree(rows3, 1)).all()


def test_reference_proof_merkle_paths_on_gpu(prover, golden_dir):
    """Leaf rows + siblings taken from a REFERENCE proof (example.bin) hash up to the proof's own cap."""
    from proof_format import parse_proof, find_leaf_index, reference_proofs
    pf = parse_proof(reference_proofs(golden_dir)[0][1])
    caps = [pf["wires_cap"], pf["zs_pp_cap"], pf["quotient_cap"]]
    q = pf["queries"][0]
    # oracle finds the index from the wires path; GPU recomputes the leaf digests and the path
    leaf, sib = q["initial"][1]
    idx = find_leaf_index(leaf, sib, caps[0], O)
    assert idx is not None
    for t, cap in ((1, caps[0]), (2, caps[1]), (3, caps[2])):
        leaf, sib = q["initial"][t]
        cur = prover.hash_no_pad(np.array(leaf, np.uint64))
        i = idx
        for s in sib:
            s = np.array(s, np.uint64)
            cur = prover.two_to_one(cur, s) if i & 1 == 0 else prover.two_to_one(s, cur)
            i >>= 1
        assert cur.tolist() == cap[i]


# ---- NTT -----------------------------------------------------------------------------------
@pytest.mark.parametrize("log_n", [0, 1, 2, 3, 5, 8, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21])
def test_ntt_forward_inverse_match_oracle(prover, log_n):
    import cityprover as cp
    n = 1 << log_n
    batch = 3 if log_n <= 16 else 1
    x = felts(n * batch, log_n).reshape(batch, n)
    if n >= 16:
        x[0, :12] = EDGE
    f = prover.ntt(x)
    for b in range(batch):
        assert (f[b] == O.ntt(x[b])).all(), f"forward log_n={log_n}"
    fb = prover.ntt(x, flags=cp.NTT_BITREV_OUT)
    assert (fb[0] == O.bit_reverse(O.ntt(x[0]))).all()
    i = prover.intt(x)
    for b in range(batch):
        assert (i[b] == O.intt(x[b])).all(), f"inverse log_n={log_n}"
    assert (prover.intt(f) == x).all()


def test_ntt_bitrev_in(prover):
    import cityprover as cp
    x = felts(1 << 12, 5)
    got = prover.ntt(O.bit_reverse(x), flags=cp.NTT_BITREV_IN)
    assert (got == O.ntt(x)).all()


def test_coset_ntt_and_lde(prover):
    import cityprover as cp
    c = felts(1 << 12, 9)
    want = O.coset_lde(c, 0, 7)
    assert (prover.ntt(c, flags=cp.NTT_COSET, shift=7) == want).all()
    # inverse coset undoes it
    assert (prover.ntt(want, flags=cp.NTT_COSET | cp.NTT_INVERSE, shift=7) == c).all()
    lde = prover.lde(c, 3)
    assert (lde == O.coset_lde(c, 3, 7)).all()
    assert (prover.lde(c, 3, bitrev=True) == O.bit_reverse(O.coset_lde(c, 3, 7))).all()
    cs = felts(4 << 10, 10).reshape(4, -1)
    got = prover.lde(cs, 2, shift=3)
    for b in range(4):
        assert (got[b] == O.coset_lde(cs[b], 2, 3)).all()


def test_coset_ntt_above_the_exchange_switch(prover):
    """2^17 = a 5-stage column pass + a 12-stage row pass, both with the half-tile exchange (transforms of 2^16 and up,
    ntt16.h): coset forward against the oracle, inverse coset back, and the rate-2 LDE (pre-scale table path)."""
    import cityprover as cp
    c = felts(2 << 17, 91).reshape(2, -1)
    got = prover.ntt(c, flags=cp.NTT_COSET, shift=7)
    for b in range(2):
        assert (got[b] == O.coset_lde(c[b], 0, 7)).all()
    assert (prover.ntt(got, flags=cp.NTT_COSET | cp.NTT_INVERSE, shift=7) == c).all()
    lde = prover.lde(c, 1, bitrev=True)
    for b in range(2):
        assert (lde[b] == O.bit_reverse(O.coset_lde(c[b], 1, 7))).all()


def test_ntt_strided_batch(prover):
    import cityprover as cp
    n, stride, batch = 1 << 10, (1 << 10) + 24, 5
    host = felts(stride * batch, 31)
    buf = prover.to_device(host)
    prover.ntt_dev(buf.ptr, 10, batch, stride)
    out = buf.download()
    buf.free()
    for b in range(batch):
        assert (out[b * stride:b * stride + n] == O.ntt(host[b * stride:b * stride + n])).all()
        assert (out[b * stride + n:(b + 1) * stride] == host[b * stride + n:(b + 1) * stride]).all()


def test_ntt_full_size_properties(prover):
    """BASELINE full size (2^20, batch 16): round trip + linearity + one oracle column."""
    n, batch = 1 << 20, 16
    x = felts(n * batch, 1234).reshape(batch, n)
    f = prover.ntt(x)
    assert (f[3] == O.ntt(x[3])).all()
    assert (prover.intt(f) == x).all()
    s = ((x[0].astype(object) + x[1].astype(object)) % P).astype(np.uint64)
    fs = ((f[0].astype(object) + f[1].astype(object)) % P).astype(np.uint64)
    assert (prover.ntt(s) == fs).all()


def test_lde_full_size_properties(prover):
    """BASELINE full size: rate-8 coset LDE 2^20 -> 2^23 (shift 7), 3 polynomials: one oracle column, linearity, and
    the defining property LDE(c)[8 j] on the sub-coset == coset NTT of c."""
    import cityprover as cp
    n = 1 << 20
    c = felts(3 * n, 77).reshape(3, n)
    c[2] = ((c[0].astype(object) + c[1].astype(object)) % P).astype(np.uint64)
    lde = prover.lde(c, 3)
    assert lde.shape == (3, n << 3)
    assert (lde[1] == O.coset_lde(c[1], 3, 7)).all()
    assert (lde[2] == ((lde[0].astype(object) + lde[1].astype(object)) % P).astype(np.uint64)).all()
    assert (lde[0][::8] == prover.ntt(c[0], flags=cp.NTT_COSET, shift=7)).all()


# ---- commit (PolynomialBatch::from_values) ---------------------------------------------------
@pytest.mark.parametrize("k,log_n,rate,cap_h", [(3, 4, 3, 2), (20, 10, 3, 4), (135, 12, 3, 4), (16, 12, 3, 4),
                                                (2, 13, 1, 0)])
def test_commit_matches_oracle(prover, k, log_n, rate, cap_h):
    vals = felts(k << log_n, k * 31 + log_n).reshape(k, -1)
    got = prover.commit(vals, rate, cap_h, want=("coeffs", "lde", "cap", "digests"))
    O.lib().or_set_threads(8)
    want = O.commit_batch(vals, rate, cap_h, want=("coeffs", "lde", "cap", "digests"))
    O.lib().or_set_threads(1)
    assert (got["coeffs"] == want["coeffs"]).all()
    assert (got["lde"] == want["lde"]).all()
    assert (got["digests"] == want["digests"]).all()
    assert (got["cap"] == want["cap"]).all()


def test_error_paths_do_not_abort(prover):
    import cityprover as cp
    with pytest.raises(cp.CityProverError):
        prover.ntt_dev(0, 10)
    with pytest.raises(cp.CityProverError):
        prover.ntt_dev(1, 40)
    with pytest.raises(cp.CityProverError):
        prover.merkle_cols(np.zeros((3, 12), np.uint64), 2)  # 12 leaves: not a power of two
    with pytest.raises(cp.CityProverError):
        prover.merkle_cols(np.zeros((3, 8), np.uint64), 4)  # cap larger than the tree
    # context still usable afterwards
    assert (prover.ntt(np.arange(8, dtype=np.uint64)) == O.ntt(np.arange(8, dtype=np.uint64))).all()
