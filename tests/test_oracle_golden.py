"""Pin the CPU oracle against the reference's own known-answer data.

Nothing here touches the GPU or /root/reference; the fixtures under tests/golden/ were
extracted by tests/golden/make_golden.py."""
import json
import os

import numpy as np
import pytest

import oracle_lib as O

P = O.P


def load(golden_dir, name):
    return json.load(open(os.path.join(golden_dir, name)))


def test_field_fast_reduce_matches_slow():
    rng = np.random.default_rng(1)
    L = O.lib()
    edge = [0, 1, 2, P - 1, P - 2, 0xFFFFFFFF, 0x100000000, 0xFFFFFFFF00000000, 1 << 63]
    vals = edge + [int(x) % P for x in rng.integers(0, 2**64, 2000, dtype=np.uint64)]
    for a in vals[:60]:
        for b in vals[:60]:
            assert L.or_gl_mul(a, b) == (a * b) % P == L.or_gl_mul_slow(a, b)
            assert L.or_gl_add(a, b) == (a + b) % P
            assert L.or_gl_sub(a, b) == (a - b) % P
    for a in vals[1:300]:
        if a:
            assert L.or_gl_mul(a, L.or_gl_inv(a)) == 1


def test_roots_of_unity():
    L = O.lib()
    for k in range(1, 33):
        w = L.or_gl_root_of_unity(k)
        assert pow(w, 1 << k, P) == 1 and pow(w, 1 << (k - 1), P) == P - 1
    # 2^32-th root is 7^((p-1)/2^32); consecutive roots square into each other
    assert L.or_gl_root_of_unity(32) == pow(7, (P - 1) >> 32, P)
    assert pow(L.or_gl_root_of_unity(20), 2, P) == L.or_gl_root_of_unity(19)


def test_poseidon_first_round_constant():
    rc = np.zeros(360, np.uint64)
    O.lib().or_poseidon_round_constants(O.ptr(rc))
    # SURVEY.md finding #5 quotes the first constant of the upstream table
    assert int(rc[0]) == 0xB585F766F2144405
    assert all(int(x) < P for x in rc)
    assert len(set(int(x) for x in rc)) == 360


def test_p1_iterated_zero_hashes(golden_dir):
    z = load(golden_dir, "poseidon_zero_hashes.json")["two_to_one"]
    assert z[0] == [0, 0, 0, 0]
    cur = np.zeros(4, np.uint64)
    for i in range(1, 128):
        cur = O.two_to_one(cur, cur)
        assert cur.tolist() == z[i], f"level {i}"


def test_p2_iterated_marked_leaf_hashes(golden_dir):
    z = load(golden_dir, "poseidon_zero_hashes.json")["marked_leaf"]
    # level 0 -> 1 uses the marked form hash_no_pad(l || r || 1) (9 elements: 2 permutations,
    # overwrite-mode sponge); the levels above are plain two_to_one
    # (city_crypto/src/hash/traits/hasher.rs:82-95, merkle/core.rs marked variants).
    cur = np.zeros(4, np.uint64)
    first = O.hash_no_pad(list(cur) + list(cur) + [1])
    assert first.tolist() == z[1]
    cur = first
    for i in range(2, 128):
        cur = O.two_to_one(cur, cur)
        assert cur.tolist() == z[i], f"level {i}"


def test_p3_circuit_fingerprint_roots(golden_dir):
    fps = load(golden_dir, "circuit_fingerprints.json")
    assert len(fps) >= 6
    for f in fps:
        assert O.two_to_one(f["leaf"], f["aggregator"]).tolist() == f["root"], f["name"]


def _root_from_path(value, index, siblings):
    cur = np.array(value, np.uint64)
    for i, s in enumerate(siblings):
        cur = O.two_to_one(cur, s) if (index >> i) & 1 == 0 else O.two_to_one(s, cur)
    return cur.tolist()


def test_p4_merkle_proofs(golden_dir):
    cases = load(golden_dir, "merkle_proofs.json")
    assert len(cases) == 39
    for c in cases:
        assert _root_from_path(c["value"], c["index"], c["siblings"]) == c["root"]
        # negative: flipping one element of the value must break the root
        bad = list(c["value"])
        bad[0] = (bad[0] + 1) % P
        assert _root_from_path(bad, c["index"], c["siblings"]) != c["root"]


def test_p4_delta_merkle_proofs(golden_dir):
    cases = load(golden_dir, "delta_merkle_proofs.json")
    assert len(cases) == 12
    for c in cases:
        assert _root_from_path(c["old_value"], c["index"], c["siblings"]) == c["old_root"]
        assert _root_from_path(c["new_value"], c["index"], c["siblings"]) == c["new_root"]


def test_p6_example_bin_delta_merkle_witnesses(golden_dir):
    ws = load(golden_dir, "example_delta_merkle.json")
    fps = {tuple(f["root"]) for f in load(golden_dir, "circuit_fingerprints.json")}
    n = 0
    for w in ws:
        assert tuple(w["allowed_circuit_hashes_root"]) in fps
        for p in w["proofs"]:
            assert len(p["siblings"]) == 32
            assert _root_from_path(p["old_value"], p["index"], p["siblings"]) == p["old_root"]
            assert _root_from_path(p["new_value"], p["index"], p["siblings"]) == p["new_root"]
            n += 1
    assert n >= 26


def test_merkle_verify_uses_same_direction_rule():
    rng = np.random.default_rng(5)
    leaves = rng.integers(0, P, (64, 7), dtype=np.uint64)
    cap, dig = O.merkle_tree(leaves, 2, want_digests=True)
    # path for leaf 37: levels 64, 32, 16, 8 -> cap of 4
    idx, sib, off, m = 37, [], 0, 64
    i = idx
    while m > 4:
        sib.append(dig[off + (i ^ 1)])
        off += m
        m //= 2
        i >>= 1
    assert O.merkle_verify(leaves[idx], idx, np.array(sib), cap, 2)
    assert not O.merkle_verify(leaves[idx], idx ^ 1, np.array(sib), cap, 2)


def test_fast_mds_form_is_the_same_permutation(golden_dir):
    """or_set_fast_poseidon(1) — the multiplier-free 32-bit-plane MDS the CPU baseline of bench.py runs on — is the same
    map as the textbook layer: equal on random and edge states, and it reproduces the reference's iterated zero hashes."""
    L = O.lib()
    st = O.splitmix64_felts(77, 12 * 5000).reshape(-1, 12).copy()
    st[0] = 0
    st[1] = O.P - 1
    st[2, ::2] = O.P - 1
    try:
        L.or_set_fast_poseidon(0)
        want = O.permute_many(st.copy())
        L.or_set_fast_poseidon(1)
        assert (O.permute_many(st.copy()) == want).all()
        zh = load(golden_dir, "poseidon_zero_hashes.json")["two_to_one"]
        cur = np.zeros(4, np.uint64)
        for i in range(1, 40):
            cur = O.two_to_one(cur, cur)
            assert [int(x) for x in cur] == zh[i]
    finally:
        L.or_set_fast_poseidon(0)
