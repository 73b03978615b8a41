"""The AVX-512 Poseidon of the oracle (oracle/poseidon_simd.c: eight states per permutation, used ONLY by bench.py's cpu_baseline
legs as "port-simd") against the scalar textbook form that is the tests' checker: permutations, column-major leaf hashes, tree
levels, whole commitments and a whole proof — bit for bit. Skipped on a CPU without AVX-512."""
import ctypes

import numpy as np
import pytest

import oracle_lib as O

L = O.lib()
L.or_simd_available.restype = ctypes.c_int
L.or_simd_poseidon_enabled.restype = ctypes.c_int
pytestmark = pytest.mark.skipif(not L.or_simd_available(), reason="no AVX-512 on this CPU")
P = O.P


@pytest.fixture
def simd():
    L.or_set_simd_poseidon(1)
    assert L.or_simd_poseidon_enabled() == 1
    yield
    L.or_set_simd_poseidon(0)
    assert L.or_simd_poseidon_enabled() == 0


def test_permutation_of_eight_states_equals_the_scalar_one():
    rng = np.random.default_rng(1)
    edge = np.array([0, 1, P - 1, P - 2, 2**32 - 1, 2**32, 2**63, P - 2**32], dtype=np.uint64)
    for trial in range(50):
        st = rng.integers(0, P, (8, 12), dtype=np.uint64)
        if trial < 8:
            st[:, trial % 12] = edge                       # carry / borrow corners in every lane
        if trial == 8:
            st[:] = 0
        if trial == 9:
            st[:] = P - 1
        want = st.copy()
        L.or_set_simd_poseidon(0)
        L.or_poseidon_permute_many(O.ptr(want), ctypes.c_size_t(8))
        soa = np.ascontiguousarray(st.T)                   # [12][8]
        L.or_poseidon_permute_x8(O.ptr(soa))
        assert (soa.T == want).all()
    # a chain: the output of one permutation is the input of the next, 200 deep
    st = rng.integers(0, P, (8, 12), dtype=np.uint64)
    want, soa = st.copy(), np.ascontiguousarray(st.T)
    for _ in range(200):
        L.or_poseidon_permute_many(O.ptr(want), ctypes.c_size_t(8))
        L.or_poseidon_permute_x8(O.ptr(soa))
    assert (soa.T == want).all() and int(soa.max()) < P


def test_known_answer_of_the_reference_through_the_simd_path(simd):
    # Z[1] = two_to_one(0, 0) of city_crypto/src/hash/cached_zero_hashes.rs:19-26, here as a Merkle tree of 16 zero digests over
    # two levels: every node of level 1 is Z[1], every node of level 2 two_to_one(Z[1], Z[1]) = Z[2]
    z1 = (4330397376401421145, 14124799381142128323, 8742572140681234676, 14345658006221440202)
    st = np.zeros((16, 12), dtype=np.uint64)
    L.or_poseidon_permute_many(O.ptr(st), ctypes.c_size_t(16))   # SIMD path: 2 groups of 8
    assert all(tuple(int(x) for x in row[:4]) == z1 for row in st)


@pytest.mark.parametrize("n_leaves,leaf_len,cap_height", [(64, 5, 0), (256, 135, 4), (8, 9, 3), (1024, 20, 4), (24 * 8, 16, 3), (4, 12, 1)])
def test_merkle_tree_over_columns(simd, n_leaves, leaf_len, cap_height):
    if n_leaves & (n_leaves - 1):
        n_leaves = 128
    cols = O.splitmix64_felts(n_leaves + leaf_len, n_leaves * leaf_len).reshape(leaf_len, n_leaves)
    L.or_set_simd_poseidon(0)
    want = O.merkle_tree_cols(cols, cap_height)
    L.or_set_simd_poseidon(1)
    got = O.merkle_tree_cols(cols, cap_height)
    for a, b in zip(want, got):
        assert (np.asarray(a) == np.asarray(b)).all()


def test_a_whole_commitment_and_a_whole_proof(simd):
    vals = O.splitmix64_felts(7, 20 * 256).reshape(20, 256)
    L.or_set_simd_poseidon(0)
    a = O.Batch(vals, 3, 4)
    L.or_set_simd_poseidon(1)
    b = O.Batch(vals, 3, 4)
    assert (a.cap() == b.cap()).all() and (a.lde() == b.lde()).all()
    a.close()
    b.close()
    import synth_gates as SG
    c = SG.build_gate_set(SG.CITY_COMMON, db=6, seed=17, arity_bits=(2,))
    digest = [6, 6, 6, 6]
    L.or_set_simd_poseidon(0)
    p0, _ = O.prove_full(c["shape"], c["gates"], digest, c["public_inputs"], c["cs_values"], c["wires"])
    L.or_set_simd_poseidon(1)
    p1, _ = O.prove_full(c["shape"], c["gates"], digest, c["public_inputs"], c["cs_values"], c["wires"])
    assert p0 == p1 and len(p0) > 1000
