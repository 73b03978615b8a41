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
