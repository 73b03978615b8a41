"""ctypes loader for the CPU oracle (oracle/libcityoracle.so) — test infrastructure only."""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
P = 0xFFFFFFFF00000001

_u64p = ctypes.POINTER(ctypes.c_uint64)
_lib = None


def build():
    subprocess.run(["make", "-s", "-C", ORACLE_DIR], check=True)


def lib():
    global _lib
    if _lib is not None:
        return _lib
    so = os.environ.get("CITYORACLE_SO")      # e.g. a sanitizer build (make -C oracle sanitize)
    if not so:
        so = os.path.join(ORACLE_DIR, "libcityoracle.so")
        srcs = [os.path.join(ORACLE_DIR, f) for f in os.listdir(ORACLE_DIR) if f.endswith((".c", ".h"))]
        if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
            build()
    L = ctypes.CDLL(so)
    u64, sz, i = ctypes.c_uint64, ctypes.c_size_t, ctypes.c_int
    sigs = {
        "or_gl_add": (u64, [u64, u64]), "or_gl_sub": (u64, [u64, u64]),
        "or_gl_mul": (u64, [u64, u64]), "or_gl_mul_slow": (u64, [u64, u64]),
        "or_gl_inv": (u64, [u64]), "or_gl_pow": (u64, [u64, u64]),
        "or_gl_root_of_unity": (u64, [i]),
        "or_poseidon_round_constants": (None, [_u64p]),
        "or_poseidon_mds": (None, [_u64p, _u64p]),
        "or_poseidon_permute": (None, [_u64p]),
        "or_poseidon_permute_many": (None, [_u64p, sz]),
        "or_hash_no_pad": (None, [_u64p, sz, _u64p]),
        "or_hash_or_noop": (None, [_u64p, sz, _u64p]),
        "or_two_to_one": (None, [_u64p, _u64p, _u64p]),
        "or_merkle_tree": (None, [_u64p, sz, sz, i, _u64p, _u64p]),
        "or_merkle_tree_cols": (None, [_u64p, sz, sz, sz, i, _u64p, _u64p]),
        "or_merkle_verify": (i, [_u64p, sz, sz, _u64p, sz, _u64p, i]),
        "or_ntt": (None, [_u64p, i]), "or_intt": (None, [_u64p, i]),
        "or_coset_lde": (None, [_u64p, i, i, u64, _u64p]),
        "or_bit_reverse": (None, [_u64p, i]),
        "or_dft_naive": (None, [_u64p, _u64p, i]),
        "or_commit_batch": (None, [_u64p, sz, i, i, i, _u64p, _u64p, _u64p, _u64p]),
        "or_set_threads": (None, [i]), "or_get_threads": (i, []),
        "or_fri_compute_evaluation": (None, [u64, sz, i, _u64p, _u64p, _u64p]),
        "or_fri_query_point": (u64, [sz, i]),
        "or_free": (None, [ctypes.c_void_p]),
    }
    for name, (res, args) in sigs.items():
        f = getattr(L, name)
        f.restype, f.argtypes = res, args
    _lib = L
    return L


def ptr(a):
    if a is None:
        return None
    assert a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_u64p)


def arr(x):
    return np.ascontiguousarray(np.array(x, dtype=np.uint64))


# ---- thin numpy helpers -----------------------------------------------------
def permute(state):
    s = arr(state).copy()
    lib().or_poseidon_permute(ptr(s))
    return s


def permute_many(states):
    s = arr(states).copy()
    lib().or_poseidon_permute_many(ptr(s), s.size // 12)
    return s


def hash_no_pad(x):
    x = arr(x)
    o = np.zeros(4, np.uint64)
    lib().or_hash_no_pad(ptr(x), x.size, ptr(o))
    return o


def hash_or_noop(x):
    x = arr(x)
    o = np.zeros(4, np.uint64)
    lib().or_hash_or_noop(ptr(x), x.size, ptr(o))
    return o


def two_to_one(l, r):
    l, r = arr(l), arr(r)
    o = np.zeros(4, np.uint64)
    lib().or_two_to_one(ptr(l), ptr(r), ptr(o))
    return o


def merkle_tree(leaves, cap_height, want_digests=False):
    leaves = arr(leaves)
    n, k = leaves.shape
    cap = np.zeros((1 << cap_height, 4), np.uint64)
    dig = None
    if want_digests:
        tot = 0
        m = n
        while m > (1 << cap_height):
            tot += m
            m //= 2
        dig = np.zeros((tot, 4), np.uint64)
    lib().or_merkle_tree(ptr(leaves), n, k, cap_height, ptr(dig), ptr(cap))
    return (cap, dig) if want_digests else cap


def merkle_tree_cols(cols, cap_height, want_digests=False):
    cols = arr(cols)
    k, n = cols.shape
    cap = np.zeros((1 << cap_height, 4), np.uint64)
    dig = None
    if want_digests:
        tot, m = 0, n
        while m > (1 << cap_height):
            tot += m
            m //= 2
        dig = np.zeros((tot, 4), np.uint64)
    lib().or_merkle_tree_cols(ptr(cols), n, k, n, cap_height, ptr(dig), ptr(cap))
    return (cap, dig) if want_digests else cap


def merkle_verify(leaf, index, siblings, cap, cap_height):
    leaf, siblings, cap = arr(leaf), arr(siblings), arr(cap)
    return bool(lib().or_merkle_verify(ptr(leaf), leaf.size, index, ptr(siblings),
                                       siblings.size // 4, ptr(cap), cap_height))


def ntt(a):
    a = arr(a).copy()
    lib().or_ntt(ptr(a), int(a.size).bit_length() - 1)
    return a


def intt(a):
    a = arr(a).copy()
    lib().or_intt(ptr(a), int(a.size).bit_length() - 1)
    return a


def coset_lde(coeffs, rate_bits, shift=7):
    c = arr(coeffs)
    out = np.zeros(c.size << rate_bits, np.uint64)
    lib().or_coset_lde(ptr(c), int(c.size).bit_length() - 1, rate_bits, shift, ptr(out))
    return out


def bit_reverse(a):
    a = arr(a).copy()
    lib().or_bit_reverse(ptr(a), int(a.size).bit_length() - 1)
    return a


def commit_batch(values, rate_bits, cap_height, want=("coeffs", "lde", "cap")):
    v = arr(values)
    k, n = v.shape
    log_n = int(n).bit_length() - 1
    N = n << rate_bits
    coeffs = np.zeros((k, n), np.uint64) if "coeffs" in want else None
    lde = np.zeros((k, N), np.uint64) if "lde" in want else None
    cap = np.zeros((1 << cap_height, 4), np.uint64)
    dig = None
    if "digests" in want:
        tot, m = 0, N
        while m > (1 << cap_height):
            tot += m
            m //= 2
        dig = np.zeros((tot, 4), np.uint64)
    lib().or_commit_batch(ptr(v), k, log_n, rate_bits, cap_height, ptr(coeffs), ptr(lde), ptr(dig),
                          ptr(cap))
    return {"coeffs": coeffs, "lde": lde, "cap": cap, "digests": dig}


def splitmix64_felts(seed, n):
    """Deterministic canonical field elements: splitmix64(seed + i) mod p."""
    i = np.arange(n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = (np.uint64(seed) + i) * np.uint64(1)  # copy
        z = z + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z % np.uint64(P)


# ---- transcript / FRI / proof tail -----------------------------------------------------------
class Shape(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int) for n in (
        "degree_bits", "num_constants", "num_routed_wires", "num_wires", "num_challenges",
        "num_partial_products", "quotient_degree_factor", "rate_bits", "cap_height", "pow_bits",
        "num_query_rounds", "n_arity")] + [("arity_bits", ctypes.c_int * 8), ("zero_knowledge", ctypes.c_int)]


class TailDebug(ctypes.Structure):
    _fields_ = [("betas", ctypes.c_uint64 * 8), ("gammas", ctypes.c_uint64 * 8), ("alphas", ctypes.c_uint64 * 8),
                ("zeta", ctypes.c_uint64 * 2), ("fri_betas", (ctypes.c_uint64 * 2) * 8),
                ("pow_response", ctypes.c_uint64), ("query_indices", ctypes.c_uint64 * 64)]


class Challenger(ctypes.Structure):
    _fields_ = [("state", ctypes.c_uint64 * 12), ("inb", ctypes.c_uint64 * 8), ("n_in", ctypes.c_int),
                ("out", ctypes.c_uint64 * 8), ("n_out", ctypes.c_int)]


def standard_shape(degree_bits=12, num_wires=135, num_routed=80, num_constants=5, num_challenges=2,
                   num_partial_products=9, quotient_degree_factor=8, rate_bits=3, cap_height=4, pow_bits=16,
                   num_query_rounds=28, arity_bits=(4, 4)):
    """standard_recursion_config shape as measured from the reference proofs."""
    s = Shape(degree_bits, num_constants, num_routed, num_wires, num_challenges, num_partial_products,
              quotient_degree_factor, rate_bits, cap_height, pow_bits, num_query_rounds, len(arity_bits))
    for i, a in enumerate(arity_bits):
        s.arity_bits[i] = a
    return s


def prove_tail(shape, circuit_digest, public_inputs, cs_values, wires_values, zs_pp_values, quotient_coeffs,
               pow_override=None):
    L = lib()
    L.or_prove_tail.restype = ctypes.c_int
    out = ctypes.POINTER(ctypes.c_uint8)()
    ln = ctypes.c_size_t()
    dbg = TailDebug()
    cd, pi = arr(circuit_digest), arr(public_inputs)
    a, b, c, d = arr(cs_values), arr(wires_values), arr(zs_pp_values), arr(quotient_coeffs)
    rc = L.or_prove_tail(ctypes.byref(shape), ptr(cd), ptr(pi), ctypes.c_size_t(pi.size), ptr(a), ptr(b), ptr(c),
                         ptr(d), ctypes.c_int(0 if pow_override is None else 1),
                         ctypes.c_uint64(pow_override or 0), ctypes.byref(out), ctypes.byref(ln),
                         ctypes.byref(dbg))
    assert rc == 0, rc
    data = ctypes.string_at(out, ln.value)
    L.or_free(out)
    return data, dbg


def verify_tail(shape, circuit_digest, cs_cap, proof_bytes):
    L = lib()
    L.or_verify_tail.restype = ctypes.c_int
    dbg = TailDebug()
    cd, cap = arr(circuit_digest), arr(cs_cap)
    buf = (ctypes.c_uint8 * len(proof_bytes)).from_buffer_copy(proof_bytes)
    rc = L.or_verify_tail(ctypes.byref(shape), ptr(cd), ptr(cap), buf, ctypes.c_size_t(len(proof_bytes)),
                          ctypes.byref(dbg))
    return rc, dbg


def zs_partial_products(shape, wires_values, sigma_values, k_is, betas, gammas):
    L = lib()
    w, s, k, b, g = arr(wires_values), arr(sigma_values), arr(k_is), arr(betas), arr(gammas)
    n = 1 << shape.degree_bits
    out = np.zeros((shape.num_challenges * (1 + shape.num_partial_products), n), np.uint64)
    L.or_zs_partial_products(ctypes.byref(shape), ptr(w), ptr(s), ptr(k), ptr(b), ptr(g), ptr(out))
    return out


# ---- gates / quotient / full prover ---------------------------------------------------------------
GATE_NOOP, GATE_CONSTANT, GATE_PUBLIC_INPUT, GATE_ARITHMETIC = 0, 1, 2, 3


class Gate(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int) for n in ("type", "selector_index", "group_start", "group_end", "param", "param2", "param3")]


class Gates(ctypes.Structure):
    _fields_ = [("n_gates", ctypes.c_int), ("gates", Gate * 32), ("num_selectors", ctypes.c_int),
                ("k_is", ctypes.c_uint64 * 256)]


def make_gates(gate_list, num_selectors, k_is):
    """gate_list: [(type, selector_index, group_start, group_end, param)] in gate-index order."""
    g = Gates()
    g.n_gates = len(gate_list)
    for i, t in enumerate(gate_list):
        g.gates[i] = Gate(*(tuple(t) + (0,) * (7 - len(t))))
    g.num_selectors = num_selectors
    for i, k in enumerate(k_is):
        g.k_is[i] = k
    return g


def prove_full(shape, gates, circuit_digest, public_inputs, cs_values, wires_values, pow_override=None):
    L = lib()
    L.or_prove_full.restype = ctypes.c_int
    out = ctypes.POINTER(ctypes.c_uint8)()
    ln = ctypes.c_size_t()
    dbg = TailDebug()
    cd, pi, a, b = arr(circuit_digest), arr(public_inputs), arr(cs_values), arr(wires_values)
    rc = L.or_prove_full(ctypes.byref(shape), ctypes.byref(gates), ptr(cd), ptr(pi), ctypes.c_size_t(pi.size), ptr(a),
                         ptr(b), ctypes.c_int(0 if pow_override is None else 1), ctypes.c_uint64(pow_override or 0),
                         ctypes.byref(out), ctypes.byref(ln), ctypes.byref(dbg))
    assert rc == 0, rc
    data = ctypes.string_at(out, ln.value)
    L.or_free(out)
    return data, dbg


SALT_SIZE = 4


def prove_full_zk(shape, gates, circuit_digest, public_inputs, cs_values, wires_values, salts, pow_override=None):
    """Zero-knowledge circuits (shape.zero_knowledge = 1): salts = [3][SALT_SIZE][N] uint64, leaf order."""
    L = lib()
    L.or_prove_full_zk.restype = ctypes.c_int
    out = ctypes.POINTER(ctypes.c_uint8)()
    ln = ctypes.c_size_t()
    dbg = TailDebug()
    cd, pi, a, b, sl = arr(circuit_digest), arr(public_inputs), arr(cs_values), arr(wires_values), arr(salts)
    assert sl.size == 3 * SALT_SIZE << (shape.degree_bits + shape.rate_bits)
    rc = L.or_prove_full_zk(ctypes.byref(shape), ctypes.byref(gates), ptr(cd), ptr(pi), ctypes.c_size_t(pi.size), ptr(a),
                            ptr(b), ptr(sl), ctypes.c_int(0 if pow_override is None else 1),
                            ctypes.c_uint64(pow_override or 0), ctypes.byref(out), ctypes.byref(ln), ctypes.byref(dbg))
    assert rc == 0, rc
    data = ctypes.string_at(out, ln.value)
    L.or_free(out)
    return data, dbg


def verify_full(shape, gates, circuit_digest, cs_cap, proof_bytes):
    """FRI / transcript / Merkle checks (verify_tail) + the vanishing identity at zeta from the openings."""
    from proof_format import parse_proof
    rc, dbg = verify_tail(shape, circuit_digest, cs_cap, proof_bytes)
    if rc != 0:
        return rc
    p = parse_proof(proof_bytes)
    o = p["openings"]
    flat = lambda k: arr([c for e in o[k] for c in e])
    pi_hash = hash_no_pad(arr(p["public_inputs"]))
    nc = shape.num_challenges
    L = lib()
    L.or_check_vanishing.restype = ctypes.c_int
    rc = L.or_check_vanishing(ctypes.byref(shape), ctypes.byref(gates), ptr(pi_hash), ptr(arr(list(dbg.zeta))),
                              ptr(flat("constants")), ptr(flat("plonk_sigmas")), ptr(flat("wires")),
                              ptr(flat("plonk_zs")), ptr(flat("plonk_zs_next")), ptr(flat("partial_products")),
                              ptr(flat("quotient_polys")), ptr(arr(list(dbg.betas)[:nc])),
                              ptr(arr(list(dbg.gammas)[:nc])), ptr(arr(list(dbg.alphas)[:nc])))
    return 0 if rc == 0 else -1000 + rc
GATE_POSEIDON = 4
GATE_COMPARISON, GATE_U32_ARITHMETIC, GATE_U32_RANGE_CHECK = 5, 6, 7
GATE_U32_ADD_MANY, GATE_U32_SUBTRACTION, GATE_U32_INTERLEAVE, GATE_UNINTERLEAVE_TO_U32, GATE_UNINTERLEAVE_TO_B32 = 8, 9, 10, 11, 12
(GATE_ARITHMETIC_EXT, GATE_MUL_EXT, GATE_BASE_SUM, GATE_RANDOM_ACCESS, GATE_REDUCING, GATE_REDUCING_EXT, GATE_POSEIDON_MDS,
 GATE_COSET_INTERPOLATION) = 13, 14, 15, 16, 17, 18, 19, 20
GATE_EXPONENTIATION = 21


# ---- BLS12-381 G1 (oracle/bls12_381.c) ---------------------------------------------------------------------
def _limbs(v, n):
    return np.array([(int(v) >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(n)], dtype=np.uint64)


def _int(limbs):
    return sum(int(x) << (64 * i) for i, x in enumerate(limbs))


def bls_constants():
    p, r, g = np.zeros(6, np.uint64), np.zeros(4, np.uint64), np.zeros(12, np.uint64)
    lib().or_bls_constants(ptr(p), ptr(r), ptr(g))
    return _int(p), _int(r), (_int(g[:6]), _int(g[6:]))


def bls_point(pt):
    """pt: None (infinity) or (x, y) ints -> (xy limbs, inf flag)"""
    if pt is None:
        return np.zeros(12, np.uint64), 1
    return np.concatenate([_limbs(pt[0], 6), _limbs(pt[1], 6)]), 0


def _bls_out(xy, inf):
    return None if inf.value else (_int(xy[:6]), _int(xy[6:]))


def bls_g1_mul(pt, k):
    xy, inf = bls_point(pt)
    out, oi = np.zeros(12, np.uint64), ctypes.c_int()
    kk = _limbs(k, 4)
    lib().or_bls_g1_mul(ptr(xy), ctypes.c_int(inf), ptr(kk), ptr(out), ctypes.byref(oi))
    return _bls_out(out, oi)


def bls_g1_add(a, b):
    axy, ai = bls_point(a)
    bxy, bi = bls_point(b)
    out, oi = np.zeros(12, np.uint64), ctypes.c_int()
    lib().or_bls_g1_add(ptr(axy), ctypes.c_int(ai), ptr(bxy), ctypes.c_int(bi), ptr(out), ctypes.byref(oi))
    return _bls_out(out, oi)


def bls_g1_on_curve(pt):
    xy, _ = bls_point(pt)
    lib().or_bls_g1_on_curve.restype = ctypes.c_int
    return bool(lib().or_bls_g1_on_curve(ptr(xy)))


def bls_g1_msm(scalars, points_xy, points_inf=None):
    """scalars: (n, 4) uint64; points_xy: (n, 12) uint64 affine canonical; points_inf: optional (n,) uint8"""
    s, pxy = arr(scalars), arr(points_xy)
    n = s.shape[0]
    out, oi = np.zeros(12, np.uint64), ctypes.c_int()
    pi = None if points_inf is None else np.ascontiguousarray(points_inf, dtype=np.uint8)
    lib().or_bls_g1_msm(ptr(s), ptr(pxy), None if pi is None else pi.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)),
                        ctypes.c_size_t(n), ptr(out), ctypes.byref(oi))
    return _bls_out(out, oi)


# ---- G2: points are ((x0, x1), (y0, y1)) with coordinates c0 + c1*u ----
def bls_g2_generator():
    g = np.zeros(24, np.uint64)
    lib().or_bls_g2_generator(ptr(g))
    return ((_int(g[0:6]), _int(g[6:12])), (_int(g[12:18]), _int(g[18:24])))


def bls_point2(pt):
    if pt is None:
        return np.zeros(24, np.uint64), 1
    (x0, x1), (y0, y1) = pt
    return np.concatenate([_limbs(x0, 6), _limbs(x1, 6), _limbs(y0, 6), _limbs(y1, 6)]), 0


def _bls_out2(xy, inf):
    return None if inf.value else ((_int(xy[0:6]), _int(xy[6:12])), (_int(xy[12:18]), _int(xy[18:24])))


def bls_g2_mul(pt, k):
    xy, inf = bls_point2(pt)
    out, oi = np.zeros(24, np.uint64), ctypes.c_int()
    kk = _limbs(k, 4)
    lib().or_bls_g2_mul(ptr(xy), ctypes.c_int(inf), ptr(kk), ptr(out), ctypes.byref(oi))
    return _bls_out2(out, oi)


def bls_g2_add(a, b):
    axy, ai = bls_point2(a)
    bxy, bi = bls_point2(b)
    out, oi = np.zeros(24, np.uint64), ctypes.c_int()
    lib().or_bls_g2_add(ptr(axy), ctypes.c_int(ai), ptr(bxy), ctypes.c_int(bi), ptr(out), ctypes.byref(oi))
    return _bls_out2(out, oi)


def bls_g2_on_curve(pt):
    xy, _ = bls_point2(pt)
    lib().or_bls_g2_on_curve.restype = ctypes.c_int
    return bool(lib().or_bls_g2_on_curve(ptr(xy)))


def bls_g2_msm(scalars, points_xy, points_inf=None):
    s, pxy = arr(scalars), arr(points_xy)
    n = s.shape[0]
    out, oi = np.zeros(24, np.uint64), ctypes.c_int()
    pi = None if points_inf is None else np.ascontiguousarray(points_inf, dtype=np.uint8)
    lib().or_bls_g2_msm(ptr(s), ptr(pxy), None if pi is None else pi.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)),
                        ctypes.c_size_t(n), ptr(out), ctypes.byref(oi))
    return _bls_out2(out, oi)


# ---- F_r and its NTT ----
def fr_ntt(values, inverse=False, shift=None):
    """values: (n, 4) uint64 canonical F_r elements; returns the transformed copy"""
    a = arr(values).copy()
    log_n = int(a.shape[0]).bit_length() - 1
    sh = None if shift is None else _limbs(shift, 4)
    lib().or_fr_ntt(ptr(a), ctypes.c_int(log_n), ctypes.c_int(1 if inverse else 0), None if sh is None else ptr(sh))
    return a


def groth16_quotient(a, b, c):
    """h = (a b - c) / (x^n - 1) from (n, 4) evaluation arrays; returns h's coefficients"""
    a, b, c = arr(a).copy(), arr(b).copy(), arr(c).copy()
    lib().or_groth16_quotient(ptr(a), ptr(b), ptr(c), ctypes.c_int(int(a.shape[0]).bit_length() - 1))
    return a


def fr_dft_naive(values):
    a = arr(values)
    out = np.zeros_like(a)
    lib().or_fr_dft_naive(ptr(a), ptr(out), ctypes.c_int(int(a.shape[0]).bit_length() - 1))
    return out


def fr_root_of_unity(log_n):
    out = np.zeros(4, np.uint64)
    lib().or_fr_root_of_unity(ctypes.c_int(log_n), ptr(out))
    return _int(out)


# ---- the generic seams: PolynomialBatch handles, prove_openings / verify_fri_proof over any instance -------------------
class FriParams(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int) for n in ("degree_bits", "rate_bits", "cap_height", "pow_bits", "num_query_rounds", "n_arity")] + [
        ("arity_bits", ctypes.c_int * 8)]


class FriRange(ctypes.Structure):
    _fields_ = [("oracle", ctypes.c_uint32), ("first", ctypes.c_uint32), ("count", ctypes.c_uint32)]


class FriBatch(ctypes.Structure):
    _fields_ = [("point", ctypes.c_uint64 * 2), ("ranges", ctypes.POINTER(FriRange)), ("n_ranges", ctypes.c_size_t)]


def fri_params(degree_bits, rate_bits, cap_height, pow_bits, num_query_rounds, arity_bits):
    p = FriParams(degree_bits, rate_bits, cap_height, pow_bits, num_query_rounds, len(arity_bits))
    for i, a in enumerate(arity_bits):
        p.arity_bits[i] = a
    return p


def _fri_batches(batches):
    a = (FriBatch * len(batches))()
    keep = []
    for i, (pt, ranges) in enumerate(batches):
        rr = (FriRange * max(1, len(ranges)))()
        for j, (o, f, c) in enumerate(ranges):
            rr[j] = FriRange(o, f, c)
        keep.append(rr)
        a[i].point[0], a[i].point[1] = int(pt[0]), int(pt[1])
        a[i].ranges = ctypes.cast(rr, ctypes.POINTER(FriRange))
        a[i].n_ranges = len(ranges)
    return a, keep


class Batch:
    """or_batch: PolynomialBatch on the CPU (coefficients, bit-reversed LDE, Merkle tree)."""

    def __init__(self, polys, rate_bits, cap_height, from_coeffs=False, salts=None):
        L = lib()
        L.or_batch_commit.restype = ctypes.c_void_p
        v = arr(polys)
        self.k, n = v.shape
        self.log_n, self.rate_bits, self.cap_height = int(n).bit_length() - 1, rate_bits, cap_height
        s = None if salts is None else arr(salts)
        self.blinding = salts is not None
        self.h = ctypes.c_void_p(L.or_batch_commit(ptr(v), ctypes.c_size_t(self.k), self.log_n, rate_bits, cap_height,
                                                   1 if from_coeffs else 0, ptr(s)))

    def _view(self, fn, count):
        L = lib()
        f = getattr(L, fn)
        f.restype = _u64p
        f.argtypes = [ctypes.c_void_p]
        return np.ctypeslib.as_array(f(self.h), shape=(count,)).copy()

    def cap(self):
        return self._view("or_batch_cap", 4 << self.cap_height).reshape(-1, 4)

    def coeffs(self):
        return self._view("or_batch_coeffs", self.k << self.log_n).reshape(self.k, -1)

    def lde(self):
        return self._view("or_batch_lde", (self.k + (4 if self.blinding else 0)) << (self.log_n + self.rate_bits)).reshape(-1, 1 << (self.log_n + self.rate_bits))

    def eval_ext(self, point, first=0, count=None):
        count = self.k - first if count is None else count
        out = np.zeros((count, 2), np.uint64)
        pt = arr(point)
        L = lib()
        L.or_batch_eval_ext.restype = None
        L.or_batch_eval_ext(self.h, ctypes.c_size_t(first), ctypes.c_size_t(count), ptr(pt), ptr(out))
        return out

    def lde_rows(self, first_index, count, step=1):
        out = np.zeros((count, self.k), np.uint64)
        L = lib()
        L.or_batch_lde_rows.restype = None
        L.or_batch_lde_rows(self.h, ctypes.c_size_t(first_index), ctypes.c_size_t(count), ctypes.c_size_t(step), ptr(out))
        return out

    def close(self):
        if self.h:
            L = lib()
            L.or_batch_free.restype = None
            L.or_batch_free(self.h)
            self.h = None


def challenger_new():
    c = Challenger()
    lib().or_ch_init(ctypes.byref(c))
    return c


def challenger_observe(c, elems):
    e = arr(elems).ravel()
    L = lib()
    L.or_ch_observe.restype = None
    L.or_ch_observe(ctypes.byref(c), ptr(e), ctypes.c_size_t(e.size))


def challenger_challenges(c, count):
    L = lib()
    L.or_ch_challenge.restype = ctypes.c_uint64
    return np.array([L.or_ch_challenge(ctypes.byref(c)) for _ in range(count)], dtype=np.uint64)


def challenger_tuple(c):
    return (tuple(c.state), tuple(c.inb[:c.n_in]), tuple(c.out[:c.n_out]))


def fri_prove(oracles, batches, params, challenger, pow_override=None):
    """or_fri_prove; advances `challenger` (oracle_lib.Challenger). Returns (FriProof bytes, TailDebug)."""
    L = lib()
    L.or_fri_prove.restype = ctypes.c_int
    hs = (ctypes.c_void_p * len(oracles))(*[o.h for o in oracles])
    a, keep = _fri_batches(batches)
    out = ctypes.POINTER(ctypes.c_uint8)()
    ln = ctypes.c_size_t()
    dbg = TailDebug()
    rc = L.or_fri_prove(hs, ctypes.c_size_t(len(oracles)), a, ctypes.c_size_t(len(batches)), ctypes.byref(params),
                        ctypes.byref(challenger), ctypes.c_int(0 if pow_override is None else 1),
                        ctypes.c_uint64(pow_override or 0), ctypes.byref(out), ctypes.byref(ln), ctypes.byref(dbg))
    assert rc == 0, rc
    data = ctypes.string_at(out, ln.value)
    L.or_free(out)
    return data, dbg


def fri_verify(params, oracle_infos, caps, batches, opened_values, challenger, proof_bytes):
    """or_fri_verify -> (rc, TailDebug); rc == 0: accepted and `challenger` advanced."""
    L = lib()
    L.or_fri_verify.restype = ctypes.c_int
    nps = (ctypes.c_uint32 * len(oracle_infos))(*[int(k) for k, _ in oracle_infos])
    bl = (ctypes.c_uint32 * len(oracle_infos))(*[int(bool(b)) for _, b in oracle_infos])
    cap_arrs = [arr(c) for c in caps]
    capp = (_u64p * len(cap_arrs))(*[ptr(c) for c in cap_arrs])
    a, keep = _fri_batches(batches)
    ov = [arr(o) for o in opened_values]
    ovp = (_u64p * len(ov))(*[ptr(o) for o in ov])
    buf = (ctypes.c_uint8 * len(proof_bytes)).from_buffer_copy(proof_bytes)
    dbg = TailDebug()
    rc = L.or_fri_verify(ctypes.byref(params), nps, bl, ctypes.c_size_t(len(oracle_infos)), capp, a, ctypes.c_size_t(len(batches)), ovp,
                         ctypes.byref(challenger), buf, ctypes.c_size_t(len(proof_bytes)), ctypes.byref(dbg))
    return rc, dbg


# ---- generic AIR machinery (oracle/stark_air.c): checker of cp_air_* / cp_stark_* ----------------------------------------
class AirProgramC(ctypes.Structure):
    _fields_ = [("map", ctypes.c_int), ("ops", ctypes.c_void_p), ("n_ops", ctypes.c_size_t), ("consts", _u64p), ("n_consts", ctypes.c_size_t),
                ("n_columns", ctypes.c_uint32), ("n_public", ctypes.c_uint32), ("n_global", ctypes.c_uint32), ("n_challenge", ctypes.c_uint32),
                ("n_out_columns", ctypes.c_uint32)]


class StarkStepC(ctypes.Structure):
    _fields_ = [("kind", ctypes.c_int), ("program", ctypes.POINTER(AirProgramC)), ("first", ctypes.c_uint32), ("count", ctypes.c_uint32),
                ("flags", ctypes.c_uint32), ("modulus", ctypes.c_uint64 * 2)]


class StarkDescC(ctypes.Structure):
    _fields_ = [("degree_bits", ctypes.c_int), ("quotient_degree_bits", ctypes.c_int), ("num_challenges", ctypes.c_uint32), ("fri", FriParams),
                ("n_trace_columns", ctypes.c_uint32), ("n_extended_columns", ctypes.c_uint32), ("n_round_challenges", ctypes.c_uint32),
                ("n_public", ctypes.c_uint32), ("n_global", ctypes.c_uint32), ("steps", ctypes.POINTER(StarkStepC)), ("n_steps", ctypes.c_size_t),
                ("constraints", ctypes.POINTER(AirProgramC))]


class AirProgram:
    """or_air_program over numpy arrays it keeps alive. ops: (n, 4) uint32 rows (op, a, b, 0)."""

    def __init__(self, kind, ops, consts=(), n_columns=0, n_public=0, n_global=0, n_challenge=0, n_out_columns=0):
        self.ops = np.ascontiguousarray(np.asarray(ops, dtype=np.uint32).reshape(-1, 4))
        self.consts = arr(list(consts) if not isinstance(consts, np.ndarray) else consts)
        self.c = AirProgramC(kind, self.ops.ctypes.data if self.ops.size else None, len(self.ops), ptr(self.consts) if self.consts.size else None,
                             self.consts.size, n_columns, n_public, n_global, n_challenge, n_out_columns)
        self.kind, self.n_columns, self.n_out_columns = kind, n_columns, n_out_columns

    def check(self):
        L = lib()
        L.or_air_check.restype = ctypes.c_size_t
        return L.or_air_check(ctypes.byref(self.c))

    def num_constraints(self):
        L = lib()
        L.or_air_num_constraints.restype = ctypes.c_size_t
        return L.or_air_num_constraints(ctypes.byref(self.c))

    def eval_row(self, local, nxt, publics=(), globals_=(), challenges=()):
        vals = np.zeros(max(len(self.ops), 1), np.uint64)
        a = [arr(x) for x in (local, nxt, publics, globals_, challenges)]
        L = lib()
        L.or_air_eval_row.restype = None
        L.or_air_eval_row(ctypes.byref(self.c), *[ptr(x) for x in a], ptr(vals))
        return vals[:len(self.ops)]

    def eval_ext(self, local, nxt, publics=(), globals_=(), challenges=()):
        n = self.num_constraints()
        out = np.zeros((max(n, 1), 2), np.uint64)
        kinds = np.zeros(max(n, 1), np.uint32)
        a = [arr(x) for x in (local, nxt, publics, globals_, challenges)]
        L = lib()
        L.or_air_eval_ext.restype = None
        L.or_air_eval_ext(ctypes.byref(self.c), *[ptr(x) for x in a], ptr(out), kinds.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)))
        return out[:n], kinds[:n]

    def map(self, in_cols, publics=(), globals_=(), challenges=()):
        v = arr(in_cols)
        n = v.shape[1]
        out = np.zeros((max(self.n_out_columns, 1), n), np.uint64)
        a = [arr(x) for x in (publics, globals_, challenges)]
        L = lib()
        L.or_air_map.restype = None
        L.or_air_map(ctypes.byref(self.c), ptr(v), ptr(out), ctypes.c_size_t(n), *[ptr(x) for x in a])
        return out[:self.n_out_columns]


def air_quotient(program, oracles, qdb, alphas, publics=(), globals_=(), challenges=()):
    """or_air_quotient -> (len(alphas) * 2^qdb, n) coefficient vectors"""
    L = lib()
    L.or_air_quotient.restype = ctypes.c_int
    hs = (ctypes.c_void_p * len(oracles))(*[o.h for o in oracles])
    n = 1 << oracles[0].log_n
    al = arr(alphas)
    out = np.zeros((al.size << qdb, n), np.uint64)
    a = [arr(x) for x in (publics, globals_, challenges)]
    rc = L.or_air_quotient(ctypes.byref(program.c), hs, ctypes.c_size_t(len(oracles)), ctypes.c_int(qdb), *[ptr(x) for x in a], ptr(al),
                           ctypes.c_size_t(al.size), ptr(out))
    assert rc == 0, rc
    return out


def cubic_mul(m, a, b):
    out = np.zeros(3, np.uint64)
    lib().or_cubic_mul(ptr(arr(m)), ptr(arr(a)), ptr(arr(b)), ptr(out))
    return out


def cubic_batch_inverse(m, cols):
    v = arr(cols).copy()
    L = lib()
    L.or_cubic_batch_inverse.restype = None
    L.or_cubic_batch_inverse(ptr(arr(m)), ptr(v), ctypes.c_size_t(v.shape[0] // 3), ctypes.c_size_t(v.shape[1]))
    return v


def column_prefix_sum(cols, exclusive=False):
    v = arr(cols).copy()
    L = lib()
    L.or_column_prefix_sum.restype = None
    L.or_column_prefix_sum(ptr(v), ctypes.c_size_t(v.shape[0]), ctypes.c_size_t(v.shape[1]), ctypes.c_int(int(exclusive)))
    return v


def stark_desc(degree_bits, quotient_degree_bits, num_challenges, fri, n_trace_columns, constraints, n_extended_columns=0, n_round_challenges=0,
               n_public=0, n_global=0, steps=()):
    """or_stark_desc; steps as cityprover.stark_desc takes them (with oracle_lib.AirProgram objects). Returns (desc, keep-alive)."""
    sa = (StarkStepC * max(1, len(steps)))()
    for i, st in enumerate(steps):
        if st[0] == "map":
            sa[i].kind, sa[i].program = 0, ctypes.pointer(st[1].c)
        elif st[0] == "cubic_inverse":
            sa[i].kind, sa[i].first, sa[i].count = 1, st[1], st[2]
            sa[i].modulus[0], sa[i].modulus[1] = int(st[3][0]), int(st[3][1])
        else:
            sa[i].kind, sa[i].first, sa[i].count, sa[i].flags = 2, st[1], st[2], int(bool(st[3]))
    d = StarkDescC(degree_bits, quotient_degree_bits, num_challenges, fri, n_trace_columns, n_extended_columns, n_round_challenges, n_public, n_global,
                   ctypes.cast(sa, ctypes.POINTER(StarkStepC)), len(steps), ctypes.pointer(constraints.c))
    return d, (sa, steps, constraints)


def stark_prove(desc, trace, challenger, publics=(), globals_=(), pow_override=None):
    L = lib()
    L.or_stark_prove.restype = ctypes.c_int
    t = arr(trace)
    out, ln = ctypes.POINTER(ctypes.c_uint8)(), ctypes.c_size_t(0)
    rc = L.or_stark_prove(ctypes.byref(desc), ptr(t), ptr(arr(publics)), ptr(arr(globals_)), ctypes.byref(challenger),
                          ctypes.c_int(0 if pow_override is None else 1), ctypes.c_uint64(pow_override or 0), ctypes.byref(out), ctypes.byref(ln))
    assert rc == 0, rc
    data = ctypes.string_at(out, ln.value)
    L.or_free(out)
    return data


def stark_verify(desc, challenger, proof, publics=(), globals_=()):
    L = lib()
    L.or_stark_verify.restype = ctypes.c_int
    buf = (ctypes.c_uint8 * len(proof)).from_buffer_copy(proof)
    return L.or_stark_verify(ctypes.byref(desc), ptr(arr(publics)), ptr(arr(globals_)), ctypes.byref(challenger), buf, ctypes.c_size_t(len(proof)))
