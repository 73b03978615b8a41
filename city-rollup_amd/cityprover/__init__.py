"""Host-side binding of libcityprover_hip.so (the C ABI in include/cityprover.h).

This is plumbing only: every computation happens in the HIP library. There is no CPU fallback —
constructing a `Prover` without a GPU raises `CityProverError`.

The method names mirror what the reference reaches through plonky2 (`PolynomialBatch::from_values`,
`MerkleTree::new`, `PoseidonHash::{hash_no_pad,two_to_one}`, `fft`/`ifft`/`coset_fft`), see
SURVEY.md §8(a).
"""
import ctypes
import os

import numpy as np

from . import build as _build

P = 0xFFFFFFFF00000001
NTT_INVERSE, NTT_BITREV_OUT, NTT_COSET, NTT_BITREV_IN = 1, 2, 4, 8

_u64p = ctypes.POINTER(ctypes.c_uint64)
_vp = ctypes.c_void_p

# name -> (restype, argtypes); must list every symbol declared in include/cityprover.h
ABI = {
    "cp_abi_version": (ctypes.c_int, []),
    "cp_device_count": (ctypes.c_int, []),
    "cp_fault_inject": (ctypes.c_int, [ctypes.c_int, ctypes.c_long]),
    "cp_ctx_create": (_vp, [ctypes.c_int]),
    "cp_ctx_destroy": (None, [_vp]),
    "cp_ctx_set_lanes": (ctypes.c_int, [_vp, ctypes.c_int]),
    "cp_ctx_set_device_transcript": (ctypes.c_int, [_vp, ctypes.c_int]),
    "cp_ctx_set_option": (ctypes.c_int, [_vp, ctypes.c_char_p, ctypes.c_long]),
    "cp_last_error": (ctypes.c_char_p, [_vp]),
    "cp_dev_alloc": (ctypes.c_int, [_vp, ctypes.c_size_t, ctypes.POINTER(_vp)]),
    "cp_dev_free": (ctypes.c_int, [_vp, _vp]),
    "cp_host_alloc": (ctypes.c_int, [_vp, ctypes.c_size_t, ctypes.POINTER(_vp)]),
    "cp_host_free": (ctypes.c_int, [_vp, _vp]),
    "cp_h2d": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_size_t]),
    "cp_d2h": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_size_t]),
    "cp_d2d": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_size_t]),
    "cp_sync": (ctypes.c_int, [_vp]),
    "cp_event_create": (ctypes.c_int, [_vp, ctypes.POINTER(_vp)]),
    "cp_event_destroy": (ctypes.c_int, [_vp, _vp]),
    "cp_event_record": (ctypes.c_int, [_vp, _vp]),
    "cp_event_elapsed_ms": (ctypes.c_int, [_vp, _vp, _vp, ctypes.POINTER(ctypes.c_float)]),
    "cp_profile_begin": (ctypes.c_int, [_vp]),
    "cp_profile_end": (ctypes.c_int, [_vp, ctypes.c_char_p, ctypes.c_size_t]),
    "cp_ntt_dev": (ctypes.c_int, [_vp, _vp, ctypes.c_int, ctypes.c_size_t, ctypes.c_size_t,
                                  ctypes.c_uint, ctypes.c_uint64]),
    "cp_ntt": (ctypes.c_int, [_vp, _u64p, ctypes.c_int, ctypes.c_size_t, ctypes.c_uint,
                              ctypes.c_uint64]),
    "cp_lde_dev": (ctypes.c_int, [_vp, _vp, ctypes.c_size_t, ctypes.c_int, ctypes.c_int,
                                  ctypes.c_size_t, ctypes.c_uint64, ctypes.c_uint, _vp,
                                  ctypes.c_size_t]),
    "cp_poseidon_permute_dev": (ctypes.c_int, [_vp, _vp, ctypes.c_size_t]),
    "cp_field_mul": (ctypes.c_int, [_vp, _u64p, _u64p, _u64p, ctypes.c_size_t]),
    "cp_poseidon_permute": (ctypes.c_int, [_vp, _u64p, ctypes.c_size_t]),
    "cp_hash_no_pad": (ctypes.c_int, [_vp, _u64p, ctypes.c_size_t, ctypes.c_size_t, _u64p]),
    "cp_two_to_one": (ctypes.c_int, [_vp, _u64p, _u64p, ctypes.c_size_t, _u64p]),
    "cp_merkle_cols_dev": (ctypes.c_int, [_vp, _vp, ctypes.c_size_t, ctypes.c_size_t,
                                          ctypes.c_size_t, ctypes.c_int, _vp, _vp]),
    "cp_merkle_cap": (ctypes.c_int, [_vp, _u64p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int,
                                     _u64p]),
    "cp_commit_dev": (ctypes.c_int, [_vp, _vp, ctypes.c_size_t, ctypes.c_int, ctypes.c_int,
                                     ctypes.c_int, _vp, _vp, _vp, _vp]),
    "cp_commit_batch_dev": (ctypes.c_int, [_vp, _vp, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int,
                                           ctypes.c_int, ctypes.c_int, _vp, _vp, _vp, _vp]),
}


class CityProverError(RuntimeError):
    pass


_lib = None


def load_library(build_if_missing=True):
    """dlopen the in-tree HIP library; raises if it is missing (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    so = _build.SO
    if not os.path.exists(so):
        if not build_if_missing:
            raise CityProverError(f"{so} not built; run __graft_entry__.build()")
        _build.build()
    lib = ctypes.CDLL(so)
    for name, (res, args) in ABI.items():
        f = getattr(lib, name)  # AttributeError if the symbol is not exported
        f.restype, f.argtypes = res, args
    _lib = lib
    return lib


def _ptr(a):
    assert isinstance(a, np.ndarray) and a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_u64p)


def _as_u64(x):
    return np.ascontiguousarray(np.asarray(x, dtype=np.uint64))


class DeviceBuffer:
    """A u64 buffer in HBM owned by a Prover context."""

    def __init__(self, prover, n_elems):
        self.prover, self.n = prover, int(n_elems)
        p = _vp()
        prover._check(prover.lib.cp_dev_alloc(prover.ctx, self.n * 8, ctypes.byref(p)))
        self.ptr = p.value

    def upload(self, arr):
        arr = _as_u64(arr)
        assert arr.size <= self.n
        self.prover._check(self.prover.lib.cp_h2d(self.prover.ctx, self.ptr, arr.ctypes.data, arr.size * 8))
        return self

    def download(self, n=None, offset=0):
        n = self.n - offset if n is None else int(n)
        out = np.empty(n, np.uint64)
        self.prover._check(self.prover.lib.cp_d2h(self.prover.ctx, out.ctypes.data, self.ptr + offset * 8, n * 8))
        return out

    def at(self, offset):
        return self.ptr + offset * 8

    def free(self):
        if self.ptr:
            self.prover._check(self.prover.lib.cp_dev_free(self.prover.ctx, self.ptr))
            self.ptr = None


class Prover:
    """One context per GPU (one consumer per device)."""

    def __init__(self, device=0):
        self.lib = load_library()
        if self.lib.cp_device_count() <= 0:
            raise CityProverError("no HIP device visible: cityprover has no CPU fallback")
        self.ctx = self.lib.cp_ctx_create(device)
        if not self.ctx:
            raise CityProverError(self.lib.cp_last_error(None).decode())
        self.device = device

    def close(self):
        if getattr(self, "ctx", None):
            self.free_pinned()
            self.lib.cp_ctx_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise CityProverError(f"[{rc}] " + self.lib.cp_last_error(self.ctx).decode())

    # ---- memory / timing -----------------------------------------------------------------
    def alloc(self, n_elems):
        return DeviceBuffer(self, n_elems)

    def set_lanes(self, lanes):
        """internal pipelining of cp_prove_batch_host for a single-threaded caller (cp_ctx_set_lanes)"""
        self._check(self.lib.cp_ctx_set_lanes(self.ctx, int(lanes)))

    def set_option(self, name, value):
        """cp_ctx_set_option: one of the CITYPROVER_* switches for this context only"""
        self._check(self.lib.cp_ctx_set_option(self.ctx, name.encode(), int(value)))

    def set_device_transcript(self, mode):
        """where the Fiat-Shamir transcripts are hashed: 1 device, 0 host, -1 by batch size (cp_ctx_set_device_transcript)"""
        self._check(self.lib.cp_ctx_set_device_transcript(self.ctx, int(mode)))

    def to_device(self, arr):
        arr = _as_u64(arr)
        return DeviceBuffer(self, arr.size).upload(arr)

    def pinned(self, arr):
        """Copy of `arr` (uint64) in page-locked host memory (cp_host_alloc); freed with the Prover or free_pinned()."""
        arr = _as_u64(arr)
        p = _vp()
        self._check(self.lib.cp_host_alloc(self.ctx, arr.size * 8, ctypes.byref(p)))
        out = np.ctypeslib.as_array(ctypes.cast(p, _u64p), shape=(arr.size,)).reshape(arr.shape)
        out[...] = arr
        self._pinned = getattr(self, "_pinned", [])
        self._pinned.append(p.value)
        return out

    def free_pinned(self):
        for p in getattr(self, "_pinned", []):
            self.lib.cp_host_free(self.ctx, p)
        self._pinned = []

    def sync(self):
        self._check(self.lib.cp_sync(self.ctx))

    def event(self):
        e = _vp()
        self._check(self.lib.cp_event_create(self.ctx, ctypes.byref(e)))
        return e.value

    def record(self, ev):
        self._check(self.lib.cp_event_record(self.ctx, ev))

    def elapsed_ms(self, e0, e1):
        ms = ctypes.c_float()
        self._check(self.lib.cp_event_elapsed_ms(self.ctx, e0, e1, ctypes.byref(ms)))
        return ms.value

    def profile_begin(self):
        self._check(self.lib.cp_profile_begin(self.ctx))

    def profile_end(self):
        import json
        buf = ctypes.create_string_buffer(1 << 16)
        self._check(self.lib.cp_profile_end(self.ctx, buf, len(buf)))
        return json.loads(buf.value.decode())

    # ---- device-resident entry points ----------------------------------------------------
    def ntt_dev(self, buf_ptr, log_n, batch=1, stride=None, flags=0, shift=0):
        stride = (1 << log_n) if stride is None else stride
        self._check(self.lib.cp_ntt_dev(self.ctx, buf_ptr, log_n, batch, stride, flags, shift))

    def lde_dev(self, coeffs_ptr, log_n, rate_bits, batch, out_ptr, shift=7, flags=NTT_BITREV_OUT,
                in_stride=None, out_stride=None):
        n = 1 << log_n
        self._check(self.lib.cp_lde_dev(self.ctx, coeffs_ptr, in_stride or n, log_n, rate_bits, batch,
                                        shift, flags, out_ptr, out_stride or (n << rate_bits)))

    def poseidon_permute_dev(self, states_ptr, count):
        self._check(self.lib.cp_poseidon_permute_dev(self.ctx, states_ptr, count))

    def merkle_cols_dev(self, cols_ptr, n_leaves, leaf_len, cap_height, cap_ptr, digests_ptr=None,
                        col_stride=None):
        self._check(self.lib.cp_merkle_cols_dev(self.ctx, cols_ptr, n_leaves, leaf_len,
                                                col_stride or n_leaves, cap_height, digests_ptr, cap_ptr))

    def commit_dev(self, values_ptr, k, log_n, rate_bits, cap_height, lde_ptr, cap_ptr,
                   coeffs_ptr=None, digests_ptr=None):
        self._check(self.lib.cp_commit_dev(self.ctx, values_ptr, k, log_n, rate_bits, cap_height,
                                           coeffs_ptr, lde_ptr, digests_ptr, cap_ptr))

    def commit_batch_dev(self, values_ptr, k, n_trees, log_n, rate_bits, cap_height, lde_ptr, caps_ptr,
                         coeffs_ptr=None, digests_ptr=None):
        self._check(self.lib.cp_commit_batch_dev(self.ctx, values_ptr, k, n_trees, log_n, rate_bits,
                                                 cap_height, coeffs_ptr, lde_ptr, digests_ptr, caps_ptr))

    # ---- host-array conveniences (numpy in / numpy out) -----------------------------------
    def ntt(self, a, flags=0, shift=0):
        a = _as_u64(a).copy()
        batch = 1 if a.ndim == 1 else a.shape[0]
        n = a.shape[-1]
        log_n = int(n).bit_length() - 1
        assert 1 << log_n == n
        self._check(self.lib.cp_ntt(self.ctx, _ptr(a), log_n, batch, flags, shift))
        return a

    def intt(self, a, flags=0, shift=0):
        return self.ntt(a, flags | NTT_INVERSE, shift)

    def lde(self, coeffs, rate_bits, shift=7, bitrev=False):
        c = _as_u64(coeffs)
        c2 = c.reshape(1, -1) if c.ndim == 1 else c
        k, n = c2.shape
        log_n = int(n).bit_length() - 1
        din, dout = self.to_device(c2), self.alloc(k * (n << rate_bits))
        try:
            self.lde_dev(din.ptr, log_n, rate_bits, k, dout.ptr, shift, NTT_BITREV_OUT if bitrev else 0)
            out = dout.download().reshape(k, n << rate_bits)
        finally:
            din.free()
            dout.free()
        return out[0] if c.ndim == 1 else out

    def field_mul(self, a, b):
        """a * b mod p element-wise with the device's multiplication; a, b: any u64 (lazy representatives allowed)."""
        x, y = _as_u64(a).ravel().copy(), _as_u64(b).ravel().copy()
        assert x.size == y.size
        out = np.zeros(x.size, np.uint64)
        self._check(self.lib.cp_field_mul(self.ctx, _ptr(x), _ptr(y), _ptr(out), x.size))
        return out

    def poseidon_permute(self, states):
        s = _as_u64(states).copy()
        assert s.size % 12 == 0
        self._check(self.lib.cp_poseidon_permute(self.ctx, _ptr(s), s.size // 12))
        return s

    def hash_no_pad(self, inputs):
        x = _as_u64(inputs)
        x2 = x.reshape(1, -1) if x.ndim == 1 else x
        out = np.zeros((x2.shape[0], 4), np.uint64)
        self._check(self.lib.cp_hash_no_pad(self.ctx, _ptr(x2), x2.shape[0], x2.shape[1], _ptr(out)))
        return out[0] if x.ndim == 1 else out

    def two_to_one(self, left, right):
        l, r = _as_u64(left), _as_u64(right)
        l2, r2 = l.reshape(-1, 4), r.reshape(-1, 4)
        out = np.zeros_like(l2)
        self._check(self.lib.cp_two_to_one(self.ctx, _ptr(l2), _ptr(r2), l2.shape[0], _ptr(out)))
        return out.reshape(l.shape)

    def merkle_cap(self, rows, cap_height):
        rows = _as_u64(rows)
        n, k = rows.shape
        cap = np.zeros((1 << cap_height, 4), np.uint64)
        self._check(self.lib.cp_merkle_cap(self.ctx, _ptr(rows), n, k, cap_height, _ptr(cap)))
        return cap

    def merkle_cols(self, cols, cap_height, want_digests=False):
        cols = _as_u64(cols)
        k, n = cols.shape
        dc, dcap = self.to_device(cols), self.alloc(4 << cap_height)
        nd = 2 * n - (2 << cap_height)
        dd = self.alloc(nd * 4) if want_digests and nd > 0 else None
        try:
            self.merkle_cols_dev(dc.ptr, n, k, cap_height, dcap.ptr, dd.ptr if dd else None)
            cap = dcap.download().reshape(-1, 4)
            dig = dd.download().reshape(-1, 4) if dd else None
        finally:
            dc.free()
            dcap.free()
            if dd:
                dd.free()
        return (cap, dig) if want_digests else cap

    def commit(self, values, rate_bits, cap_height, want=("coeffs", "lde", "cap")):
        v = _as_u64(values)
        k, n = v.shape
        log_n = int(n).bit_length() - 1
        N = n << rate_bits
        dv, dl, dcap = self.to_device(v), self.alloc(k * N), self.alloc(4 << cap_height)
        dco = self.alloc(k * n) if "coeffs" in want else None
        nd = 2 * N - (2 << cap_height)
        dd = self.alloc(nd * 4) if "digests" in want and nd > 0 else None
        try:
            self.commit_dev(dv.ptr, k, log_n, rate_bits, cap_height, dl.ptr, dcap.ptr,
                            dco.ptr if dco else None, dd.ptr if dd else None)
            res = {"cap": dcap.download().reshape(-1, 4)}
            res["lde"] = dl.download().reshape(k, N) if "lde" in want else None
            res["coeffs"] = dco.download().reshape(k, n) if dco else None
            res["digests"] = dd.download().reshape(-1, 4) if dd else None
        finally:
            for b in (dv, dl, dcap, dco, dd):
                if b:
                    b.free()
        return res


# ---- circuits and the proof tail ---------------------------------------------------------------
class Shape(ctypes.Structure):
    """cp_shape (include/cityprover.h)."""
    _fields_ = [(n, ctypes.c_int) for n in (
        "degree_bits", "num_constants", "num_routed_wires", "num_wires", "num_challenges",
        "num_partial_products", "quotient_degree_factor", "rate_bits", "cap_height", "pow_bits",
        "num_query_rounds", "n_arity")] + [("arity_bits", ctypes.c_int * 8), ("zero_knowledge", ctypes.c_int),
                                            ("num_public_inputs", ctypes.c_int)]


SALT_SIZE = 4


def standard_recursion_shape(**over):
    """CircuitConfig::standard_recursion_config() at degree 2^12, the only configuration the worker
    circuits use (SURVEY.md §5 'Config / flags', Appendix A)."""
    d = dict(degree_bits=12, num_constants=5, num_routed_wires=80, num_wires=135, num_challenges=2,
             num_partial_products=9, quotient_degree_factor=8, rate_bits=3, cap_height=4, pow_bits=16,
             num_query_rounds=28, arity_bits=(4, 4), zero_knowledge=0, num_public_inputs=0)
    d.update(over)
    ab = d.pop("arity_bits")
    s = Shape(**d, n_arity=len(ab))
    for i, a in enumerate(ab):
        s.arity_bits[i] = a
    return s


ABI.update({
    "cp_circuit_load": (_vp, [_vp, ctypes.POINTER(Shape), _u64p, _u64p, _u64p]),
    "cp_zs_partial_products_dev": (ctypes.c_int, [_vp, ctypes.c_size_t, ctypes.POINTER(_vp), _vp, _u64p, _u64p, _vp]),
    "cp_circuit_destroy": (None, [_vp]),
    "cp_circuit_cs_cap": (ctypes.c_int, [_vp, _u64p]),
    "cp_circuits_batch_compatible": (ctypes.c_int, [_vp, _vp]),
    "cp_prove_tail": (ctypes.c_int, [_vp, _u64p, ctypes.c_size_t, _vp, _vp, _vp, ctypes.c_int, ctypes.c_uint64,
                                     ctypes.POINTER(ctypes.POINTER(ctypes.c_uint8)),
                                     ctypes.POINTER(ctypes.c_size_t)]),
    "cp_free": (None, [_vp]),
})


class Circuit:
    """Device-resident circuit: the mirror of a built `CircuitData` (constants/sigmas commitment)."""

    def __init__(self, prover, shape, circuit_digest, cs_values, k_is=None):
        self.prover, self.shape = prover, shape
        cd, cs = _as_u64(circuit_digest), _as_u64(cs_values)
        n = 1 << shape.degree_bits
        assert cs.shape == (shape.num_constants + shape.num_routed_wires, n)
        ks = None if k_is is None else _as_u64(k_is)
        self.handle = prover.lib.cp_circuit_load(prover.ctx, ctypes.byref(shape), _ptr(cd), _ptr(cs),
                                                 None if ks is None else _ptr(ks))
        if not self.handle:
            raise CityProverError(prover.lib.cp_last_error(prover.ctx).decode())

    def cs_cap(self):
        cap = np.zeros((1 << self.shape.cap_height, 4), np.uint64)
        self.prover._check(self.prover.lib.cp_circuit_cs_cap(self.handle, _ptr(cap)))
        return cap

    def prove_tail(self, public_inputs, wires_values, zs_pp_values, quotient_coeffs, pow_override=None):
        """numpy in (uploaded here), bincode ProofWithPublicInputs bytes out."""
        p = self.prover
        pi = _as_u64(public_inputs)
        bufs = [p.to_device(_as_u64(a)) for a in (wires_values, zs_pp_values, quotient_coeffs)]
        try:
            return self.prove_tail_dev(pi, bufs[0].ptr, bufs[1].ptr, bufs[2].ptr, pow_override)
        finally:
            for b in bufs:
                b.free()

    def prove_tail_dev(self, public_inputs, wires_ptr, zs_pp_ptr, quotient_ptr, pow_override=None):
        p = self.prover
        pi = _as_u64(public_inputs)
        out = ctypes.POINTER(ctypes.c_uint8)()
        ln = ctypes.c_size_t()
        p._check(p.lib.cp_prove_tail(self.handle, _ptr(pi) if pi.size else None, pi.size, wires_ptr, zs_pp_ptr,
                                     quotient_ptr, 0 if pow_override is None else 1, pow_override or 0,
                                     ctypes.byref(out), ctypes.byref(ln)))
        data = ctypes.string_at(out, ln.value)
        p.lib.cp_free(out)
        return data

    def close(self):
        if self.handle:
            self.prover.lib.cp_circuit_destroy(self.handle)
            self.handle = None


ABI["cp_prove_tail_batch"] = (ctypes.c_int, [_vp, ctypes.c_size_t, ctypes.POINTER(_vp), ctypes.POINTER(_u64p),
                                             ctypes.POINTER(ctypes.c_size_t), _vp, _vp, _vp,
                                             ctypes.POINTER(ctypes.c_int), _u64p,
                                             ctypes.POINTER(ctypes.POINTER(ctypes.c_uint8)),
                                             ctypes.POINTER(ctypes.c_size_t)])


def prove_tail_batch_dev(prover, circuits, public_inputs, wires_ptr, zs_pp_ptr, quotient_ptr, pow_overrides=None):
    """circuits: list of Circuit (same shape); public_inputs: list of u64 sequences; device pointers hold
    [proof][poly][n]. Returns a list of proof byte strings."""
    B = len(circuits)
    cs = (_vp * B)(*[c.handle for c in circuits])
    pis = [_as_u64(p) for p in public_inputs]
    pi_ptrs = (_u64p * B)(*[_ptr(p) if p.size else None for p in pis])
    n_pis = (ctypes.c_size_t * B)(*[p.size for p in pis])
    use = (ctypes.c_int * B)(*[0 if (pow_overrides is None or pow_overrides[i] is None) else 1 for i in range(B)])
    ov = np.array([0 if (pow_overrides is None or pow_overrides[i] is None) else pow_overrides[i] for i in range(B)],
                  dtype=np.uint64)
    outs = (ctypes.POINTER(ctypes.c_uint8) * B)()
    lens = (ctypes.c_size_t * B)()
    prover._check(prover.lib.cp_prove_tail_batch(prover.ctx, B, cs, pi_ptrs, n_pis, wires_ptr, zs_pp_ptr, quotient_ptr,
                                                 use, _ptr(ov), outs, lens))
    res = []
    for i in range(B):
        res.append(ctypes.string_at(outs[i], lens[i]))
        prover.lib.cp_free(outs[i])
    return res


def zs_partial_products_dev(prover, circuits, wires_ptr, betas, gammas, out_ptr):
    """A7 for a batch: betas/gammas are (B, num_challenges) arrays; device buffers as in the header."""
    B = len(circuits)
    cs = (_vp * B)(*[c.handle for c in circuits])
    b, g = _as_u64(betas).reshape(B, -1), _as_u64(gammas).reshape(B, -1)
    prover._check(prover.lib.cp_zs_partial_products_dev(prover.ctx, B, cs, wires_ptr, _ptr(b), _ptr(g), out_ptr))


# ---- gates and the whole proof ------------------------------------------------------------------------
GATE_NOOP, GATE_CONSTANT, GATE_PUBLIC_INPUT, GATE_ARITHMETIC, GATE_POSEIDON = 0, 1, 2, 3, 4
GATE_COMPARISON, GATE_U32_ARITHMETIC, GATE_U32_RANGE_CHECK = 5, 6, 7
GATE_U32_ADD_MANY, GATE_U32_SUBTRACTION, GATE_U32_INTERLEAVE, GATE_UNINTERLEAVE_TO_U32, GATE_UNINTERLEAVE_TO_B32 = 8, 9, 10, 11, 12
(GATE_ARITHMETIC_EXT, GATE_MUL_EXT, GATE_BASE_SUM, GATE_RANDOM_ACCESS, GATE_REDUCING, GATE_REDUCING_EXT, GATE_POSEIDON_MDS,
 GATE_COSET_INTERPOLATION) = 13, 14, 15, 16, 17, 18, 19, 20
GATE_EXPONENTIATION = 21


class Gate(ctypes.Structure):
    """cp_gate"""
    _fields_ = [(n, ctypes.c_int) for n in ("type", "selector_index", "group_start", "group_end", "param", "param2", "param3")]


ABI["cp_circuit_set_gates"] = (ctypes.c_int, [_vp, ctypes.POINTER(Gate), ctypes.c_size_t, ctypes.c_int])
ABI["cp_prove_batch"] = (ctypes.c_int, [_vp, ctypes.c_size_t, ctypes.POINTER(_vp), ctypes.POINTER(_u64p),
                                        ctypes.POINTER(ctypes.c_size_t), _vp, ctypes.POINTER(ctypes.c_int), _u64p,
                                        ctypes.POINTER(ctypes.POINTER(ctypes.c_uint8)),
                                        ctypes.POINTER(ctypes.c_size_t)])


def set_gates(circuit, gate_list, num_selectors):
    """gate_list: [(type, selector_index, group_start, group_end, param[, param2[, param3]])] in gate-index order."""
    arr = (Gate * len(gate_list))(*[Gate(*(tuple(g) + (0,) * (7 - len(g)))) for g in gate_list])
    circuit.prover._check(circuit.prover.lib.cp_circuit_set_gates(circuit.handle, arr, len(gate_list), num_selectors))


def prove_batch_dev(prover, circuits, public_inputs, wires_ptr, pow_overrides=None):
    """wires -> proofs for a batch (A7 + A8 + tail). wires_ptr: device [proof][num_wires][n]."""
    B = len(circuits)
    cs = (_vp * B)(*[c.handle for c in circuits])
    pis = [_as_u64(p) for p in public_inputs]
    pi_ptrs = (_u64p * B)(*[_ptr(p) if p.size else None for p in pis])
    n_pis = (ctypes.c_size_t * B)(*[p.size for p in pis])
    use = (ctypes.c_int * B)(*[0 if (pow_overrides is None or pow_overrides[i] is None) else 1 for i in range(B)])
    ov = np.array([0 if (pow_overrides is None or pow_overrides[i] is None) else pow_overrides[i] for i in range(B)],
                  dtype=np.uint64)
    outs = (ctypes.POINTER(ctypes.c_uint8) * B)()
    lens = (ctypes.c_size_t * B)()
    prover._check(prover.lib.cp_prove_batch(prover.ctx, B, cs, pi_ptrs, n_pis, wires_ptr, use, _ptr(ov), outs, lens))
    res = []
    for i in range(B):
        res.append(ctypes.string_at(outs[i], lens[i]))
        prover.lib.cp_free(outs[i])
    return res


ABI["cp_prove_batch_host"] = (ctypes.c_int, [_vp, ctypes.c_size_t, ctypes.POINTER(_vp), ctypes.POINTER(_u64p),
                                             ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(_u64p),
                                             ctypes.POINTER(ctypes.c_int), _u64p,
                                             ctypes.POINTER(ctypes.POINTER(ctypes.c_uint8)),
                                             ctypes.POINTER(ctypes.c_size_t)])
ABI["cp_prove"] = (ctypes.c_int, [_vp, _u64p, _u64p, ctypes.c_size_t, ctypes.c_int, ctypes.c_uint64,
                                  ctypes.POINTER(ctypes.POINTER(ctypes.c_uint8)), ctypes.POINTER(ctypes.c_size_t)])


def prove_batch(prover, circuits, public_inputs, wires, pow_overrides=None):
    """Host-memory variant (cp_prove_batch_host): wires = list of [num_wires][n] uint64 arrays (the witness generator's
    output); the library copies them to the device."""
    B = len(circuits)
    cs = (_vp * B)(*[c.handle for c in circuits])
    pis = [_as_u64(p) for p in public_inputs]
    pi_ptrs = (_u64p * B)(*[_ptr(p) if p.size else None for p in pis])
    n_pis = (ctypes.c_size_t * B)(*[p.size for p in pis])
    ws = [_as_u64(w) for w in wires]
    for c, w in zip(circuits, ws):   # the C ABI takes bare pointers: sizes are checked here
        if w.size != c.shape.num_wires << c.shape.degree_bits:
            raise ValueError("wires must be [num_wires][n] = %d x %d values" % (c.shape.num_wires, 1 << c.shape.degree_bits))
    w_ptrs = (_u64p * B)(*[_ptr(w) for w in ws])
    use = (ctypes.c_int * B)(*[0 if (pow_overrides is None or pow_overrides[i] is None) else 1 for i in range(B)])
    ov = np.array([0 if (pow_overrides is None or pow_overrides[i] is None) else pow_overrides[i] for i in range(B)],
                  dtype=np.uint64)
    outs = (ctypes.POINTER(ctypes.c_uint8) * B)()
    lens = (ctypes.c_size_t * B)()
    prover._check(prover.lib.cp_prove_batch_host(prover.ctx, B, cs, pi_ptrs, n_pis, w_ptrs, use, _ptr(ov), outs, lens))
    res = []
    for i in range(B):
        res.append(ctypes.string_at(outs[i], lens[i]))
        prover.lib.cp_free(outs[i])
    return res


ABI["cp_prove_batch_zk_host"] = (ctypes.c_int, [_vp, ctypes.c_size_t, ctypes.POINTER(_vp), ctypes.POINTER(_u64p),
                                                ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(_u64p), ctypes.POINTER(_u64p),
                                                ctypes.POINTER(ctypes.c_int), _u64p,
                                                ctypes.POINTER(ctypes.POINTER(ctypes.c_uint8)),
                                                ctypes.POINTER(ctypes.c_size_t)])


def prove_batch_zk(prover, circuits, public_inputs, wires, salts, pow_overrides=None):
    """Zero-knowledge circuits: salts[p] = [3][SALT_SIZE][N] uint64 random field elements (cp_prove_batch_zk_host)."""
    B = len(circuits)
    cs = (_vp * B)(*[c.handle for c in circuits])
    pis = [_as_u64(p) for p in public_inputs]
    pi_ptrs = (_u64p * B)(*[_ptr(p) if p.size else None for p in pis])
    n_pis = (ctypes.c_size_t * B)(*[p.size for p in pis])
    ws, sl = [_as_u64(w) for w in wires], [_as_u64(x) for x in salts]
    for c, w, x in zip(circuits, ws, sl):   # the C ABI takes bare pointers: sizes are checked here
        if w.size != c.shape.num_wires << c.shape.degree_bits:
            raise ValueError("wires must be [num_wires][n]")
        if x.size != 3 * SALT_SIZE << (c.shape.degree_bits + c.shape.rate_bits):
            raise ValueError("salts must be [3][SALT_SIZE][N]")
    w_ptrs = (_u64p * B)(*[_ptr(w) for w in ws])
    s_ptrs = (_u64p * B)(*[_ptr(x) for x in sl])
    use = (ctypes.c_int * B)(*[0 if (pow_overrides is None or pow_overrides[i] is None) else 1 for i in range(B)])
    ov = np.array([0 if (pow_overrides is None or pow_overrides[i] is None) else pow_overrides[i] for i in range(B)],
                  dtype=np.uint64)
    outs = (ctypes.POINTER(ctypes.c_uint8) * B)()
    lens = (ctypes.c_size_t * B)()
    prover._check(prover.lib.cp_prove_batch_zk_host(prover.ctx, B, cs, pi_ptrs, n_pis, w_ptrs, s_ptrs, use, _ptr(ov), outs, lens))
    res = []
    for i in range(B):
        res.append(ctypes.string_at(outs[i], lens[i]))
        prover.lib.cp_free(outs[i])
    return res


def prove(circuit, wires, public_inputs, pow_override=None):
    """cp_prove: CircuitData::prove after witness generation for ONE proof, wires in host memory."""
    pr = circuit.prover
    w, pi = _as_u64(wires), _as_u64(public_inputs)
    if w.size != circuit.shape.num_wires << circuit.shape.degree_bits:
        raise ValueError("wires must be [num_wires][n]")
    out = ctypes.POINTER(ctypes.c_uint8)()
    n = ctypes.c_size_t()
    pr._check(pr.lib.cp_prove(circuit.handle, _ptr(w), _ptr(pi) if pi.size else None, pi.size,
                              0 if pow_override is None else 1, 0 if pow_override is None else pow_override,
                              ctypes.byref(out), ctypes.byref(n)))
    res = ctypes.string_at(out, n.value)
    pr.lib.cp_free(out)
    return res


class BatcherStats(ctypes.Structure):
    """cp_batcher_stats"""
    _fields_ = [("calls", ctypes.c_uint64), ("batches", ctypes.c_uint64), ("proofs", ctypes.c_uint64),
                ("largest_batch", ctypes.c_uint64), ("retried_singly", ctypes.c_uint64)]


ABI["cp_batcher_create"] = (_vp, [_vp, ctypes.c_size_t, ctypes.c_uint])
ABI["cp_batcher_prove"] = (ctypes.c_int, [_vp, _vp, _u64p, _u64p, ctypes.c_size_t, ctypes.c_int, ctypes.c_uint64,
                                          ctypes.POINTER(ctypes.POINTER(ctypes.c_uint8)), ctypes.POINTER(ctypes.c_size_t)])
ABI["cp_batcher_get_stats"] = (ctypes.c_int, [_vp, ctypes.POINTER(BatcherStats)])
ABI["cp_batcher_destroy"] = (None, [_vp])


class Batcher:
    """cp_batcher: merges the concurrent one-proof calls of several threads (the reference's worker loops,
    city_rollup_core_worker/src/actors/simple.rs:32-56) into cp_prove_batch_host launches. `prove` is thread-safe
    and blocking (ctypes releases the GIL for the call)."""

    def __init__(self, prover, max_batch=32, linger_us=0):
        self.prover = prover
        self.handle = prover.lib.cp_batcher_create(prover.ctx, max_batch, linger_us)
        if not self.handle:
            raise CityProverError(prover.lib.cp_last_error(None).decode())

    def prove(self, circuit, wires, public_inputs, pow_override=None):
        lib = self.prover.lib
        w, pi = _as_u64(wires), _as_u64(public_inputs)
        if w.size != circuit.shape.num_wires << circuit.shape.degree_bits:
            raise ValueError("wires must be [num_wires][n]")
        out = ctypes.POINTER(ctypes.c_uint8)()
        n = ctypes.c_size_t()
        rc = lib.cp_batcher_prove(self.handle, circuit.handle, _ptr(w), _ptr(pi) if pi.size else None, pi.size,
                                  0 if pow_override is None else 1, 0 if pow_override is None else pow_override,
                                  ctypes.byref(out), ctypes.byref(n))
        if rc != 0:  # the message is the calling thread's, not the context's
            raise CityProverError(f"[{rc}] " + lib.cp_last_error(None).decode())
        res = ctypes.string_at(out, n.value)
        lib.cp_free(out)
        return res

    def stats(self):
        st = BatcherStats()
        self.prover._check(self.prover.lib.cp_batcher_get_stats(self.handle, ctypes.byref(st)))
        return {k: int(getattr(st, k)) for k, _ in BatcherStats._fields_}

    def close(self):
        if self.handle:
            self.prover.lib.cp_batcher_destroy(self.handle)
            self.handle = None


ABI["cp_verify"] = (ctypes.c_int, [_vp, ctypes.c_char_p, ctypes.c_size_t])


def verify(circuit, proof_bytes):
    """CircuitData::verify. Returns None when the proof is accepted; raises CityProverError naming the
    first failing check otherwise."""
    circuit.prover._check(circuit.prover.lib.cp_verify(circuit.handle, proof_bytes, len(proof_bytes)))


# ---- BLS12-381 G1 MSM -------------------------------------------------------------------
_u8p = ctypes.POINTER(ctypes.c_uint8)
ABI["cp_msm_bls12381_g1"] = (ctypes.c_int, [_vp, _u64p, _u64p, _u8p, ctypes.c_size_t, _u64p, ctypes.POINTER(ctypes.c_int)])
ABI["cp_msm_bls12381_g1_prepare_dev"] = (ctypes.c_int, [_vp, _vp, ctypes.c_size_t, _vp])
ABI["cp_msm_bls12381_g1_dev"] = (ctypes.c_int, [_vp, _vp, _vp, _vp, ctypes.c_size_t, _u64p, ctypes.POINTER(ctypes.c_int)])
ABI["cp_msm_bls12381_g1_synthetic_points_dev"] = (ctypes.c_int, [_vp, _u64p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_size_t, _vp])
G1_AFFINE_BYTES = 112


def _g1_out(xy, inf):
    if inf.value:
        return None
    return (sum(int(v) << (64 * i) for i, v in enumerate(xy[:6])), sum(int(v) << (64 * i) for i, v in enumerate(xy[6:])))


def msm_g1(prover, scalars, points_xy, points_inf=None):
    """sum_i scalars[i] * points[i] on BLS12-381 G1. scalars: (n, 4) uint64 LE limbs; points_xy: (n, 12) uint64 affine
    canonical x || y; points_inf: optional (n,) uint8. Returns None (infinity) or (x, y) Python ints."""
    s, p = _as_u64(scalars).reshape(-1, 4), _as_u64(points_xy).reshape(-1, 12)
    if s.shape[0] != p.shape[0]:
        raise ValueError("scalars and points differ in count")
    n = s.shape[0]
    inf = None if points_inf is None else np.ascontiguousarray(points_inf, dtype=np.uint8)
    out, oi = np.zeros(12, np.uint64), ctypes.c_int()
    prover._check(prover.lib.cp_msm_bls12381_g1(prover.ctx, _ptr(s) if n else None, _ptr(p) if n else None,
                                                None if inf is None else inf.ctypes.data_as(_u8p), n, _ptr(out),
                                                ctypes.byref(oi)))
    return _g1_out(out, oi)


class G1Points:
    """A fixed point set resident on the device in the library's internal form (a proving key)."""

    def __init__(self, prover, points_xy):
        p = _as_u64(points_xy).reshape(-1, 12)
        self.prover, self.n = prover, p.shape[0]
        raw = prover.to_device(p)
        self.buf = prover.alloc(self.n * G1_AFFINE_BYTES // 8)
        prover._check(prover.lib.cp_msm_bls12381_g1_prepare_dev(prover.ctx, raw.ptr, self.n, self.buf.ptr))
        prover.sync()
        raw.free()

    @classmethod
    def synthetic(cls, prover, generator, a, b, n):
        """P_i = (a*i + b) * generator, built on the device (bench / large tests)."""
        self = cls.__new__(cls)
        self.prover, self.n = prover, n
        self.buf = prover.alloc(n * G1_AFFINE_BYTES // 8)
        g = np.array([(int(generator[h]) >> (64 * i)) & (2**64 - 1) for h in range(2) for i in range(6)], dtype=np.uint64)
        prover._check(prover.lib.cp_msm_bls12381_g1_synthetic_points_dev(prover.ctx, _ptr(g), a, b, n, self.buf.ptr))
        prover.sync()
        return self

    def msm_dev(self, scalars_ptr):
        out, oi = np.zeros(12, np.uint64), ctypes.c_int()
        self.prover._check(self.prover.lib.cp_msm_bls12381_g1_dev(self.prover.ctx, scalars_ptr, self.buf.ptr, None, self.n,
                                                                  _ptr(out), ctypes.byref(oi)))
        return _g1_out(out, oi)

    def free(self):
        self.buf.free()


# ---- G2 (coordinates c0 + c1*u in F_p^2; a point is 24 uint64: x.c0, x.c1, y.c0, y.c1) ----
ABI["cp_msm_bls12381_g2"] = (ctypes.c_int, [_vp, _u64p, _u64p, _u8p, ctypes.c_size_t, _u64p, ctypes.POINTER(ctypes.c_int)])
ABI["cp_msm_bls12381_g2_prepare_dev"] = (ctypes.c_int, [_vp, _vp, ctypes.c_size_t, _vp])
ABI["cp_msm_bls12381_g2_synthetic_points_dev"] = (ctypes.c_int, [_vp, _u64p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_size_t, _vp])
ABI["cp_msm_bls12381_g2_dev"] = (ctypes.c_int, [_vp, _vp, _vp, _vp, ctypes.c_size_t, _u64p, ctypes.POINTER(ctypes.c_int)])
G2_AFFINE_BYTES = 224


def _g2_out(xy, inf):
    if inf.value:
        return None
    v = [sum(int(x) << (64 * i) for i, x in enumerate(xy[6 * k:6 * k + 6])) for k in range(4)]
    return ((v[0], v[1]), (v[2], v[3]))


def msm_g2(prover, scalars, points_xy, points_inf=None):
    """G2 MSM. points_xy: (n, 24) uint64. Returns None or ((x0, x1), (y0, y1))."""
    s, p = _as_u64(scalars).reshape(-1, 4), _as_u64(points_xy).reshape(-1, 24)
    if s.shape[0] != p.shape[0]:
        raise ValueError("scalars and points differ in count")
    n = s.shape[0]
    inf = None if points_inf is None else np.ascontiguousarray(points_inf, dtype=np.uint8)
    out, oi = np.zeros(24, np.uint64), ctypes.c_int()
    prover._check(prover.lib.cp_msm_bls12381_g2(prover.ctx, _ptr(s) if n else None, _ptr(p) if n else None,
                                                None if inf is None else inf.ctypes.data_as(_u8p), n, _ptr(out),
                                                ctypes.byref(oi)))
    return _g2_out(out, oi)


class G2Points:
    """A fixed G2 point set resident on the device in the library's internal form."""

    def __init__(self, prover, points_xy):
        p = _as_u64(points_xy).reshape(-1, 24)
        self.prover, self.n = prover, p.shape[0]
        raw = prover.to_device(p)
        self.buf = prover.alloc(self.n * G2_AFFINE_BYTES // 8)
        prover._check(prover.lib.cp_msm_bls12381_g2_prepare_dev(prover.ctx, raw.ptr, self.n, self.buf.ptr))
        prover.sync()
        raw.free()

    @classmethod
    def synthetic(cls, prover, generator, a, b, n):
        self = cls.__new__(cls)
        self.prover, self.n = prover, n
        self.buf = prover.alloc(n * G2_AFFINE_BYTES // 8)
        (x0, x1), (y0, y1) = generator
        g = np.array([(int(c) >> (64 * i)) & (2**64 - 1) for c in (x0, x1, y0, y1) for i in range(6)], dtype=np.uint64)
        prover._check(prover.lib.cp_msm_bls12381_g2_synthetic_points_dev(prover.ctx, _ptr(g), a, b, n, self.buf.ptr))
        prover.sync()
        return self

    def msm_dev(self, scalars_ptr):
        out, oi = np.zeros(24, np.uint64), ctypes.c_int()
        self.prover._check(self.prover.lib.cp_msm_bls12381_g2_dev(self.prover.ctx, scalars_ptr, self.buf.ptr, None, self.n,
                                                                  _ptr(out), ctypes.byref(oi)))
        return _g2_out(out, oi)

    def free(self):
        self.buf.free()


# ---- NTT over the BLS12-381 scalar field ----
ABI["cp_ntt_bls12381_fr"] = (ctypes.c_int, [_vp, _u64p, ctypes.c_int, ctypes.c_uint, _u64p])
ABI["cp_ntt_bls12381_fr_dev"] = (ctypes.c_int, [_vp, _vp, ctypes.c_int, ctypes.c_uint, _u64p])
ABI["cp_groth16_quotient_bls12381_dev"] = (ctypes.c_int, [_vp, _vp, _vp, _vp, ctypes.c_int])
ABI["cp_groth16_quotient_bls12381"] = (ctypes.c_int, [_vp, _u64p, _u64p, _u64p, ctypes.c_int])


def fr_ntt(prover, values, inverse=False, shift=None):
    """values: (n, 4) uint64 canonical F_r elements -> transformed copy (natural order). shift: coset shift (int)."""
    a = _as_u64(values).reshape(-1, 4).copy()
    log_n = int(a.shape[0]).bit_length() - 1
    flags = (NTT_INVERSE if inverse else 0) | (NTT_COSET if shift is not None else 0)
    sh = None if shift is None else np.array([(int(shift) >> (64 * j)) & (2**64 - 1) for j in range(4)], dtype=np.uint64)
    prover._check(prover.lib.cp_ntt_bls12381_fr(prover.ctx, _ptr(a), log_n, flags, None if sh is None else _ptr(sh)))
    return a


def fr_ntt_dev(prover, data_ptr, log_n, inverse=False, shift=None):
    flags = (NTT_INVERSE if inverse else 0) | (NTT_COSET if shift is not None else 0)
    sh = None if shift is None else np.array([(int(shift) >> (64 * j)) & (2**64 - 1) for j in range(4)], dtype=np.uint64)
    prover._check(prover.lib.cp_ntt_bls12381_fr_dev(prover.ctx, data_ptr, log_n, flags, None if sh is None else _ptr(sh)))


def groth16_quotient(prover, a, b, c):
    """a, b, c: (n, 4) uint64 canonical F_r evaluations of (A w), (B w), (C w) on <omega_n> -> the n coefficients of
    h = (a b - c) / (x^n - 1) (cp_groth16_quotient_bls12381)."""
    a = _as_u64(a).reshape(-1, 4).copy()
    b = np.ascontiguousarray(_as_u64(b).reshape(-1, 4))
    c = np.ascontiguousarray(_as_u64(c).reshape(-1, 4))
    if not (a.shape == b.shape == c.shape):
        raise ValueError("a, b, c must have the same length")
    log_n = int(a.shape[0]).bit_length() - 1
    prover._check(prover.lib.cp_groth16_quotient_bls12381(prover.ctx, _ptr(a), _ptr(b), _ptr(c), log_n))
    return a


def groth16_quotient_dev(prover, a_ptr, b_ptr, c_ptr, log_n):
    prover._check(prover.lib.cp_groth16_quotient_bls12381_dev(prover.ctx, a_ptr, b_ptr, c_ptr, log_n))


class Groth16Pk(ctypes.Structure):
    _fields_ = [("n_wires", ctypes.c_size_t), ("n_private", ctypes.c_size_t), ("log_domain", ctypes.c_int),
                ("a_g1", _vp), ("b_g1", _vp), ("b_g2", _vp), ("k_g1", _vp), ("z_g1", _vp), ("a_inf", _vp), ("b_inf", _vp),
                ("alpha_g1", ctypes.c_uint64 * 12), ("beta_g1", ctypes.c_uint64 * 12), ("delta_g1", ctypes.c_uint64 * 12),
                ("beta_g2", ctypes.c_uint64 * 24), ("delta_g2", ctypes.c_uint64 * 24)]


ABI["cp_groth16_prove_bls12381"] = (ctypes.c_int, [_vp, ctypes.POINTER(Groth16Pk), _vp, _vp, _vp, _vp, _u64p, _u64p,
                                                   _u64p, _u64p, _u64p])


def groth16_prove(prover, pk, witness_ptr, a_ptr, b_ptr, c_ptr, r, s):
    """cp_groth16_prove_bls12381: pk = Groth16Pk; device pointers; r, s ints. Returns (A, B, C) as coordinate tuples of ints:
    A, C = (x, y); B = ((x0, x1), (y0, y1))."""
    lim = lambda v: np.array([(int(v) >> (64 * j)) & (2**64 - 1) for j in range(4)], dtype=np.uint64)
    oa, ob, oc = np.zeros(12, np.uint64), np.zeros(24, np.uint64), np.zeros(12, np.uint64)
    rr, ss = lim(r), lim(s)
    prover._check(prover.lib.cp_groth16_prove_bls12381(prover.ctx, ctypes.byref(pk), witness_ptr, a_ptr, b_ptr, c_ptr, _ptr(rr), _ptr(ss),
                                                       _ptr(oa), _ptr(ob), _ptr(oc)))
    val = lambda a: sum(int(v) << (64 * j) for j, v in enumerate(a))
    return ((val(oa[:6]), val(oa[6:])), ((val(ob[:6]), val(ob[6:12])), (val(ob[12:18]), val(ob[18:]))), (val(oc[:6]), val(oc[6:])))


# ---- circuit files (N1) and the stored Groth16 proof form ----
ABI["cp_circuit_load_file"] = (_vp, [_vp, ctypes.c_char_p])
ABI["cp_circuit_save_file"] = (ctypes.c_int, [_vp, ctypes.c_char_p])
ABI["cp_circuit_file_info"] = (ctypes.c_int, [ctypes.c_char_p, ctypes.POINTER(Shape), _u64p, ctypes.POINTER(ctypes.c_size_t),
                                              ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_uint)])
ABI["cp_circuit_set_public_input_targets"] = (ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_uint32), ctypes.c_size_t])
ABI["cp_circuit_public_inputs_from_wires"] = (ctypes.c_int, [_vp, _u64p, _u64p])
ABI["cp_circuit_shape"] = (ctypes.c_int, [_vp, ctypes.POINTER(Shape), _u64p])
ABI["cp_groth16_proof_pack_city"] = (ctypes.c_int, [_u64p, _u64p, _u64p, _u8p])
ABI["cp_groth16_proof_unpack_city"] = (ctypes.c_int, [_u8p, _u64p, _u64p, _u64p])


def circuit_file_info(path):
    """cp_circuit_file_info: parse + validate a .cpcirc file on the host (no GPU). Returns a dict; raises on a bad file."""
    lib = load_library()
    sh, dg = Shape(), np.zeros(4, np.uint64)
    ng, ns, fl = ctypes.c_size_t(), ctypes.c_int(), ctypes.c_uint()
    rc = lib.cp_circuit_file_info(os.fsencode(path), ctypes.byref(sh), _ptr(dg), ctypes.byref(ng), ctypes.byref(ns), ctypes.byref(fl))
    if rc != 0:
        raise CityProverError(f"[{rc}] " + lib.cp_last_error(None).decode())
    return dict(shape=sh, digest=[int(x) for x in dg], n_gates=ng.value, num_selectors=ns.value, flags=fl.value)


def load_circuit_file(prover, path):
    """cp_circuit_load_file -> Circuit"""
    h = prover.lib.cp_circuit_load_file(prover.ctx, os.fsencode(path))
    if not h:
        raise CityProverError(prover.lib.cp_last_error(prover.ctx).decode())
    c = Circuit.__new__(Circuit)
    c.prover, c.handle, c.shape = prover, h, Shape()
    prover._check(prover.lib.cp_circuit_shape(h, ctypes.byref(c.shape), None))
    return c


def save_circuit_file(circuit, path):
    circuit.prover._check(circuit.prover.lib.cp_circuit_save_file(circuit.handle, os.fsencode(path)))


def set_public_input_targets(circuit, row_wire_pairs):
    t = np.ascontiguousarray(np.asarray(row_wire_pairs, dtype=np.uint32).reshape(-1, 2))
    circuit.prover._check(circuit.prover.lib.cp_circuit_set_public_input_targets(
        circuit.handle, t.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)), t.shape[0]))


def public_inputs_from_wires(circuit, wires):
    w = _as_u64(wires)
    out = np.zeros(circuit.shape.num_public_inputs, np.uint64)
    circuit.prover._check(circuit.prover.lib.cp_circuit_public_inputs_from_wires(circuit.handle, _ptr(w), _ptr(out) if out.size else None))
    return out


def _limbs6(v):
    return [(int(v) >> (64 * i)) & (2**64 - 1) for i in range(6)]


def groth16_pack_city(A, B, C):
    """(A, B, C) as groth16_prove returns them -> the 192 bytes of CityGroth16ProofData (pi_a | pi_b_a0 | pi_b_a1 | pi_c)."""
    lib = load_library()
    a = np.array(_limbs6(A[0]) + _limbs6(A[1]), dtype=np.uint64)
    b = np.array(_limbs6(B[0][0]) + _limbs6(B[0][1]) + _limbs6(B[1][0]) + _limbs6(B[1][1]), dtype=np.uint64)
    c = np.array(_limbs6(C[0]) + _limbs6(C[1]), dtype=np.uint64)
    out = np.zeros(192, np.uint8)
    rc = lib.cp_groth16_proof_pack_city(_ptr(a), _ptr(b), _ptr(c), out.ctypes.data_as(_u8p))
    if rc != 0:
        raise CityProverError(f"[{rc}] " + lib.cp_last_error(None).decode())
    return out.tobytes()


def groth16_unpack_city(data):
    lib = load_library()
    buf = np.frombuffer(bytes(data), dtype=np.uint8).copy()
    if buf.size != 192:
        raise ValueError("CityGroth16ProofData is 192 bytes")
    a, b, c = np.zeros(12, np.uint64), np.zeros(24, np.uint64), np.zeros(12, np.uint64)
    rc = lib.cp_groth16_proof_unpack_city(buf.ctypes.data_as(_u8p), _ptr(a), _ptr(b), _ptr(c))
    if rc != 0:
        raise CityProverError(f"[{rc}] " + lib.cp_last_error(None).decode())
    val = lambda x: sum(int(v) << (64 * j) for j, v in enumerate(x))
    return ((val(a[:6]), val(a[6:])), ((val(b[:6]), val(b[6:12])), (val(b[12:18]), val(b[18:]))), (val(c[:6]), val(c[6:])))


# ---- FRI primitives and the verifier's audit hook ----
ABI["cp_verify_fri_queries_with_challenges"] = (ctypes.c_int, [ctypes.POINTER(Shape), ctypes.c_char_p, ctypes.c_size_t, _u64p, _u64p, _u64p, _u64p])
ABI["cp_fri_combine_dev"] = (ctypes.c_int, [_vp, _vp, ctypes.c_size_t, ctypes.c_size_t, _u64p, _vp])
ABI["cp_fri_fold_dev"] = (ctypes.c_int, [_vp, _vp, ctypes.c_size_t, ctypes.c_int, _u64p, _vp])


def verify_fri_queries_with_challenges(shape, proof_bytes, alpha, zeta, fri_betas, x_indices):
    """cp_verify's query phase with supplied challenges (no circuit, no GPU). Raises CityProverError on rejection."""
    lib = load_library()
    a, z = _as_u64(alpha), _as_u64(zeta)
    fb = _as_u64(fri_betas).reshape(-1)
    xi = _as_u64(x_indices)
    rc = lib.cp_verify_fri_queries_with_challenges(ctypes.byref(shape), proof_bytes, len(proof_bytes), _ptr(a), _ptr(z),
                                                   _ptr(fb) if fb.size else None, _ptr(xi))
    if rc != 0:
        raise CityProverError(f"[{rc}] " + lib.cp_last_error(None).decode())


def fri_combine(prover, polys, alpha):
    """sum_j alpha^j polys[j] over F_p^2 on the GPU (cp_fri_combine_dev). polys: (k, n) uint64 -> (n, 2) uint64."""
    f = _as_u64(polys)
    k, n = f.shape
    din, dout = prover.to_device(f), prover.alloc(2 * n)
    try:
        a = _as_u64(alpha)
        prover._check(prover.lib.cp_fri_combine_dev(prover.ctx, din.ptr, k, n, _ptr(a), dout.ptr))
        return dout.download().reshape(n, 2)
    finally:
        din.free()
        dout.free()


def fri_fold(prover, coeffs_re_im, arity_bits, beta):
    """one FRI reduction layer in coefficient space (cp_fri_fold_dev). coeffs_re_im: (2, n_in) uint64 -> (2, n_in >> arity_bits)."""
    c = _as_u64(coeffs_re_im)
    n_in = c.shape[1]
    din, dout = prover.to_device(c), prover.alloc(2 * (n_in >> arity_bits))
    try:
        b = _as_u64(beta)
        prover._check(prover.lib.cp_fri_fold_dev(prover.ctx, din.ptr, n_in, arity_bits, _ptr(b), dout.ptr))
        return dout.download().reshape(2, n_in >> arity_bits)
    finally:
        din.free()
        dout.free()


# ---- the two plonky2-level seams: PolynomialBatch handles and prove_openings / verify_fri_proof -------------------
class ChallengerState(ctypes.Structure):
    """cp_challenger_state: plonky2 `Challenger` by value (sponge state, pending inputs, unread outputs)."""
    _fields_ = [("sponge_state", ctypes.c_uint64 * 12), ("input_buffer", ctypes.c_uint64 * 8),
                ("output_buffer", ctypes.c_uint64 * 8), ("n_input", ctypes.c_uint32), ("n_output", ctypes.c_uint32)]

    def observe(self, elements):
        e = _as_u64(elements).ravel()
        lib = load_library()
        rc = lib.cp_challenger_observe(ctypes.byref(self), _ptr(e) if e.size else None, e.size)
        if rc != 0:
            raise CityProverError(f"[{rc}] " + lib.cp_last_error(None).decode())
        return self

    def challenges(self, count):
        out = np.zeros(count, np.uint64)
        lib = load_library()
        rc = lib.cp_challenger_challenges(ctypes.byref(self), _ptr(out), count)
        if rc != 0:
            raise CityProverError(f"[{rc}] " + lib.cp_last_error(None).decode())
        return out

    def ext_challenge(self):
        return self.challenges(2)

    def copy(self):
        c = ChallengerState()
        ctypes.memmove(ctypes.byref(c), ctypes.byref(self), ctypes.sizeof(ChallengerState))
        return c

    def as_tuple(self):
        return (tuple(self.sponge_state), tuple(self.input_buffer[:self.n_input]), tuple(self.output_buffer[:self.n_output]))


class FriParams(ctypes.Structure):
    """cp_fri_params (plonky2 FriParams)."""
    _fields_ = [(n, ctypes.c_int) for n in ("degree_bits", "rate_bits", "cap_height", "pow_bits", "num_query_rounds", "n_arity")] + [
        ("arity_bits", ctypes.c_int * 8)]


def fri_params(degree_bits, rate_bits, cap_height, pow_bits, num_query_rounds, arity_bits):
    p = FriParams(degree_bits, rate_bits, cap_height, pow_bits, num_query_rounds, len(arity_bits))
    for i, a in enumerate(arity_bits):
        p.arity_bits[i] = a
    return p


class FriPolyRange(ctypes.Structure):
    _fields_ = [("oracle", ctypes.c_uint32), ("first", ctypes.c_uint32), ("count", ctypes.c_uint32)]


class FriBatch(ctypes.Structure):
    """cp_fri_batch (plonky2 FriBatchInfo with the polynomial list as runs)."""
    _fields_ = [("point", ctypes.c_uint64 * 2), ("ranges", ctypes.POINTER(FriPolyRange)), ("n_ranges", ctypes.c_size_t)]


class BatchPoolInfo(ctypes.Structure):
    _fields_ = [("bytes", ctypes.c_size_t), ("buffers", ctypes.c_size_t), ("hits", ctypes.c_size_t), ("misses", ctypes.c_size_t),
                ("trims", ctypes.c_size_t), ("cap_bytes", ctypes.c_size_t), ("pooled", ctypes.c_int)]


class FriOracleInfo(ctypes.Structure):
    _fields_ = [("num_polys", ctypes.c_uint32), ("blinding", ctypes.c_uint32)]


def _fri_batches(batches):
    """[(point (2,), [(oracle, first, count), ...]), ...] -> (ctypes array of FriBatch, keep-alive list)"""
    arr = (FriBatch * len(batches))()
    keep = []
    for i, (pt, ranges) in enumerate(batches):
        rr = (FriPolyRange * max(1, len(ranges)))()
        for j, (o, f, c) in enumerate(ranges):
            rr[j] = FriPolyRange(o, f, c)
        keep.append(rr)
        arr[i].point[0], arr[i].point[1] = int(pt[0]), int(pt[1])
        arr[i].ranges = ctypes.cast(rr, ctypes.POINTER(FriPolyRange))
        arr[i].n_ranges = len(ranges)
    return arr, keep


BATCH_FROM_COEFFS = 1
_u8pp = ctypes.POINTER(ctypes.POINTER(ctypes.c_uint8))
ABI.update({
    "cp_batch_commit": (ctypes.c_int, [_vp, _u64p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_uint, _u64p,
                                       ctypes.POINTER(_vp)]),
    "cp_batch_commit_dev": (ctypes.c_int, [_vp, _vp, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_uint, _vp,
                                           ctypes.POINTER(_vp)]),
    "cp_batch_destroy": (None, [_vp]),
    "cp_batch_pool_stats": (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(BatchPoolInfo)]),
    "cp_batch_pool_trim": (ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_size_t)]),
    "cp_batch_info": (ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_size_t)] + [ctypes.POINTER(ctypes.c_int)] * 4),
    "cp_batch_cap": (ctypes.c_int, [_vp, _u64p]),
    "cp_batch_eval_ext": (ctypes.c_int, [_vp, ctypes.c_size_t, ctypes.c_size_t, _u64p, _u64p]),
    "cp_batch_lde_rows": (ctypes.c_int, [_vp, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_size_t, _u64p]),
    "cp_batch_leaves": (ctypes.c_int, [_vp, ctypes.c_size_t, ctypes.c_size_t, _u64p]),
    "cp_batch_coeffs": (ctypes.c_int, [_vp, ctypes.c_size_t, ctypes.c_size_t, _u64p]),
    "cp_batch_device_ptrs": (ctypes.c_int, [_vp, ctypes.POINTER(_vp), ctypes.POINTER(_vp)]),
    "cp_fri_prove": (ctypes.c_int, [_vp, ctypes.POINTER(_vp), ctypes.c_size_t, ctypes.POINTER(FriBatch), ctypes.c_size_t,
                                    ctypes.POINTER(FriParams), ctypes.POINTER(ChallengerState), ctypes.c_int, ctypes.c_uint64,
                                    _u8pp, ctypes.POINTER(ctypes.c_size_t)]),
    "cp_fri_verify": (ctypes.c_int, [ctypes.POINTER(FriParams), ctypes.POINTER(FriOracleInfo), ctypes.c_size_t,
                                     ctypes.POINTER(_u64p), ctypes.POINTER(FriBatch), ctypes.c_size_t, ctypes.POINTER(_u64p),
                                     ctypes.POINTER(ChallengerState), ctypes.c_char_p, ctypes.c_size_t]),
    "cp_challenger_observe": (ctypes.c_int, [ctypes.POINTER(ChallengerState), _u64p, ctypes.c_size_t]),
    "cp_challenger_challenges": (ctypes.c_int, [ctypes.POINTER(ChallengerState), _u64p, ctypes.c_size_t]),
})


def batch_pool_stats(device=0):
    """the batch-handle buffer pool of `device` (include/cityprover.h cp_batch_pool_stats) as a dict"""
    info = BatchPoolInfo()
    rc = load_library().cp_batch_pool_stats(device, ctypes.byref(info))
    if rc != 0:
        raise CityProverError(rc, load_library().cp_last_error(None).decode())
    return {f: getattr(info, f) for f, _ in BatchPoolInfo._fields_}


def batch_pool_trim(prover):
    released = ctypes.c_size_t(0)
    prover._check(prover.lib.cp_batch_pool_trim(prover.ctx, ctypes.byref(released)))
    return released.value


class PolyBatch:
    """cp_poly_batch: plonky2 `PolynomialBatch` resident on the device (coefficients, bit-reversed LDE, Merkle tree)."""

    def __init__(self, prover, polys, rate_bits, cap_height, from_coeffs=False, salts=None, device_ptr=None, shape=None):
        self.prover = prover
        h = _vp()
        flags = BATCH_FROM_COEFFS if from_coeffs else 0
        if device_ptr is not None:
            k, n = shape
            sp = None if salts is None else salts
            prover._check(prover.lib.cp_batch_commit_dev(prover.ctx, device_ptr, k, int(n).bit_length() - 1, rate_bits, cap_height,
                                                         flags, sp, ctypes.byref(h)))
        else:
            v = _as_u64(polys)
            k, n = v.shape
            s = None if salts is None else _as_u64(salts)
            prover._check(prover.lib.cp_batch_commit(prover.ctx, _ptr(v), k, int(n).bit_length() - 1, rate_bits, cap_height, flags,
                                                     None if s is None else _ptr(s), ctypes.byref(h)))
        self.handle = h.value
        self.k, self.degree_bits, self.rate_bits, self.cap_height = k, int(n).bit_length() - 1, rate_bits, cap_height
        self.blinding = salts is not None

    def cap(self):
        cap = np.zeros((1 << self.cap_height, 4), np.uint64)
        self.prover._check(self.prover.lib.cp_batch_cap(self.handle, _ptr(cap)))
        return cap

    def eval_ext(self, point, first=0, count=None):
        count = self.k - first if count is None else count
        out = np.zeros((count, 2), np.uint64)
        pt = _as_u64(point)
        self.prover._check(self.prover.lib.cp_batch_eval_ext(self.handle, first, count, _ptr(pt), _ptr(out)))
        return out

    def lde_rows(self, first_index, count, step=1):
        out = np.zeros((count, self.k), np.uint64)
        self.prover._check(self.prover.lib.cp_batch_lde_rows(self.handle, first_index, count, step, _ptr(out)))
        return out

    def leaves(self, first_leaf, count):
        out = np.zeros((count, self.k + (SALT_SIZE if self.blinding else 0)), np.uint64)
        self.prover._check(self.prover.lib.cp_batch_leaves(self.handle, first_leaf, count, _ptr(out)))
        return out

    def coeffs(self, first=0, count=None):
        count = self.k - first if count is None else count
        out = np.zeros((count, 1 << self.degree_bits), np.uint64)
        self.prover._check(self.prover.lib.cp_batch_coeffs(self.handle, first, count, _ptr(out)))
        return out

    def device_ptrs(self):
        c, l = _vp(), _vp()
        self.prover._check(self.prover.lib.cp_batch_device_ptrs(self.handle, ctypes.byref(c), ctypes.byref(l)))
        return c.value, l.value

    def close(self):
        if self.handle:
            self.prover.lib.cp_batch_destroy(self.handle)
            self.handle = None


def fri_prove(prover, oracles, batches, params, challenger, pow_override=None):
    """cp_fri_prove: `PolynomialBatch::prove_openings`. oracles: [PolyBatch]; batches: [(point, [(oracle, first, count)])];
    challenger: ChallengerState (advanced in place). Returns bincode FriProof bytes."""
    hs = (_vp * len(oracles))(*[o.handle for o in oracles])
    arr, keep = _fri_batches(batches)
    out = ctypes.POINTER(ctypes.c_uint8)()
    ln = ctypes.c_size_t()
    prover._check(prover.lib.cp_fri_prove(prover.ctx, hs, len(oracles), arr, len(batches), ctypes.byref(params), ctypes.byref(challenger),
                                          0 if pow_override is None else 1, pow_override or 0, ctypes.byref(out), ctypes.byref(ln)))
    try:
        return ctypes.string_at(out, ln.value)
    finally:
        prover.lib.cp_free(out)


def fri_verify(params, oracle_infos, caps, batches, opened_values, challenger, proof_bytes):
    """cp_fri_verify: `fri_challenges` + `verify_fri_proof`. oracle_infos: [(num_polys, blinding)]; caps: [array (2^cap_height, 4)];
    opened_values: per batch an (n_polys, 2) array. Raises CityProverError on rejection; advances `challenger` on success."""
    lib = load_library()
    infos = (FriOracleInfo * len(oracle_infos))(*[FriOracleInfo(int(k), int(bool(b))) for k, b in oracle_infos])
    cap_arrs = [_as_u64(c) for c in caps]
    capp = (_u64p * len(cap_arrs))(*[_ptr(c) for c in cap_arrs])
    arr, keep = _fri_batches(batches)
    ov = [_as_u64(o) for o in opened_values]
    ovp = (_u64p * len(ov))(*[_ptr(o) if o.size else None for o in ov])
    rc = lib.cp_fri_verify(ctypes.byref(params), infos, len(oracle_infos), capp, arr, len(batches), ovp, ctypes.byref(challenger),
                           proof_bytes, len(proof_bytes))
    if rc != 0:
        raise CityProverError(f"[{rc}] " + lib.cp_last_error(None).decode())


# ---- generic AIR machinery (include/cityprover.h "the STARK's own two steps as GENERIC device machinery") -----------------
(AIR_LOCAL, AIR_NEXT, AIR_PUBLIC, AIR_GLOBAL, AIR_CHALLENGE, AIR_CONST, AIR_ADD, AIR_SUB, AIR_MUL, AIR_NEG, AIR_INV, AIR_ASSERT_ZERO,
 AIR_ASSERT_ZERO_TRANSITION, AIR_ASSERT_ZERO_FIRST_ROW, AIR_ASSERT_ZERO_LAST_ROW, AIR_STORE) = range(16)
AIR_CONSTRAINTS, AIR_MAP = 0, 1
STARK_STEP_MAP, STARK_STEP_CUBIC_INVERSE, STARK_STEP_PREFIX_SUM = 0, 1, 2


class AirProgramDesc(ctypes.Structure):
    _fields_ = [("kind", ctypes.c_int), ("ops", _vp), ("n_ops", ctypes.c_size_t), ("consts", _u64p), ("n_consts", ctypes.c_size_t),
                ("n_columns", ctypes.c_uint32), ("n_public", ctypes.c_uint32), ("n_global", ctypes.c_uint32), ("n_challenge", ctypes.c_uint32),
                ("n_out_columns", ctypes.c_uint32)]


class AirProgramInfo(ctypes.Structure):
    _fields_ = [("n_ops", ctypes.c_size_t), ("n_live_ops", ctypes.c_size_t), ("n_constraints", ctypes.c_size_t),
                ("max_constraint_degree", ctypes.c_uint32), ("n_segments_max", ctypes.c_uint32), ("n_slots", ctypes.c_uint32),
                ("n_instructions", ctypes.c_size_t)]


class StarkStep(ctypes.Structure):
    _fields_ = [("kind", ctypes.c_int), ("program", _vp), ("first", ctypes.c_uint32), ("count", ctypes.c_uint32), ("flags", ctypes.c_uint32),
                ("modulus", ctypes.c_uint64 * 2)]


class StarkDesc(ctypes.Structure):
    _fields_ = [("degree_bits", ctypes.c_int), ("quotient_degree_bits", ctypes.c_int), ("num_challenges", ctypes.c_uint32), ("fri", FriParams),
                ("n_trace_columns", ctypes.c_uint32), ("n_extended_columns", ctypes.c_uint32), ("n_round_challenges", ctypes.c_uint32),
                ("n_public", ctypes.c_uint32), ("n_global", ctypes.c_uint32), ("steps", ctypes.POINTER(StarkStep)), ("n_steps", ctypes.c_size_t),
                ("constraints", _vp)]


ABI.update({
    "cp_air_program_create": (_vp, [_vp, ctypes.POINTER(AirProgramDesc)]),
    "cp_air_program_destroy": (None, [_vp]),
    "cp_air_program_get_info": (ctypes.c_int, [_vp, ctypes.POINTER(AirProgramInfo)]),
    "cp_air_program_eval_ext": (ctypes.c_int, [_vp, _u64p, _u64p, _u64p, _u64p, _u64p, _u64p, ctypes.POINTER(ctypes.c_uint32)]),
    "cp_air_quotient_commit": (ctypes.c_int, [_vp, _vp, ctypes.POINTER(_vp), ctypes.c_size_t, ctypes.c_int, _u64p, _u64p, _u64p, _u64p,
                                              ctypes.c_size_t, ctypes.POINTER(_vp)]),
    "cp_air_map_dev": (ctypes.c_int, [_vp, _vp, _vp, _vp, ctypes.c_size_t, _u64p, _u64p, _u64p]),
    "cp_cubic_batch_inverse_dev": (ctypes.c_int, [_vp, _u64p, _vp, ctypes.c_size_t, ctypes.c_size_t]),
    "cp_column_prefix_sum_dev": (ctypes.c_int, [_vp, _vp, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_int]),
    "cp_stark_prove": (ctypes.c_int, [_vp, ctypes.POINTER(StarkDesc), _vp, ctypes.c_int, _u64p, _u64p, ctypes.POINTER(ChallengerState), ctypes.c_int,
                                      ctypes.c_uint64, _u8pp, ctypes.POINTER(ctypes.c_size_t)]),
    "cp_stark_verify": (ctypes.c_int, [ctypes.POINTER(StarkDesc), _u64p, _u64p, ctypes.POINTER(ChallengerState), ctypes.c_char_p, ctypes.c_size_t]),
})


def _opt_u64(x):
    """None / empty -> (NULL, keep-alive None); else a contiguous u64 array and its pointer"""
    if x is None:
        return None, None
    a = _as_u64(x)
    return (_ptr(a) if a.size else None), a


class AirProgram:
    """cp_air_program: a straight-line program over F_p. ops: (n, 4) uint32 rows (op, a, b, 0) with the AIR_* codes."""

    def __init__(self, prover, kind, ops, consts=(), n_columns=0, n_public=0, n_global=0, n_challenge=0, n_out_columns=0):
        self.prover = prover
        o = np.ascontiguousarray(np.asarray(ops, dtype=np.uint32).reshape(-1, 4))
        c = _as_u64(np.asarray(list(consts) if not isinstance(consts, np.ndarray) else consts, dtype=np.uint64))
        d = AirProgramDesc(kind, o.ctypes.data if o.size else None, len(o), _ptr(c) if c.size else None, c.size, n_columns, n_public, n_global,
                           n_challenge, n_out_columns)
        self.handle = prover.lib.cp_air_program_create(prover.ctx, ctypes.byref(d))
        if not self.handle:
            raise CityProverError(prover.lib.cp_last_error(prover.ctx).decode())
        self.kind, self.n_columns, self.n_public, self.n_global, self.n_challenge, self.n_out_columns = kind, n_columns, n_public, n_global, n_challenge, n_out_columns

    def info(self):
        i = AirProgramInfo()
        self.prover._check(self.prover.lib.cp_air_program_get_info(self.handle, ctypes.byref(i)))
        return {f: getattr(i, f) for f, _ in AirProgramInfo._fields_}

    def eval_ext(self, local, nxt, publics=None, globals_=None, challenges=None):
        """constraint values on one row over F_p^2: ((n_constraints, 2) array, kinds)"""
        n = self.info()["n_constraints"]
        out = np.zeros((n, 2), np.uint64)
        kinds = np.zeros(max(n, 1), np.uint32)
        ptrs, keep = zip(*[_opt_u64(x) for x in (local, nxt, publics, globals_, challenges)])
        self.prover._check(self.prover.lib.cp_air_program_eval_ext(self.handle, *ptrs, _ptr(out), kinds.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32))))
        return out, kinds[:n]

    def close(self):
        if self.handle:
            self.prover.lib.cp_air_program_destroy(self.handle)
            self.handle = None


def _wrap_batch(prover, handle):
    b = PolyBatch.__new__(PolyBatch)
    b.prover, b.handle = prover, handle
    k, db, rb, ch, ns = ctypes.c_size_t(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    prover._check(prover.lib.cp_batch_info(handle, ctypes.byref(k), ctypes.byref(db), ctypes.byref(rb), ctypes.byref(ch), ctypes.byref(ns)))
    b.k, b.degree_bits, b.rate_bits, b.cap_height, b.blinding = k.value, db.value, rb.value, ch.value, ns.value != 0
    return b


def air_quotient_commit(prover, program, oracles, quotient_degree_bits, alphas, publics=None, globals_=None, challenges=None):
    """cp_air_quotient_commit -> PolyBatch of len(alphas) * 2^q quotient chunk polynomials"""
    hs = (_vp * len(oracles))(*[o.handle for o in oracles])
    al = _as_u64(alphas)
    (pp, gp, cp_), keep = zip(*[_opt_u64(x) for x in (publics, globals_, challenges)])
    out = _vp()
    prover._check(prover.lib.cp_air_quotient_commit(prover.ctx, program.handle, hs, len(oracles), quotient_degree_bits, pp, gp, cp_, _ptr(al), al.size,
                                                    ctypes.byref(out)))
    return _wrap_batch(prover, out.value)


def air_map(prover, program, in_cols, publics=None, globals_=None, challenges=None):
    """cp_air_map_dev on host arrays: in_cols (n_columns, n) -> (n_out_columns, n) (columns never stored stay 0)"""
    v = _as_u64(in_cols)
    n = v.shape[1]
    din = prover.to_device(v)
    dout = prover.alloc(max(program.n_out_columns, 1) * n)
    try:
        dout.upload(np.zeros(max(program.n_out_columns, 1) * n, np.uint64))
        (pp, gp, cp_), keep = zip(*[_opt_u64(x) for x in (publics, globals_, challenges)])
        prover._check(prover.lib.cp_air_map_dev(prover.ctx, program.handle, din.ptr, dout.ptr, n, pp, gp, cp_))
        return dout.download(program.n_out_columns * n).reshape(program.n_out_columns, n)
    finally:
        din.free()
        dout.free()


def cubic_batch_inverse(prover, modulus, cols):
    """cp_cubic_batch_inverse_dev on a host array: cols (3 * count, n) -> the inverses, same layout"""
    v = _as_u64(cols)
    d = prover.to_device(v)
    try:
        m = _as_u64(modulus)
        prover._check(prover.lib.cp_cubic_batch_inverse_dev(prover.ctx, _ptr(m), d.ptr, v.shape[0] // 3, v.shape[1]))
        return d.download(v.size).reshape(v.shape)
    finally:
        d.free()


def column_prefix_sum(prover, cols, exclusive=False):
    v = _as_u64(cols)
    d = prover.to_device(v)
    try:
        prover._check(prover.lib.cp_column_prefix_sum_dev(prover.ctx, d.ptr, v.shape[0], v.shape[1], int(exclusive)))
        return d.download(v.size).reshape(v.shape)
    finally:
        d.free()


def stark_desc(degree_bits, quotient_degree_bits, num_challenges, fri, n_trace_columns, constraints, n_extended_columns=0, n_round_challenges=0,
               n_public=0, n_global=0, steps=()):
    """cp_stark_desc. steps: [("map", AirProgram) | ("cubic_inverse", first, count, (m0, m1)) | ("prefix_sum", first, count, exclusive)].
    Returns (desc, keep-alive) — hold on to both."""
    arr = (StarkStep * max(1, len(steps)))()
    for i, st in enumerate(steps):
        if st[0] == "map":
            arr[i].kind, arr[i].program = STARK_STEP_MAP, st[1].handle
        elif st[0] == "cubic_inverse":
            arr[i].kind, arr[i].first, arr[i].count = STARK_STEP_CUBIC_INVERSE, st[1], st[2]
            arr[i].modulus[0], arr[i].modulus[1] = int(st[3][0]), int(st[3][1])
        elif st[0] == "prefix_sum":
            arr[i].kind, arr[i].first, arr[i].count, arr[i].flags = STARK_STEP_PREFIX_SUM, st[1], st[2], int(bool(st[3]))
        else:
            raise ValueError(st[0])
    d = StarkDesc(degree_bits, quotient_degree_bits, num_challenges, fri, n_trace_columns, n_extended_columns, n_round_challenges, n_public, n_global,
                  ctypes.cast(arr, ctypes.POINTER(StarkStep)), len(steps), constraints.handle)
    return d, (arr, steps, constraints)


def stark_prove(prover, desc, trace, challenger, publics=None, globals_=None, pow_override=None):
    """cp_stark_prove: trace (n_trace_columns, n) host array -> proof bytes; advances `challenger`"""
    t = _as_u64(trace)
    (pp, gp), keep = zip(*[_opt_u64(x) for x in (publics, globals_)])
    out, ln = ctypes.POINTER(ctypes.c_uint8)(), ctypes.c_size_t(0)
    prover._check(prover.lib.cp_stark_prove(prover.ctx, ctypes.byref(desc), t.ctypes.data, 0, pp, gp, ctypes.byref(challenger),
                                            0 if pow_override is None else 1, 0 if pow_override is None else int(pow_override),
                                            ctypes.byref(out), ctypes.byref(ln)))
    try:
        return ctypes.string_at(out, ln.value)
    finally:
        prover.lib.cp_free(out)


def stark_verify(desc, challenger, proof, publics=None, globals_=None):
    lib = load_library()
    (pp, gp), keep = zip(*[_opt_u64(x) for x in (publics, globals_)])
    rc = lib.cp_stark_verify(ctypes.byref(desc), pp, gp, ctypes.byref(challenger), proof, len(proof))
    if rc != 0:
        raise CityProverError(f"[{rc}] " + lib.cp_last_error(None).decode())
