"""Writers / readers of the two flat file formats of the import bridge, host side, no GPU:

  .cpcirc  one built circuit (layout: csrc/circuit_file.inc; read by cp_circuit_load_file / cp_circuit_file_info)
  .cpwit   one witness of such a circuit: the wire matrix `CircuitData::prove` commits to after witness generation,
           the public inputs and, optionally, the proof bytes the CPU prover produced for it (to check byte parity
           under nonce injection). Read by tools/cityprover_qbench.

The Rust side (rust/plonky2-hwa-patch) writes the same bytes; these Python writers serve the synthetic circuit packs
(tools/make_circuit_pack.py) and the tests."""
import struct

import numpy as np

FLAG_COEFFS, FLAG_K_IS, FLAG_PI_TARGETS = 1, 2, 4
P = 0xFFFFFFFF00000001


def fnv1a64(data):
    """FNV-1a 64 (sequential by construction; ~0.1 s per MB here, files are a few MB)"""
    h = 0xcbf29ce484222325
    mask = (1 << 64) - 1
    for b in data:
        h = ((h ^ b) * 0x100000001b3) & mask
    return h


def shape_ints(sh):
    ab = [int(sh.arity_bits[i]) for i in range(8)]
    return [sh.degree_bits, sh.num_constants, sh.num_routed_wires, sh.num_wires, sh.num_challenges, sh.num_partial_products,
            sh.quotient_degree_factor, sh.rate_bits, sh.cap_height, sh.pow_bits, sh.num_query_rounds, sh.n_arity] + ab + \
           [sh.zero_knowledge, sh.num_public_inputs, 0, 0]


def write_circuit_file(path, shape, digest, gate_list, num_selectors, polys, k_is=None, pi_targets=None, coeffs=False):
    """shape: cityprover.Shape (or anything with the cp_shape fields); gate_list: [(type, selector_index, group_start,
    group_end, param[, param2[, param3]])]; polys: (num_constants + num_routed_wires, n) uint64."""
    polys = np.ascontiguousarray(np.asarray(polys, dtype=np.uint64))
    n = 1 << shape.degree_bits
    assert polys.shape == (shape.num_constants + shape.num_routed_wires, n), polys.shape
    flags = (FLAG_COEFFS if coeffs else 0) | (FLAG_K_IS if k_is is not None else 0) | (FLAG_PI_TARGETS if pi_targets is not None else 0)
    body = bytearray()
    body += struct.pack("<24i", *shape_ints(shape))
    body += struct.pack("<4Q", *[int(x) for x in digest])
    body += struct.pack("<II", num_selectors, len(gate_list))
    for g in gate_list:
        g = tuple(g) + (0,) * (7 - len(g))
        body += struct.pack("<7i", *g)
    if len(gate_list) % 2:
        body += b"\0" * 4
    if k_is is not None:
        assert len(k_is) == shape.num_routed_wires
        body += struct.pack("<%dQ" % len(k_is), *[int(x) for x in k_is])
    if pi_targets is not None:
        t = np.asarray(pi_targets, dtype=np.uint32).reshape(-1, 2)
        assert t.shape[0] == shape.num_public_inputs
        body += t.tobytes()
    body += polys.tobytes()
    total = 24 + len(body) + 8
    out = bytearray(b"CPCIRCv1") + struct.pack("<IIQ", 1, flags, total) + body
    out += struct.pack("<Q", fnv1a64(out))
    with open(path, "wb") as f:
        f.write(out)
    return total


WIT_MAGIC = b"CPWITNv1"


def write_witness_file(path, digest, wires, public_inputs, proof=None):
    """.cpwit: magic | u32 version = 1 | u32 flags (bit 0: proof bytes present) | u64 circuit_digest[4] |
    u32 num_wires, u32 degree_bits, u32 n_public_inputs, u32 0 | public inputs (u64 each) |
    wires [num_wires][n] u64 | (u64 proof_len, proof bytes, zero padding to 8) | u64 FNV-1a of everything before."""
    w = np.ascontiguousarray(np.asarray(wires, dtype=np.uint64))
    num_wires, n = w.shape
    db = int(n).bit_length() - 1
    assert 1 << db == n
    pi = np.asarray(public_inputs, dtype=np.uint64)
    out = bytearray(WIT_MAGIC) + struct.pack("<II", 1, 1 if proof is not None else 0)
    out += struct.pack("<4Q", *[int(x) for x in digest])
    out += struct.pack("<IIII", num_wires, db, len(pi), 0)
    out += pi.tobytes() + w.tobytes()
    if proof is not None:
        out += struct.pack("<Q", len(proof)) + bytes(proof) + b"\0" * (-len(proof) % 8)
    out += struct.pack("<Q", fnv1a64(out))
    with open(path, "wb") as f:
        f.write(out)


def read_witness_file(path):
    b = open(path, "rb").read()
    assert b[:8] == WIT_MAGIC, "not a witness file"
    version, flags = struct.unpack_from("<II", b, 8)
    assert version == 1
    assert struct.unpack_from("<Q", b, len(b) - 8)[0] == fnv1a64(b[:-8]), "checksum mismatch"
    digest = list(struct.unpack_from("<4Q", b, 16))
    num_wires, db, n_pi, _ = struct.unpack_from("<IIII", b, 48)
    o = 64
    pi = np.frombuffer(b, np.uint64, n_pi, o).copy()
    o += 8 * n_pi
    wires = np.frombuffer(b, np.uint64, num_wires << db, o).reshape(num_wires, 1 << db).copy()
    o += 8 * (num_wires << db)
    proof = None
    if flags & 1:
        ln = struct.unpack_from("<Q", b, o)[0]
        proof = b[o + 8:o + 8 + ln]
    return dict(digest=digest, public_inputs=pi, wires=wires, proof=proof)
