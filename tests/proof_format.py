"""bincode layout of plonky2 `ProofWithPublicInputs<GoldilocksField, PoseidonGoldilocksConfig, 2>`
as recovered from the reference proofs in qbench_data/example.bin.
Test-side parser; the product-side (de)serialiser lives in the C++ host."""
import struct


class _R:
    def __init__(self, b):
        self.b, self.o = b, 0

    def u64(self):
        v = struct.unpack_from("<Q", self.b, self.o)[0]
        self.o += 8
        return v

    def felts(self, n):
        v = list(struct.unpack_from("<%dQ" % n, self.b, self.o))
        self.o += 8 * n
        return v

    def hash(self):
        return self.felts(4)

    def vec(self, item):
        return [item() for _ in range(self.u64())]

    def ext(self):
        return self.felts(2)


def parse_proof(b):
    r = _R(b)
    p = {}
    p["wires_cap"] = r.vec(r.hash)
    p["zs_pp_cap"] = r.vec(r.hash)
    p["quotient_cap"] = r.vec(r.hash)
    names = ["constants", "plonk_sigmas", "wires", "plonk_zs", "plonk_zs_next", "partial_products",
             "quotient_polys", "lookup_zs", "lookup_zs_next"]
    p["openings"] = {n: r.vec(r.ext) for n in names}
    p["commit_caps"] = r.vec(lambda: r.vec(r.hash))
    qs = []
    for _ in range(r.u64()):
        initial = r.vec(lambda: (r.vec(r.u64), r.vec(r.hash)))
        steps = r.vec(lambda: (r.vec(r.ext), r.vec(r.hash)))
        qs.append({"initial": initial, "steps": steps})
    p["queries"] = qs
    p["final_poly"] = r.vec(r.ext)
    p["pow_witness"] = r.u64()
    p["public_inputs"] = r.vec(r.u64)
    assert r.o == len(b), (r.o, len(b))
    return p


def serialize_proof(p):
    out = bytearray()

    def u64(v):
        out.extend(struct.pack("<Q", v))

    def felts(v):
        out.extend(struct.pack("<%dQ" % len(v), *v))

    def vec(items, f):
        u64(len(items))
        for it in items:
            f(it)

    vec(p["wires_cap"], felts)
    vec(p["zs_pp_cap"], felts)
    vec(p["quotient_cap"], felts)
    for n in ["constants", "plonk_sigmas", "wires", "plonk_zs", "plonk_zs_next", "partial_products",
              "quotient_polys", "lookup_zs", "lookup_zs_next"]:
        vec(p["openings"][n], felts)
    vec(p["commit_caps"], lambda c: vec(c, felts))
    u64(len(p["queries"]))
    for q in p["queries"]:
        vec(q["initial"], lambda e: (vec(e[0], u64), vec(e[1], felts)))
        vec(q["steps"], lambda e: (vec(e[0], felts), vec(e[1], felts)))
    vec(p["final_poly"], felts)
    u64(p["pow_witness"])
    vec(p["public_inputs"], u64)
    return bytes(out)


def find_leaf_index(leaf, siblings, cap, O):
    """Recover the leaf index of a Merkle path whose cap is known: try both orders per level
    (2^depth candidates explored as a DFS pruned only at the cap)."""
    import numpy as np
    cap_set = {tuple(c): i for i, c in enumerate(cap)}
    depth = len(siblings)
    start = O.hash_or_noop(np.array(leaf, np.uint64))

    def rec(level, cur, idx):
        if level == depth:
            t = tuple(int(x) for x in cur)
            if t in cap_set:
                return idx | (cap_set[t] << depth)
            return None
        s = np.array(siblings[level], np.uint64)
        for bit in (0, 1):
            nxt = O.two_to_one(cur, s) if bit == 0 else O.two_to_one(s, cur)
            r = rec(level + 1, nxt, idx | (bit << level))
            if r is not None:
                return r
        return None

    return rec(0, start, 0)


def reference_proofs(golden_dir):
    """[(meta, bytes)] for the ten reference `ProofWithPublicInputs` of qbench_data/example.bin — slices of the dump the
    reference's own q-bench harness reads, kept whole as tests/golden/qbench_example.bin (tests/golden/make_golden.py)."""
    import json
    import os
    meta = json.load(open(os.path.join(golden_dir, "example_proofs.json")))
    blob = open(os.path.join(golden_dir, meta[0]["file"]), "rb").read()
    return [(m, blob[m["offset"]:m["offset"] + m["len"]]) for m in meta]
