"""P7: the reference ProofWithPublicInputs blobs from qbench_data/example.bin pin (i) the bincode
layout byte for byte, (ii) the proof shape, (iii) Merkle-path KATs on real 85/135/20/16-felt
leaves against the in-proof caps (leaf hashing, two_to_one, cap indexing)."""
import json
import os

import numpy as np
import pytest

import oracle_lib as O
from proof_format import find_leaf_index, parse_proof, serialize_proof


def proofs(golden_dir):
    from proof_format import reference_proofs
    got = reference_proofs(golden_dir)
    assert len(got) == 10   # 8 WrappedSignatureProof + 2 Secp256K1SignatureProof: every proof the dump holds
    return got


def test_bincode_layout_roundtrip(golden_dir):
    for m, blob in proofs(golden_dir):
        p = parse_proof(blob)
        assert serialize_proof(p) == blob
        assert len(blob) == m["len"]


def test_proof_shape(golden_dir):
    for m, blob in proofs(golden_dir):
        p = parse_proof(blob)
        assert [len(p[c]) for c in ("wires_cap", "zs_pp_cap", "quotient_cap")] == [16, 16, 16]
        o = p["openings"]
        assert [len(o[k]) for k in ("constants", "plonk_sigmas", "wires", "plonk_zs", "plonk_zs_next",
                                    "partial_products", "quotient_polys", "lookup_zs",
                                    "lookup_zs_next")] == [5, 80, 135, 2, 2, 18, 16, 0, 0]
        assert [len(c) for c in p["commit_caps"]] == [16, 16]
        assert len(p["queries"]) == 28 and len(p["final_poly"]) == 16
        for q in p["queries"]:
            assert [len(e[0]) for e in q["initial"]] == [85, 135, 20, 16]
            assert all(len(e[1]) == 11 for e in q["initial"])
            assert [(len(e[0]), len(e[1])) for e in q["steps"]] == [(16, 7), (16, 3)]
        assert len(p["public_inputs"]) == (8 if m["circuit_type"] == 64 else 4)
        for v in p["public_inputs"]:
            assert v < O.P


def test_merkle_paths_against_in_proof_caps(golden_dir):
    m, blob = proofs(golden_dir)[0]
    p = parse_proof(blob)
    caps = {1: p["wires_cap"], 2: p["zs_pp_cap"], 3: p["quotient_cap"]}
    for q in p["queries"][:6]:
        leaf, sib = q["initial"][1]
        idx = find_leaf_index(leaf, sib, caps[1], O)
        assert idx is not None, "wires path does not reach the wires cap"
        for t in (2, 3):  # same x_index opens every oracle
            leaf, sib = q["initial"][t]
            assert O.merkle_verify(np.array(leaf, np.uint64), idx, np.array(sib, np.uint64),
                                   np.array(caps[t], np.uint64), 4)
        # FRI layer trees: leaf = 16 ext = 32 felts, index = x_index >> 4, then >> 8
        ev, sib = q["steps"][0]
        flat = np.array([c for e in ev for c in e], np.uint64)
        assert O.merkle_verify(flat, idx >> 4, np.array(sib, np.uint64),
                               np.array(p["commit_caps"][0], np.uint64), 4)
        ev, sib = q["steps"][1]
        flat = np.array([c for e in ev for c in e], np.uint64)
        assert O.merkle_verify(flat, idx >> 8, np.array(sib, np.uint64),
                               np.array(p["commit_caps"][1], np.uint64), 4)
