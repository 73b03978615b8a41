#!/usr/bin/env python3
"""Extract known-answer DATA from the reference tree into tests/golden/.

Run once in the build container (where /root/reference exists):

    python3 tests/golden/make_golden.py

Everything written is data (inputs / expected outputs) that the reference's own
tests and fixtures hold for the proving hot path; no reference source text is
kept. Sources (paths relative to /root/reference):

  P1/P2  city_crypto/src/hash/cached_zero_hashes.rs:11-2065
         128 iterated two_to_one zero hashes + 128 iterated marked-leaf hashes
  P3     city_rollup_core_orchestrator/src/lib.rs:52
         circuit fingerprints: root = two_to_one(leaf_fp, aggregator_fp)
         (rule: city_crypto/src/hash/merkle/treeprover/mod.rs:358-359)
  P4     city_common_circuit/src/hash/merkle/gadgets/merkle_proof.rs:243-1121
         city_common_circuit/src/hash/merkle/gadgets/delta_merkle_proof.rs:573-877
  P6/P7  qbench_data/example.bin  (bincode BlockProofStoreDump,
         city_rollup_core_worker_qbench/src/dump.rs:15-26) - the data file of the reference's own q-bench
         harness, kept WHOLE as tests/golden/qbench_example.bin (1.4 MB of data: 10 reference proofs, 46 job
         witnesses, the job DAG): it is the input tools/cityprover_qbench reads natively
  P5     city_rollup_common/src/config/sighash_wrapper_config.rs:16-1900 (+ the leaf order of
         city_store/src/store/sighash/mod.rs:50-73): the whitelist tree of 1875 circuit fingerprints and its root
  G16    city_rollup_common/src/block_template/data.rs:72-73
  G16vk  city_rollup_common/src/block_template/verifier_data.rs:1-16
         the two CityGroth16ProofData samples of `test_serde` (4 x 48-byte compressed BLS12-381 elements each)
"""
import json
import os
import re
import struct
import sys

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def hex_to_felts(h):
    """QHashOut hex = reversed little-endian bytes of the 4 u64
    (city_crypto/src/hash/qhashout.rs:28-58)."""
    b = bytes.fromhex(h)[::-1]
    b = b + b"\0" * (32 - len(b))
    return list(struct.unpack("<4Q", b))


def zero_hashes():
    src = open(f"{REF}/city_crypto/src/hash/cached_zero_hashes.rs").read()
    vals = [int(x) for x in re.findall(r"GoldilocksField\((\d+)\)", src)]
    # four tables of 128 hashes x 4 felts: HashOut plain, HashOut marked, QHashOut plain, QHashOut marked
    assert len(vals) == 4 * 128 * 4, len(vals)
    t = [[vals[(k * 128 + i) * 4:(k * 128 + i) * 4 + 4] for i in range(128)] for k in range(4)]
    assert t[0] == t[2] and t[1] == t[3]
    return {"two_to_one": t[0], "marked_leaf": t[1]}


def fingerprints():
    src = open(f"{REF}/city_rollup_core_orchestrator/src/lib.rs").read()
    lits = re.findall(r'r#"\s*(\{"network_magic".*?\})\s*"#', src, re.S)
    out = []
    seen = set()
    for lit in lits:
        try:
            d = json.loads(lit)
        except json.JSONDecodeError:
            continue
        for k, v in d.items():
            if isinstance(v, dict) and "leaf_fingerprint" in v:
                key = (v["leaf_fingerprint"], v["aggregator_fingerprint"])
                if key in seen:
                    continue
                seen.add(key)
                out.append({
                    "name": k,
                    "leaf": hex_to_felts(v["leaf_fingerprint"]),
                    "aggregator": hex_to_felts(v["aggregator_fingerprint"]),
                    "root": hex_to_felts(v["allowed_circuit_hashes_root"]),
                })
    return out


def json_cases(path):
    src = open(f"{REF}/{path}").read()
    m = re.search(r'TEST_CASES_JSON: &str = r#"(.*?)"#', src, re.S)
    cases = json.loads(m.group(1))
    out = []
    for c in cases:
        o = {"index": c["index"], "siblings": [hex_to_felts(s) for s in c["siblings"]]}
        for k in ("root", "value", "old_root", "old_value", "new_root", "new_value"):
            if k in c:
                o[k] = hex_to_felts(c[k])
        out.append(o)
    return out


class Rd:
    def __init__(self, b, o=0):
        self.b, self.o = b, o

    def u8(self):
        v = self.b[self.o]
        self.o += 1
        return v

    def u32(self):
        v = struct.unpack_from("<I", self.b, self.o)[0]
        self.o += 4
        return v

    def u64(self):
        v = struct.unpack_from("<Q", self.b, self.o)[0]
        self.o += 8
        return v

    def raw(self, n):
        v = self.b[self.o:self.o + n]
        self.o += n
        return v

    def qhash(self):
        n = self.u64()
        assert n == 64, n
        return hex_to_felts(self.raw(64).decode())

    def delta(self):
        d = {"old_root": self.qhash(), "old_value": self.qhash(), "new_root": self.qhash(),
             "new_value": self.qhash(), "index": self.u64()}
        d["siblings"] = [self.qhash() for _ in range(self.u64())]
        return d


def parse_key(k):
    topic, goal, ctype, group, sub, task, dtype, didx = struct.unpack("<BQBIIIBB", k)
    return dict(topic=topic, goal_id=goal, circuit_type=ctype, group_id=group, sub_group_id=sub,
                task_index=task, data_type=dtype, data_index=didx)


def example_bin():
    b = open(f"{REF}/qbench_data/example.bin", "rb").read()
    r = Rd(b)
    cfg = {"checkpoint_id": r.u64(), "rpc_node_id": r.u32(),
           "job_config": [r.u64() for _ in range(6)]}
    n = r.u64()
    entries = []
    for _ in range(n):
        key = r.raw(24)
        ln = r.u64()
        entries.append((key, r.raw(ln)))
    n_counters = r.u64()
    assert n_counters == 0 and r.o == len(b), (n_counters, r.o, len(b))

    index = []
    deltas = []
    proofs = []
    for key, val in entries:
        k = parse_key(key)
        index.append({**k, "key": key.hex(), "len": len(val)})
        # op-leaf witnesses: one or two DeltaMerkleProofCore (job_witnesses/op.rs:58-253)
        if k["data_type"] == 0 and k["topic"] == 0 and k["circuit_type"] in (0, 2, 10):
            rr = Rd(val)
            deltas.append({"circuit_type": k["circuit_type"], "task": k["task_index"],
                           "proofs": [rr.delta()], "allowed_circuit_hashes_root": rr.qhash()})
            assert rr.o == len(val)
        elif k["data_type"] == 0 and k["topic"] == 0 and k["circuit_type"] in (6, 8):
            rr = Rd(val)
            deltas.append({"circuit_type": k["circuit_type"], "task": k["task_index"],
                           "proofs": [rr.delta(), rr.delta()],
                           "allowed_circuit_hashes_root": rr.qhash()})
            assert rr.o + 24 == len(val)
        elif k["data_type"] == 0 and k["topic"] == 0 and k["circuit_type"] == 4:
            # CRClaimL1DepositCircuitInput (job_witnesses/op.rs:145-150): the deposit record, then two delta proofs,
            # the root and the 24-byte signature proof id; the record's size is what the fixed-size tail leaves
            delta_len = 4 * 72 + 8 + 8 + 32 * 72
            rr = Rd(val, len(val) - (2 * delta_len + 72 + 24))
            deltas.append({"circuit_type": 4, "task": k["task_index"], "proofs": [rr.delta(), rr.delta()],
                           "allowed_circuit_hashes_root": rr.qhash()})
            assert rr.o + 24 == len(val)
        elif k["data_type"] == 1 and len(val) > 0:
            proofs.append((k, val, b.index(val)))
    return cfg, index, deltas, proofs


def whitelist_tree():
    """P5: the sighash circuit whitelist — 1875 circuit fingerprints (sighash_wrapper_config.rs:24-1900) and the root of
    the height-16 Merkle tree over them (:16-23). Leaf order: the gadget ids of generate_id_permutations
    (introspection.rs:402-431), sorted by the derived Ord of SigHashGadgetId (field order at :157-163); leaf i of the
    tree holds the fingerprint of the i-th SORTED id (city_store/src/store/sighash/mod.rs:50-73)."""
    src = open(f"{REF}/city_rollup_common/src/config/sighash_wrapper_config.rs").read()
    live, commented = src.split("/*", 1)      # the file also keeps the tree of an earlier configuration (max 2 / 2) in a comment
    out = []
    for text, m in ((live, 4), (commented, 2)):
        vals = [int(x) for x in re.findall(r"GoldilocksField\((\d+)\)", text)]
        count = (m + 1) ** 3 * (m + 1) * (m + 2) // 2
        assert len(vals) == 4 + 4 * count, (len(vals), count)
        root, fps = vals[:4], [vals[4 + 4 * i:8 + 4 * i] for i in range(count)]
        ids = []
        for lw in range(m + 1):
            for ld in range(m + 1):
                for w in range(m + 1):
                    for d in range(m + 1):
                        for s in range(d + 1):
                            ids.append((d, w, ld, lw, s))   # (num_deposits, num_withdrawals, last deposits, last withdrawals, spend index)
        order = sorted(range(count), key=lambda i: ids[i])
        out.append({"max_deposits": m, "max_withdrawals": m, "height": 16, "root": root, "leaves": [fps[i] for i in order]})
    return out


def groth16_samples():
    """The two `CityGroth16ProofData` JSON samples of data.rs:72-73 (pi_a, pi_b_a0, pi_b_a1, pi_c as hex)."""
    src = open(f"{REF}/city_rollup_common/src/block_template/data.rs").read()
    out = []
    for lit in re.findall(r'r#"(\{"pi_a".*?\})"#', src):
        d = json.loads(lit)
        out.append({k: d[k] for k in ("pi_a", "pi_b_a0", "pi_b_a1", "pi_c")})
    assert len(out) == 2
    return out


def groth16_verifier_data():
    """`BLOCK_GROTH16_ENCODED_VERIFIER_DATA` (block_template/verifier_data.rs:1-12): the on-chain Groth16 verifying key as six
    80-byte script pushes, and the SHA-256 the script checks chunk 0 against (:14-16). Data only."""
    src = open(f"{REF}/city_rollup_common/src/block_template/verifier_data.rs").read()
    chunks = re.findall(r'hex!\("([0-9a-f]{160})"\)', src)
    sha = re.search(r'_0_SHA_256_HASH: \[u8; 32\] =\s*hex_literal::hex!\("([0-9a-f]{64})"\)', src).group(1)
    assert len(chunks) == 6
    return {"chunks": chunks, "chunk0_sha256": sha}


def example_job_dag():
    """The job DAG `plan_jobs` left in the dump (counter / goal / next-jobs triplets, proof_store.rs:41-87): per job
    group (topic, circuit_type, group_id, sub_group_id) the number of completions it waits for (`goal`) and the jobs it
    releases then. Data only."""
    b = open(f"{REF}/qbench_data/example.bin", "rb").read()
    r = Rd(b)
    r.u64(); r.u32(); [r.u64() for _ in range(6)]
    groups = {}
    for _ in range(r.u64()):
        k = parse_key(r.raw(24))
        val = r.raw(r.u64())
        if k["data_type"] != 16:
            continue
        g = groups.setdefault((k["topic"], k["circuit_type"], k["group_id"], k["sub_group_id"]), {})
        if k["data_index"] == 1:
            g["goal"] = struct.unpack("<I", val)[0]
        elif k["data_index"] == 2:
            m = struct.unpack("<Q", val[:8])[0]
            assert len(val) == 8 + 24 * m
            nxt = [parse_key(val[8 + 24 * i:8 + 24 * (i + 1)]) for i in range(m)]
            g["next"] = [[x["topic"], x["circuit_type"], x["group_id"], x["sub_group_id"], x["task_index"]] for x in nxt]
    return [{"group": list(k), "goal": v["goal"], "next": v["next"]} for k, v in sorted(groups.items())]


def main():
    if not os.path.isdir(REF):
        sys.exit("reference tree not present; golden files are already committed")
    json.dump(zero_hashes(), open(f"{OUT}/poseidon_zero_hashes.json", "w"))
    json.dump(fingerprints(), open(f"{OUT}/circuit_fingerprints.json", "w"), indent=1)
    json.dump(json_cases("city_common_circuit/src/hash/merkle/gadgets/merkle_proof.rs"),
              open(f"{OUT}/merkle_proofs.json", "w"))
    json.dump(json_cases("city_common_circuit/src/hash/merkle/gadgets/delta_merkle_proof.rs"),
              open(f"{OUT}/delta_merkle_proofs.json", "w"))
    cfg, index, deltas, proofs = example_bin()
    json.dump({"config": cfg, "entries": index}, open(f"{OUT}/example_dump_index.json", "w"))
    json.dump(deltas, open(f"{OUT}/example_delta_merkle.json", "w"))
    json.dump(example_job_dag(), open(f"{OUT}/example_job_dag.json", "w"), indent=0)
    # the whole dump, verbatim (data): every reference ProofWithPublicInputs is a slice of it
    import shutil
    shutil.copyfile(f"{REF}/qbench_data/example.bin", f"{OUT}/qbench_example.bin")
    kept = [{"file": "qbench_example.bin", "offset": off, **k, "len": len(v)} for k, v, off in proofs]
    json.dump(kept, open(f"{OUT}/example_proofs.json", "w"), indent=1)
    json.dump(groth16_samples(), open(f"{OUT}/groth16_proof_samples.json", "w"), indent=1)
    json.dump(groth16_verifier_data(), open(f"{OUT}/groth16_verifier_data.json", "w"), indent=1)
    json.dump(whitelist_tree(), open(f"{OUT}/sighash_whitelist_tree.json", "w"))
    print("zero hashes 2x128; fingerprints", len(fingerprints()), "; example entries", len(index),
          "; delta witnesses", len(deltas), "; proofs kept", len(kept), "of", len(proofs))


if __name__ == "__main__":
    main()
