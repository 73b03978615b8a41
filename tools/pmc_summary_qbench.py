#!/usr/bin/env python3
"""Whole-proof counter evidence: merges the counter_collection CSVs of separate `rocprofv3 --pmc` passes over
the native q-bench harness (tools/cityprover_qbench --mode throughput, ONE context so that kernels do not overlap) into one
JSON: per kernel the counters SUMMED over the run and divided by the number of proofs the run made (VALU instructions,
HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE per MI355X_MICROARCH.md's gfx950 note, busy time), the totals per proof, and for the
quotient kernels (A8) the traffic against the algorithmic bytes of SURVEY.md section 8(d): every column the piece needs, read
once. Stamped with the hash of ALL kernel sources of the proving path; bench.py refuses the file when they have changed.

usage: pmc_summary_qbench.py <dir with the rocprofv3 outputs> <out.json> <proofs made by the profiled command> "<command>" """
import csv
import glob
import hashlib
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
QBENCH_SOURCES = ["gl.h", "poseidon.h", "poseidon_tables.h", "poseidon_coop.h", "merkle.h", "ntt.h", "ntt16.h", "fri.h", "zs.h", "quotient.h", "gates.h",
                  "prover_tail.inc", "fri_engine.inc"]
N_LDE, NC, N_CONST, N_ROUTED, N_WIRES, N_ZS = 1 << 15, 2, 5, 80, 135, 20   # the product shape
GATE_NAMES = ["noop", "constant", "public_input", "arithmetic", "poseidon", "comparison", "u32_arithmetic", "u32_range_check", "u32_add_many",
              "u32_subtraction", "u32_interleave", "uninterleave_to_u32", "uninterleave_to_b32", "arithmetic_ext", "mul_ext", "base_sum",
              "random_access", "reducing", "reducing_ext", "poseidon_mds", "coset_interpolation", "exponentiation"]


def source_hash(files=QBENCH_SOURCES):
    h = hashlib.sha256()
    for f in files:
        src = open(os.path.join(ROOT, "city-rollup_amd", "csrc", f), "r").read()
        src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
        src = re.sub(r"//[^\n]*", " ", src)
        h.update(" ".join(src.split()).encode())
    return h.hexdigest()[:16]


def short(full):
    n = re.sub(r"\(.*$", "", full).replace("void ", "")
    m = re.match(r".*k_quot_gate<\(?.*?\)?(\d+)>", n)
    if m:
        return "k_quot_gate<%s>" % GATE_NAMES[int(m.group(1))]
    n = re.sub(r"^(\w+::)+", "", n)
    return n


def gate_wires():
    import synth_gates as SG
    return {GATE_NAMES[g[0]]: SG.gate_num_wires(g) for g in SG.CITY_COMMON}


def main():
    src, out, proofs = sys.argv[1], sys.argv[2], float(sys.argv[3])
    cmd = sys.argv[4] if len(sys.argv) > 4 else ""
    acc = {}
    for path in glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True):
        seen = set()
        for row in csv.DictReader(open(path)):
            k = short(row["Kernel_Name"])
            d = acc.setdefault(k, {"counters": {}, "dur_ns": {}, "launches": {}})
            d["counters"][row["Counter_Name"]] = d["counters"].get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
            key = (path, row["Dispatch_Id"])
            if key not in seen:
                seen.add(key)
                d["dur_ns"][path] = d["dur_ns"].get(path, 0.0) + float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
                d["launches"][path] = d["launches"].get(path, 0) + 1
    gw = gate_wires()
    kernels, tot = {}, {"valu_instructions": 0.0, "hbm_bytes": 0.0, "busy_us": 0.0}
    for k, d in sorted(acc.items()):
        c = d["counters"]
        r = {"launches": max(d["launches"].values()), "busy_us_per_proof": max(d["dur_ns"].values()) / 1e3 / proofs}
        if "SQ_INSTS_VALU" in c:
            r["valu_instructions_per_proof"] = c["SQ_INSTS_VALU"] / proofs
            tot["valu_instructions"] += r["valu_instructions_per_proof"]
        if "FETCH_SIZE" in c or "WRITE_SIZE" in c:
            r["fetch_bytes_per_proof"] = 2.0 * 1024.0 * c.get("FETCH_SIZE", 0.0) / proofs
            r["write_bytes_per_proof"] = 1024.0 * c.get("WRITE_SIZE", 0.0) / proofs
            r["hbm_bytes_per_proof"] = r["fetch_bytes_per_proof"] + r["write_bytes_per_proof"]
            tot["hbm_bytes"] += r["hbm_bytes_per_proof"]
        for a, b_ in (("SQ_WAIT_ANY", "wait_any_frac"), ("SQ_WAIT_INST_ANY", "wait_inst_any_frac")):
            if a in c and c.get("SQ_WAVE_CYCLES"):
                r[b_] = c[a] / c["SQ_WAVE_CYCLES"]
        tot["busy_us"] += r["busy_us_per_proof"]
        # A8: algorithmic bytes of each piece = every column it needs read once + the accumulator it adds into
        alg = None
        if k == "k_quot_perm":
            alg = 8.0 * N_LDE * (N_ROUTED + N_ROUTED + N_ZS) + 8.0 * N_LDE * NC            # routed wires, sigmas, Z / partial products; writes the accumulator
        elif k.startswith("k_quot_gate<"):
            g = k[len("k_quot_gate<"):-1]
            if g in gw:
                alg = 8.0 * N_LDE * (gw[g] + N_CONST) + 2 * 8.0 * N_LDE * NC              # the gate's wires, selectors + constants; accumulator read + write
        elif k.startswith("k_quot_arith_group"):   # Constant + PublicInput + Arithmetic + ArithmeticExtension + MulExtension: their ~80 first wires ONCE
            alg = 8.0 * N_LDE * (max(gw.get(g, 0) for g in ("ARITHMETIC", "ARITHMETIC_EXT", "MUL_EXT", "CONSTANT", "PUBLIC_INPUT")) + N_CONST) + 2 * 8.0 * N_LDE * NC
        elif k == "k_quot_finish":
            alg = 2 * 8.0 * N_LDE * NC
        if alg is not None:
            r["algorithmic_bytes_per_proof"] = alg
            if "hbm_bytes_per_proof" in r:
                r["traffic_over_algorithmic"] = r["hbm_bytes_per_proof"] / alg
        kernels[k] = r
    quot = {k: v for k, v in kernels.items() if k.startswith("k_quot")}
    q_traffic = sum(v.get("hbm_bytes_per_proof", 0.0) for v in quot.values())
    once = 8.0 * N_LDE * (N_WIRES + N_CONST + N_ROUTED + N_ZS)   # (135 + 85 + 20) columns read once = 63 MB
    json.dump({"command": cmd, "proofs_in_the_profiled_run": proofs, "kernel_source_hash": source_hash(), "kernel_sources": QBENCH_SOURCES,
               "note": "rocprofv3 --pmc, one pass per counter group (SQ group; FETCH_SIZE; WRITE_SIZE), counters summed over every launch of the run "
                       "and divided by the proofs it made; FETCH_SIZE doubled (gfx950), KB -> bytes; busy_us = sum of the kernel's durations "
                       "under the counter pass (one context: no overlap)",
               "per_proof": {"valu_instructions": tot["valu_instructions"], "valu_lane_ops": 64.0 * tot["valu_instructions"], "hbm_bytes": tot["hbm_bytes"],
                             "kernel_busy_us": tot["busy_us"]},
               "quotient": {"hbm_bytes_per_proof": q_traffic, "columns_read_once_bytes": once, "traffic_over_columns_read_once": q_traffic / once if once else None,
                            "launches_per_batch": len(quot)},
               "kernels": kernels}, open(out, "w"), indent=1)
    print("wrote", out, len(kernels), "kernels")


if __name__ == "__main__":
    main()
