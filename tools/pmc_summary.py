#!/usr/bin/env python3
"""Merges the counter_collection CSVs of separate `rocprofv3 --pmc` passes into one per-kernel JSON summary, stamped with
the hash of the kernel sources (bench.py refuses the summary when the sources have changed since).

usage: pmc_summary.py <dir with the rocprofv3 outputs of every pass> <out.json> "<command that was profiled>"

Per kernel (short names as cp_profile uses them): counters averaged per launch; derived: clock estimate
(GRBM_GUI_ACTIVE / 8 XCDs / duration, MI355X_MICROARCH.md 'DVFS give-back'), VALU lane-ops/s, VALU issue-slot utilisation
(instructions x 2 cycles / (1024 SIMDs x cycles)), wave stall fractions; HBM bytes = 2 x FETCH_SIZE (gfx950 reports
half of a coalesced streaming read) + WRITE_SIZE, both in KB (MI355X_MICROARCH.md section HBM)."""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SHORT = [("k_leaf_hash_cols", "leaf_hash_cols"), ("k_levels_coop", "merkle_levels_coop"), ("k_level_coop", "merkle_level_coop"), ("k_level_fused", "merkle_level_fused"), ("k_level", "merkle_level"),
         ("k_dif_pass16<8, 4, false", "ntt16_cols"), ("k_dif_pass16<12, 0, true", "ntt16_rows"), ("k_quot_gate", None),
         ("k_pow_grind", "fri_pow_grind")]


def short_name(full):
    for pat, name in SHORT:
        if pat in full:
            return name or full.split("(")[0]
    return full.split("(")[0].replace("void ", "")


def main():
    src, out, cmd = sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else ""
    acc = {}   # kernel -> counter -> [sum, launches]; plus durations
    for path in glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True):
        seen = set()
        for row in csv.DictReader(open(path)):
            k = short_name(row["Kernel_Name"])
            d = acc.setdefault(k, {"counters": {}, "dur_ns": 0.0, "dur_n": 0, "vgpr": int(row["VGPR_Count"]), "sgpr": int(row["SGPR_Count"]),
                                   "lds": int(row["LDS_Block_Size"]), "grid": int(row["Grid_Size"])})
            c = d["counters"].setdefault(row["Counter_Name"], [0.0, 0])
            c[0] += float(row["Counter_Value"])
            c[1] += 1
            key = (path, row["Dispatch_Id"])
            if key not in seen:
                seen.add(key)
                d["dur_ns"] += float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
                d["dur_n"] += 1
    import bench
    res = {}
    for k, d in acc.items():
        r = {name: v[0] / v[1] for name, v in d["counters"].items()}
        r["launches_seen"] = max(v[1] for v in d["counters"].values())
        r["avg_duration_us_under_pmc"] = d["dur_ns"] / max(d["dur_n"], 1) / 1e3
        r["vgpr_count"], r["sgpr_count"], r["lds_bytes"], r["grid_size"] = d["vgpr"], d["sgpr"], d["lds"], d["grid"]
        dur_s = r["avg_duration_us_under_pmc"] * 1e-6
        if "GRBM_GUI_ACTIVE" in r and dur_s > 0:
            r["clock_GHz_est"] = r["GRBM_GUI_ACTIVE"] / 8.0 / dur_s / 1e9
        if "SQ_INSTS_VALU" in r and dur_s > 0:
            r["valu_lane_ops_per_s"] = r["SQ_INSTS_VALU"] * 64.0 / dur_s
        if "SQ_INSTS_VALU" in r and "clock_GHz_est" in r and dur_s > 0:
            # issue slots used: one wave64 VALU instruction occupies its SIMD for 2 cycles; 1024 SIMDs
            r["valu_issue_utilization"] = r["SQ_INSTS_VALU"] * 2.0 / (1024.0 * r["clock_GHz_est"] * 1e9 * dur_s)
        if "SQ_ACTIVE_INST_VALU" in r and "SQ_WAVE_CYCLES" in r:
            r["valu_frac_of_wave_cycles"] = r["SQ_ACTIVE_INST_VALU"] / r["SQ_WAVE_CYCLES"]
        if "SQ_WAVE_CYCLES" in r:
            for src_name, dst in (("SQ_WAIT_ANY", "wait_any_frac"), ("SQ_WAIT_INST_ANY", "wait_inst_any_frac"), ("SQ_ACTIVE_INST_ANY", "active_any_frac")):
                if src_name in r:
                    r[dst] = r[src_name] / r["SQ_WAVE_CYCLES"]
        if "FETCH_SIZE" in r or "WRITE_SIZE" in r:
            r["fetch_bytes"] = 2.0 * 1024.0 * r.get("FETCH_SIZE", 0.0)
            r["write_bytes"] = 1024.0 * r.get("WRITE_SIZE", 0.0)
            r["hbm_bytes"] = r["fetch_bytes"] + r["write_bytes"]
        res[k] = r
    json.dump({"command": cmd, "kernel_source_hash": bench.kernel_source_hash(), "kernel_sources": bench.KERNEL_SOURCES,
               "note": "rocprofv3 --pmc, one pass per counter group (SQ group; FETCH_SIZE; WRITE_SIZE); values per launch; FETCH_SIZE doubled "
                       "(gfx950), KB -> bytes", "kernels": res}, open(out, "w"), indent=1)
    print("wrote", out, sorted(res))


if __name__ == "__main__":
    main()
