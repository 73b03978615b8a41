#!/usr/bin/env python3
"""VALU instruction counts of the MSM kernels from a `rocprofv3 --pmc SQ_INSTS_VALU ... -- python3 tools/bench_msm.py 20` pass:
per group (G1: kernels instantiated on bls::Fp, G2: on bls::Fp2; the digit sort is shared and counted with both) the wave-
instructions of ONE multi-scalar multiplication of 2^20 points, per point, and per kernel. bench.py prices its own MSM timings with
these (groth16_kernels.msm_roofline): the bucket method is integer multiply-adds - v_mad_u64_u32, a quarter-rate instruction - so
the roof is their issue rate. usage: pmc_summary_msm.py <dir with the counter CSV> <out.json> <log_n> <msm calls per group>"""
import csv
import glob
import json
import os
import re
import sys


def main():
    src, out, log_n, calls = sys.argv[1], sys.argv[2], int(sys.argv[3]), float(sys.argv[4])
    acc = {}
    for path in glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True):
        seen = set()
        for row in csv.DictReader(open(path)):
            name = row["Kernel_Name"]
            if "msm::" not in name or "k_synthetic_points" in name or "k_points_to_mont" in name:
                continue
            short = re.sub(r"\(.*$", "", name).replace("void ", "")
            d = acc.setdefault(short, {"counters": {}, "dur_ns": 0.0, "launches": 0, "vgpr": int(row["VGPR_Count"])})
            d["counters"][row["Counter_Name"]] = d["counters"].get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
            key = (path, row["Dispatch_Id"])
            if key not in seen:
                seen.add(key)
                d["dur_ns"] += float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
                d["launches"] += 1
    n = 1 << log_n
    groups = {"G1": {}, "G2": {}}
    for k, d in acc.items():
        per_call = {"valu_instructions": d["counters"].get("SQ_INSTS_VALU", 0.0) / calls, "busy_us_under_pmc": d["dur_ns"] / 1e3 / calls,
                    "launches": d["launches"] / calls, "vgpr_count": d["vgpr"]}
        if d["counters"].get("SQ_WAVE_CYCLES"):
            per_call["wait_any_frac"] = d["counters"].get("SQ_WAIT_ANY", 0.0) / d["counters"]["SQ_WAVE_CYCLES"]
        if "<bls::Fp2>" in k:
            groups["G2"][k] = per_call
        elif "<bls::Fp>" in k:
            groups["G1"][k] = per_call
        else:   # the digit sort: launched once per MSM of either group -> half of its totals belongs to each
            half = dict(per_call, valu_instructions=per_call["valu_instructions"] / 2, busy_us_under_pmc=per_call["busy_us_under_pmc"] / 2,
                        launches=per_call["launches"] / 2)
            groups["G1"][k] = half
            groups["G2"][k] = dict(half)
    res = {"log_n": log_n, "points": n, "msm_calls_per_group_in_the_profiled_run": calls, "groups": {}}
    for g, ks in groups.items():
        tot = sum(v["valu_instructions"] for v in ks.values())
        res["groups"][g] = {"valu_wave_instructions_per_msm": tot, "valu_lane_ops_per_msm": tot * 64.0, "valu_lane_ops_per_point": tot * 64.0 / n,
                            "busy_us_per_msm_under_pmc": sum(v["busy_us_under_pmc"] for v in ks.values()), "kernels": ks}
    json.dump(res, open(out, "w"), indent=1)
    print("wrote", out, {g: round(v["valu_lane_ops_per_point"]) for g, v in res["groups"].items()})


if __name__ == "__main__":
    main()
