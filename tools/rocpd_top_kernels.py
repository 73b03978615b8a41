#!/usr/bin/env python3
"""Per-kernel summary (calls, total / average duration) of a rocprofv3 --kernel-trace run, from the rocpd sqlite
database(s) it leaves under the output directory.   usage: rocpd_top_kernels.py <rocprof output dir> "<header comment>" """
import glob
import os
import sqlite3
import sys


def main():
    out_dir, comment = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
    rows = {}
    for db in glob.glob(os.path.join(out_dir, "**", "*.db"), recursive=True):
        con = sqlite3.connect(db)
        views = [r[0] for r in con.execute("select name from sqlite_master where type in ('view', 'table')")]
        if "kernels" not in views:
            continue
        cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
        name = "name" if "name" in cols else "kernel_name"
        for kname, calls, total in con.execute(f"select {name}, count(*), sum(end - start) from kernels group by {name}"):
            c, t = rows.get(kname, (0, 0))
            rows[kname] = (c + calls, t + total)
    tot = sum(t for _, t in rows.values()) or 1
    if comment:
        print("# " + comment)
    print("Name,Calls,TotalDurationUs,AverageUs,Percentage")
    for kname, (c, t) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
        print(f"\"{kname}\",{c},{t / 1e3:.3f},{t / 1e3 / c:.3f},{100.0 * t / tot:.2f}")


if __name__ == "__main__":
    main()
