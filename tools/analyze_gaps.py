#!/usr/bin/env python3
"""Idle gaps between kernels in a rocprofv3 kernel trace (rocpd sqlite database): for the last `window_ms` of the
trace, GPU-busy union, and the largest gaps with the kernels on either side. Usage: analyze_gaps.py results.db [window_ms]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
window = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 200e6
rows = sorted(db.execute("select start, end, name, stream_id from kernels"))
t_end = max(r[1] for r in rows)
rows = [r for r in rows if r[0] >= t_end - window]
t0 = rows[0][0]
busy, cur_e, gaps = 0, None, []
prev = None
for s, e, name, st in rows:
    if cur_e is None:
        cur_s, cur_e = s, e
    elif s > cur_e:
        gaps.append((s - cur_e, prev, name, (cur_e - t0) / 1e6))
        busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
    prev = name
busy += cur_e - cur_s
span = t_end - t0
print("window %.1f ms: %d kernels, busy %.1f ms (%.0f %%), %d streams" % (span / 1e6, len(rows), busy / 1e6, 100 * busy / span,
                                                                       len(set(r[3] for r in rows))))
tot = sum(g[0] for g in gaps)
print("idle %.1f ms in %d gaps; gaps > 100 us: %.1f ms" % (tot / 1e6, len(gaps), sum(g[0] for g in gaps if g[0] > 1e5) / 1e6))
short = lambda n: n.split("(")[0][-38:]
for g in sorted(gaps, reverse=True)[:25]:
    print("  %8.1f us at %7.2f ms   %-38s -> %s" % (g[0] / 1e3, g[3], short(g[1]), short(g[2])))
