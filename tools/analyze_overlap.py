#!/usr/bin/env python3
"""Given a rocprofv3 kernel_trace.csv: how much do kernels of different streams overlap in time?"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
ev = []
for r in rows:
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Stream_Id"], r["Kernel_Name"][:40]))
ev.sort()
# split into phases by gaps > 50 ms
phases, cur = [], [ev[0]]
for e in ev[1:]:
    if e[0] - max(x[1] for x in cur[-50:]) > 50e6:
        phases.append(cur); cur = [e]
    else:
        cur.append(e)
phases.append(cur)
for ph in phases:
    t0, t1 = ph[0][0], max(e[1] for e in ph)
    busy = sum(e[1] - e[0] for e in ph)
    cur_s, cur_e, union = None, None, 0
    for s, e, *_ in ph:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                union += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    union += cur_e - cur_s
    q = defaultdict(int)
    for e in ph:
        q[(e[2], e[3])] += 1
    print("kernels", len(ph), "span_ms %.1f sum_kernel_ms %.1f union_ms %.1f avg_concurrency %.2f" %
          ((t1 - t0) / 1e6, busy / 1e6, union / 1e6, busy / union), "streams", len(q))
