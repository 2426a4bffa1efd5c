#!/usr/bin/env python
"""Kernel timeline of a rocprofv3 --kernel-trace CSV (GPU box): `python tools/trace_overlap.py <kernel_trace.csv> <out.txt>` -- the
steady-state tail as (start us, end us, queue, kernel) rows plus how much of the wall time had 2 kernels in flight."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = len(rows)
rows = rows[int(n * 0.6):]
t0 = int(rows[0]["Start_Timestamp"])
ev = []
with open(sys.argv[2], "w") as out:
    for r in rows:
        s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
        ev += [(s, 1), (e, -1)]
        out.write("%9.1f %9.1f q%s %s\n" % (s / 1e3, e / 1e3, r.get("Queue_Id", "?"), r["Kernel_Name"].replace("void l2hmc::", "")[:80]))
    ev.sort()
    depth, last, hist = 0, ev[0][0], {}
    for t, d in ev:
        hist[depth] = hist.get(depth, 0) + (t - last)
        depth, last = depth + d, t
    tot = float(sum(hist.values()))
    out.write("time share by number of kernels in flight: %s\n" % {k: round(v / tot, 3) for k, v in sorted(hist.items())})
print(open(sys.argv[2]).read()[-300:])
