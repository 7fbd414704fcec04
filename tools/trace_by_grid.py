#!/usr/bin/env python3
"""Aggregate a rocprofv3 kernel trace by (kernel, grid): tools/trace_by_grid.py <dir with *kernel_trace.csv> <steps> [rows].
Launch shapes that are slow for their size (a handful of workgroups running for tens of microseconds) show up here and not in --stats."""
import collections
import csv
import glob
import sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
steps = float(sys.argv[2])
top = int(sys.argv[3]) if len(sys.argv) > 3 else 24
agg = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(f)):
    k = (r["Kernel_Name"][:44], r["Grid_Size_X"], r["Grid_Size_Y"])
    agg[k][0] += 1
    agg[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
tot = sum(v[1] for v in agg.values())
print("GPU time %.2f ms per step" % (tot / steps))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print("%-46s grid %8s x %4s  calls/step %7.1f  ms/step %8.3f  avg us %8.1f" % (k[0], k[1], k[2], v[0] / steps, v[1] / steps, 1e3 * v[1] / v[0]))
