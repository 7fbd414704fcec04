#!/usr/bin/env python3
"""Summarise rocprofv3 output dirs (kernel stats + per-kernel PMC sums) into a small text table."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
for f in glob.glob(os.path.join(root, "trace", "**", "*kernel_stats.csv"), recursive=True):
    print("== kernel stats:", f)
    rows = list(csv.DictReader(open(f)))
    for r in rows[:14]:
        print("  %-60s calls %6s  total %12.3f ms  avg %10.3f us  %5s%%" % (
            r["Name"][:60], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, r["Percentage"]))
for d in sorted(glob.glob(os.path.join(root, "pmc*"))):
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        print("== counters:", f)
        acc = defaultdict(lambda: defaultdict(float))
        cnt = defaultdict(int)
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:50]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        for k in acc:
            if any(s in k for s in ("scan_xattn", "gemm_nt", "rank", "gru_gate")):
                print("  ", k)
                for c, v in sorted(acc[k].items()):
                    print("      %-28s %.6g" % (c, v))
