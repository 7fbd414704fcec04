#!/usr/bin/env python3
"""Summarise rocprofv3 csv output dirs (kernel stats + per-kernel PMC sums) into a small text table."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
for f in glob.glob(os.path.join(root, "trace", "**", "*kernel_stats.csv"), recursive=True):
    print("== kernel stats (rocprofv3 --kernel-trace --stats):", os.path.relpath(f, root))
    print("   %-64s %7s %14s %14s %7s" % ("kernel", "calls", "total_ms", "avg_us", "pct"))
    for r in list(csv.DictReader(open(f)))[:16]:
        print("   %-64s %7s %14.3f %14.3f %7s" % (r["Name"][:64], r["Calls"], float(r["TotalDurationNs"]) / 1e6,
                                                  float(r["AverageNs"]) / 1e3, r["Percentage"]))
for d in sorted(glob.glob(os.path.join(root, "pmc*"))):
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        print("== counters (sum over dispatches / number of dispatches):", os.path.relpath(f, root))
        acc = defaultdict(lambda: defaultdict(float))
        disp = defaultdict(set)
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:60]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            disp[k].add(r["Dispatch_Id"])
        for k in acc:
            if any(s in k for s in ("scan_xattn", "gemm_nt", "rank_kernel", "rank_fused", "gru_gate", "scan_pack", "gram_kernel", "sgr_fused", "sgraf_loc")):
                n = max(1, len(disp[k]))
                print("   %s   (%d dispatches)" % (k, n))
                for c, v in sorted(acc[k].items()):
                    print("       %-28s per dispatch %.6g" % (c, v / n))
