#!/usr/bin/env python3
"""Print a rocprofv3 kernel_stats.csv as ms per step: tools/prof_table.py <csv> <steps profiled> [rows]."""
import csv
import sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2])
top = int(sys.argv[3]) if len(sys.argv) > 3 else 25
tot = sum(int(r['TotalDurationNs']) for r in rows)
print('GPU time %.2f ms per step (%d kernels)' % (tot / steps / 1e6, len(rows)))
for r in rows[:top]:
    print('%-100s %7.1f calls %8.3f ms/step %6.2f%%' % (r['Name'][:100], int(r['Calls']) / steps, int(r['TotalDurationNs']) / steps / 1e6,
                                                      100.0 * int(r['TotalDurationNs']) / tot))
