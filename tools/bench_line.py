#!/usr/bin/env python3
"""stdin: the output of bench.py -> one short summary line (workload, ms per step, roofline fraction, kernel ms, recall parity)."""
import json
import sys
d = json.loads([ln for ln in sys.stdin if ln.startswith('{')][0])
rf = d["roofline"]
print(d["config"]["workload"], round(d["ms_per_step"], 2), round(rf["frac"], 4), round(rf.get("kernel_ms", rf.get("score_kernel_ms", 0.0)), 3),
      json.dumps((d.get("cpu_baseline") or {}).get("recall_parity"))[:300])
