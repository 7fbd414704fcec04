import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][0])
print(d["config"]["workload"], round(d["ms_per_step"],2), round(d["roofline"]["frac"],4), round(d["roofline"]["score_kernel_ms"],3), json.dumps(d.get("cpu_baseline",{}).get("recall_parity"))[:600])
