#!/usr/bin/env python3
"""N = 8 readiness that one GPU can check (VERDICT r3 #6a): does the all-gather make progress while the own-column scoring launch fills
every CU?  One process, a 1-rank RCCL group (ITR_FORCE_COLLECTIVES=1), the caption axis split over 8 virtual owners
(bench.py --virtual-split 8:V): the exchange is a REAL asynchronous all_gather_into_tensor of the full packed word buffer (1.3 GB at
5k x 25k) on RCCL's stream, started before the own-column SCAN launch.  Reported: the collective alone (nothing else on the GPU), the
own-column launch, and how long the scoring stream still waits for the collective after that launch has finished.

    python3 tools/exchange_overlap.py [workload] [owner]          (run on the GPU box)"""
import json
import os
import subprocess
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
wl = sys.argv[1] if len(sys.argv) > 1 else "scan_t2i_coco5k"
owner = sys.argv[2] if len(sys.argv) > 2 else "3"

# 1. the collective alone, same size, same group kind
alone = subprocess.run([sys.executable, "-c", """
import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29571")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
n = 8 * 40700 * 1024                      # 8 owners x (325 k words / 8) rows x 1024 floats
src = torch.randn(n, device="cuda"); dst = torch.empty_like(src)
for _ in range(2): dist.all_gather_into_tensor(dst, src)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ts = []
for _ in range(5):
    e0.record(); w = dist.all_gather_into_tensor(dst, src, async_op=True); w.wait(); e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
print("ALONE", n * 4, min(ts), sum(ts) / len(ts))
dist.destroy_process_group()
"""], capture_output=True, text=True, timeout=600)
print(alone.stdout.strip() or alone.stderr[-2000:])
env = dict(os.environ, ITR_FORCE_COLLECTIVES="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29572")
r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", wl, "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-variants",
                    "--no-other-configs", "--virtual-split", "8:" + owner], env=env, capture_output=True, text=True, timeout=1200)
line = [ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')]
if not line:
    print(r.stderr[-3000:])
    sys.exit(1)
d = json.loads(line[0])
print(json.dumps({"workload": wl, "virtual_split": d.get("virtual_split"), "ms_per_step": d["ms_per_step"], "exchange": d.get("exchange")}, indent=1))
