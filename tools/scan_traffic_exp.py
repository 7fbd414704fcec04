#!/usr/bin/env python3
"""does the fabric traffic of the headline kernel (1.17 TB per launch at 5k x 25k, 454 x the algorithmic bytes,
absorbed by L2 / Infinity Cache) cost time or clock?  The kernel is run unchanged and with ITR_SCAN_DEBUG=32, where every
workgroup reads the operands of the FIRST 16 images x 4 column tiles (L2-resident) instead of its own: identical instruction
stream and matrix work, (almost) no fabric traffic.  For each arm: HIP-event time of the launch, the shader clock sustained
inside it (s_memtime / s_memrealtime of an instrumented launch) and the board power sampled from hwmon while it runs.

    python3 tools/scan_traffic_exp.py                       # both arms, interleaved, 3 rounds
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d OUT -- python3 tools/scan_traffic_exp.py --arm resident --launches 2
"""
import argparse, glob, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "image-text-retrieval_amd"))
import numpy as np, torch
from itr_amd import ops

ap = argparse.ArgumentParser()
ap.add_argument("--arm", default="both", choices=["both", "normal", "resident"])
ap.add_argument("--launches", type=int, default=3)
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--n-img", type=int, default=5000)
args = ap.parse_args()
dev = torch.device("cuda:0")
Ni, D = args.n_img, 1024
Nc = 5 * Ni
rng = np.random.RandomState(0)
lens = rng.randint(6, 21, size=Nc)
off = np.concatenate([[0], np.cumsum(lens)[:-1]])
n_rows = int(lens.sum())
g = torch.Generator(device=dev); g.manual_seed(0)
img = ops.l2norm(torch.randn(Ni, 36, D, device=dev, generator=g))
words = torch.randn(n_rows, D, device=dev, generator=g) * 0.3
plan = ops.ScanPlan(off, lens, n_rows, dev)
ws = ops.scan_prepare(img, words, plan, "t2i")
out = torch.zeros(Ni, Nc + 64, device=dev)


def power_files():
    return sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_average") + glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_input"))


class Sampler(threading.Thread):
    def __init__(self):
        super().__init__(daemon=True)
        self.files, self.vals, self.stop = power_files(), [], False

    def run(self):
        while not self.stop:
            try:
                self.vals.append(max(int(open(f).read()) for f in self.files) / 1e6)
            except Exception:
                pass
            time.sleep(0.02)


def launch(flag):
    os.environ["ITR_SCAN_DEBUG"] = str(flag)
    ops.scan_xattn_scores(img, words, plan, cross_attn="t2i", workspace=ws, out=out)


def run_arm(flag, n):
    launch(flag); torch.cuda.synchronize()
    smp = Sampler(); smp.start()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        launch(flag)
    e1.record(); torch.cuda.synchronize()
    smp.stop = True; smp.join()
    ms = e0.elapsed_time(e1) / n
    launch(flag | 16); torch.cuda.synchronize()
    c = out.view(torch.int64).flatten()[:8].cpu().numpy().astype(np.float64)
    mhz = 100.0 * c[:7].sum() / c[7]
    pw = np.asarray(smp.vals[len(smp.vals) // 4:]) if smp.vals else np.zeros(0)
    return ms, mhz, (float(pw.mean()) if len(pw) else float("nan")), len(pw)


arms = {"normal": 0, "resident": 32}
if args.arm != "both":
    for _ in range(args.launches):
        launch(arms[args.arm])
    torch.cuda.synchronize()
    print("arm", args.arm, "launches", args.launches)
    sys.exit(0)
print("SCAN t2i %d x %d, D = %d, %d words; power files: %s" % (Ni, Nc, D, n_rows, power_files() or "none readable"))
print("%-10s %10s %12s %12s" % ("arm", "ms/launch", "clock MHz", "power W"))
for r in range(args.rounds):
    for name, flag in arms.items():
        ms, mhz, pw, n = run_arm(flag, args.launches)
        print("%-10s %10.2f %12.1f %12.1f   (%d power samples)" % (name, ms, mhz, pw, n))
os.environ["ITR_SCAN_DEBUG"] = "0"
