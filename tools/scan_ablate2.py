#!/usr/bin/env python3
"""Ablations at 1 vs 2 workgroups per CU."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "image-text-retrieval_amd"))
import numpy as np, torch
from itr_amd import ops
dev = torch.device("cuda:0")
PREC = sys.argv[1] if len(sys.argv) > 1 else "fp32"      # "bf16x3": the opt-in split-bf16 main loop
Ni = 1000
Nc, D = 5 * Ni, 1024
rng = np.random.RandomState(0)
lens = rng.randint(6, 21, size=Nc)
off = np.concatenate([[0], np.cumsum(lens)[:-1]])
n_rows = int(lens.sum())
img = ops.l2norm(torch.randn(Ni, 36, D, device=dev))
words = torch.randn(n_rows, D, device=dev) * 0.3
plan = ops.ScanPlan(off, lens, n_rows, dev)
ws = ops.scan_prepare(img, words, plan)
flop = Ni * n_rows * (2 * 36 * D)
for extra in (0, 4000):
    os.environ["ITR_SCAN_LDS_EXTRA"] = str(extra)
    for name, flag in (("full", 0), ("no_epilogue", 1)):      # (the load / MFMA ablation switches of round 1 are gone from the loop)
        os.environ["ITR_SCAN_DEBUG"] = str(flag)
        for _ in range(2):
            ops.scan_xattn_scores(img, words, plan, workspace=ws, precision=PREC)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            ops.scan_xattn_scores(img, words, plan, workspace=ws, precision=PREC)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 3
        print("blocks/CU=%d %-18s %8.2f ms   %6.1f TF/s" % (2 if extra == 0 else 1, name, ms, flop / ms / 1e9))

for extra in (0, 4000):
    os.environ["ITR_SCAN_LDS_EXTRA"] = str(extra)
    for flag in (16, 17):
        os.environ["ITR_SCAN_DEBUG"] = str(flag)
        out = torch.zeros(Ni, Nc + 64, device=dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.scan_xattn_scores(img, words, plan, workspace=ws, out=out, precision=PREC)
        e1.record()
        torch.cuda.synchronize()
        wall_ms = e0.elapsed_time(e1)
        cyc = out.view(torch.int64).flatten()[:8].cpu().numpy()
        nblocks = ((Ni + 3) // 4) * plan.n_tiles
        if cyc[7] > 0:
            slots = 256 * (2 if extra == 0 else 1)
            print("   wall %.2f ms; workgroup residency (sum of per-workgroup real time / (%d slots x wall)) = %.3f; "
                  "workgroups %d" % (wall_ms, slots, float(cyc[7]) * 1e-5 / (slots * wall_ms), nblocks))
            print("   shader clock during the kernel: %.3f GHz (sum of phase cycles / 100 MHz real-time ticks)" % (
                float(cyc[:7].sum()) / float(cyc[7]) * 0.1))
        # (the prologue, the K loop and the park are one asm statement now: slot 4 holds all three, slots 0 / 1 are empty)
        print("blocks/CU=%d flag %d: cycles per workgroup: prologue + K loop + park %.0f  E1 %.0f  E2a %.0f  E2b %.0f  E2 rest + E3 %.0f" % (
            2 if extra == 0 else 1, flag, (cyc[4] + cyc[0] + cyc[1]) / nblocks, cyc[2] / nblocks, cyc[5] / nblocks, cyc[6] / nblocks, cyc[3] / nblocks))
