#!/usr/bin/env python3
"""Cycles per workgroup and phase of the i2t SCAN kernel (ITR_SCAN_DEBUG=16: the kernel adds its per-phase s_memtime deltas into
the first 8 slots of the output).  Run on the GPU box."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "image-text-retrieval_amd"))
import numpy as np, torch
from itr_amd import ops
dev = torch.device("cuda:0")
Ni = 1000
Nc, D = 5 * Ni, 1024
rng = np.random.RandomState(0)
lens = rng.randint(6, 21, size=Nc)
off = np.concatenate([[0], np.cumsum(lens)[:-1]])
n_rows = int(lens.sum())
img = ops.l2norm(torch.randn(Ni, 36, D, device=dev))
words = torch.randn(n_rows, D, device=dev) * 0.3
plan = ops.ScanPlan(off, lens, n_rows, dev)
for ca in ("t2i", "i2t"):
    ws = ops.scan_prepare(img, words, plan, ca)
    for flag in (0, 1):
        os.environ["ITR_SCAN_DEBUG"] = str(flag)
        for _ in range(2):
            ops.scan_xattn_scores(img, words, plan, cross_attn=ca, workspace=ws)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            ops.scan_xattn_scores(img, words, plan, cross_attn=ca, workspace=ws)
        e1.record(); torch.cuda.synchronize()
        print("%s %-12s %8.2f ms" % (ca, "full" if flag == 0 else "no epilogue", e0.elapsed_time(e1) / 3))
    os.environ["ITR_SCAN_DEBUG"] = "16"
    out = torch.zeros(Ni, Nc + 64, device=dev)
    ops.scan_xattn_scores(img, words, plan, cross_attn=ca, workspace=ws, out=out)
    torch.cuda.synchronize()
    cyc = out.view(torch.int64).flatten()[:8].cpu().numpy()
    nblocks = ((Ni + 3) // 4) * plan.n_tiles
    names = ["loop+park", "-", "E1", "last phase", "(loop)", "phase 5", "phase 6"]
    print("   cycles per workgroup:", "  ".join("%s %.0f" % (n, c / nblocks) for n, c in zip(names, cyc[:7])))
    os.environ["ITR_SCAN_DEBUG"] = "0"
