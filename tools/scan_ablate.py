#!/usr/bin/env python3
"""Kernel-only timing of scan_xattn_kernel with ablation switches (run on the GPU box)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "image-text-retrieval_amd"))
import numpy as np, torch
from itr_amd import ops
dev = torch.device("cuda:0")
Ni = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
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
for name, flag in (("full", 0), ("full_nostagger", 8), ("no_epilogue", 1), ("no_gload", 2), ("no_gload_no_epi", 3), ("no_mfma", 4), ("no_mfma_no_epi", 5), ("no_gload_no_mfma_no_epi", 7)):
    os.environ["ITR_SCAN_DEBUG"] = str(flag)
    for _ in range(2):
        ops.scan_xattn_scores(img, words, plan, workspace=ws)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        ops.scan_xattn_scores(img, words, plan, workspace=ws)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    print("%-26s %8.2f ms   %6.1f TF/s (raw-dot flops, %d tiles)" % (name, ms, flop / ms / 1e9, plan.n_tiles))
os.environ["ITR_SCAN_DEBUG"] = "0"
for extra in (0, 2000, 4000, 30000):
    os.environ["ITR_SCAN_LDS_EXTRA"] = str(extra)
    for _ in range(2):
        ops.scan_xattn_scores(img, words, plan, workspace=ws)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        ops.scan_xattn_scores(img, words, plan, workspace=ws)
    e1.record(); torch.cuda.synchronize()
    print("LDS extra %6d B: %8.2f ms" % (extra, e0.elapsed_time(e1) / 3))
os.environ["ITR_SCAN_DEBUG"] = "16"; os.environ["ITR_SCAN_LDS_EXTRA"] = "0"
out = torch.zeros(Ni, Nc + 64, device=dev)
ops.scan_xattn_scores(img, words, plan, workspace=ws, out=out)
torch.cuda.synchronize()
cyc = out.view(torch.int64).flatten()[:8].cpu().numpy()
nblocks = ((Ni + 3) // 4) * plan.n_tiles
print("phase cycles per workgroup (s_memtime ticks @100MHz?):", [round(float(c) / nblocks, 1) for c in cyc[:5]], "blocks", nblocks)
os.environ["ITR_SCAN_DEBUG"] = "0"
import ctypes as C
from itr_amd import _lib
b, l = C.c_int(0), C.c_int(0)
_lib.check(_lib.load().itr_debug_scan_occupancy(C.byref(b), C.byref(l)))
print("runtime occupancy: %d blocks/CU, LDS %d B/block" % (b.value, l.value))
