import os, sys, numpy as np, torch
sys.path.insert(0, "image-text-retrieval_amd")
from itr_amd import ops
dev = torch.device("cuda:0")
Ni = 1000; Nc, D = 5 * Ni, 1024
rng = np.random.RandomState(0)
lens = rng.randint(6, 21, size=Nc); off = np.concatenate([[0], np.cumsum(lens)[:-1]]); n_rows = int(lens.sum())
img = ops.l2norm(torch.randn(Ni, 36, D, device=dev)); words = ops.l2norm(torch.randn(n_rows, D, device=dev))
plan = ops.ScanPlan(off, lens, n_rows, dev); ws = ops.scan_prepare(img, words, plan)
for dbg in ("0", "1"):
    os.environ["ITR_SCAN_DEBUG"] = dbg
    for abl in ("0", "5", "9"):
        os.environ["ITR_SCAN_BF16_ABLATE"] = abl
        for _ in range(2): ops.scan_xattn_scores(img, words, plan, workspace=ws, precision='bf16x3')
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3): ops.scan_xattn_scores(img, words, plan, workspace=ws, precision='bf16x3')
        e1.record(); torch.cuda.synchronize()
        print("epilogue %s  ablate %s (5 = no global loads, 9 = no MFMA): %.2f ms" % ("off" if dbg == "1" else "on", abl, e0.elapsed_time(e1) / 3))
