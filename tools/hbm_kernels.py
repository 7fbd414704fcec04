#!/usr/bin/env python3
"""The HBM-bound kernels of the metric path alone on the GPU, against the 8 TB/s HBM3E peak (north_star: "rocprof HBM GB/s
... reported against chip peak"): algorithmic bytes per launch / HIP-event time, 20 launches after 3 warm-ups, at the sizes of the 5k x 25k step.

    python3 tools/hbm_kernels.py          (run on the GPU box; prints a markdown table)"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "image-text-retrieval_amd"))
import torch
from itr_amd import ops

HBM_PEAK = 8000.0          # GB/s, MI355X_MICROARCH.md
dev = torch.device("cuda", 0)
torch.manual_seed(0)


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(reps):
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2]


rows = []
# K1 l2norm of the region features (modalmodule/utils.py:4-8): one read + one write of 5 000 x 36 x 2048 fp32
x = torch.randn(5000 * 36, 2048, device=dev)
ms = timed(lambda: ops.l2norm(x))         # (allocates its output from torch's caching pool: no hipMalloc inside the timed region after the warm-ups)
rows.append(("norm_rows_kernel (l2norm, 180 000 x 2048)", 2 * x.numel() * 4, ms))
del x
# K9 rank counts (metricmodule/evaluation.py:156-222): ONE pass over the 5 000 x 25 000 fp32 matrix serves both directions (round 5)
S = torch.randn(5000, 25000, device=dev)
ms_both = timed(lambda: ops.rank_counts(S))
rows.append(("rank stage as the step runs it (ops.rank_counts(S): rank_prepare_kernel + rank_fused_kernel + rank_rows_finish_kernel, 5 small allocations), 5 000 x 25 000",
             S.numel() * 4, ms_both))
s_gt = ops.gather_gt(S)
bufs = ops.rank_counts(S, s_gt=s_gt)
ms_k = timed(lambda: ops.rank_counts(S, s_gt=s_gt, t2i_rank=bufs[2], t2i_best=bufs[3]))
rows.append(("the sharded form: GT scores given, caller-owned column accumulators (the same three kernels)", S.numel() * 4, ms_k))
print("| kernel (launch) | algorithmic bytes | median ms | GB/s | of 8 TB/s |")
print("|---|---|---|---|---|")
for name, b, ms in rows:
    print("| %s | %.3f GB | %.4f | %.0f | %.3f |" % (name, b / 1e9, ms, b / ms / 1e6, b / ms / 1e6 / HBM_PEAK))
print()
print("(rank_fused_kernel alone: rocprofv3 kernel table of the same round, profiles/rNN/README.md)")
