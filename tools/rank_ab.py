#!/usr/bin/env python3
"""Time the rank stage (5 000 x 25 000 and 1 000 x 5 000) of the package under <root> (default: this tree; tools/ab/<name> for an
experiment build):  python3 tools/rank_ab.py [root]"""
import os
import sys
ROOT = os.path.abspath(sys.argv[1]) if len(sys.argv) > 1 else os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "image-text-retrieval_amd"))
import torch
from itr_amd import ops
dev = torch.device("cuda", 0)
torch.manual_seed(0)


def timed(fn, reps=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2], ts[0]


for n in (5000, 1000):
    S = torch.randn(n, 5 * n, device=dev)
    s_gt = ops.gather_gt(S)
    bufs = ops.rank_counts(S, s_gt=s_gt)
    med, best = timed(lambda: ops.rank_counts(S, s_gt=s_gt, t2i_rank=bufs[2], t2i_best=bufs[3]))
    print("%s  %d x %d: prepare + fused + finish  median %.1f us  min %.1f us  -> %.0f GB/s of one read" % (
        os.path.basename(ROOT), n, 5 * n, med * 1e3, best * 1e3, S.numel() * 4 / med / 1e6))
