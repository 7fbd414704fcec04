#!/usr/bin/env python3
"""sgraf_loc_kernel: time per 32-wide slice of D.  Run under rocprofv3 --kernel-trace --stats once with D = 1024 and once with
D = 2048 (same images / captions / tiles): the difference of the kernel's average durations is 32 slices of pure loop.
    rocprofv3 --kernel-trace --stats -d OUT -- python3 tools/loc_slice_time.py 1024"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "image-text-retrieval_amd"))
import numpy as np
import torch
import bench
from itr_amd import ops

D = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dev = torch.device("cuda:0")
torch.manual_seed(0)
lengths, _ = bench.make_captions(5000, 8481)
off = np.concatenate([[0], np.cumsum(lengths)[:-1]]).astype(np.int64)
n_rows = int(lengths.sum())
img = ops.l2norm(torch.randn(64, 36, D, device=dev))
words = ops.l2norm(torch.randn(n_rows, D, device=dev))
w = {k: v.to(dev) for k, v in bench.make_sgraf_weights("SAF", D=D).items()}
plan = ops.ScanPlan(off, lengths, n_rows, dev)
for _ in range(3):
    S = ops.sgraf_scores(img, words, plan, w, "SAF", 3)
torch.cuda.synchronize()
M = 16 * plan.n_tiles * 64
print("D", D, "tiles", plan.n_tiles, "rows per launch", M, "flop per launch %.1f G" % (2 * M * (256 + 36) * D / 1e9), "checksum", float(S.double().sum()))
