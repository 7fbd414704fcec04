#!/usr/bin/env python3
"""The recurrence GEMM of the text tower (h[0:n_act] x W_hh^T: N = 3072, K = 1024, M = the active prefix) per kernel choice
(itr_gemm_nt_algo): where does the library's own choice leave time?  Run on the GPU box."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "image-text-retrieval_amd"))
import torch
from itr_amd import ops
dev = torch.device("cuda:0")
N, K = 3072, 1024
b = torch.randn(N, K, device=dev); bias = torch.randn(N, device=dev)
for M in (25000, 12000, 8000, 5000, 4000, 3000, 2000, 1400, 1000, 500, 200):
    a = torch.randn(M, K, device=dev)
    row = []
    for algo in (None, "tile", "stream_plain", "stream_xcd"):
        try:
            for _ in range(3): ops.linear(a, b, bias, algo=algo)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): ops.linear(a, b, bias, algo=algo)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 20
            row.append("%s %.1f us %.0f TF" % (algo or "auto", ms * 1e3, 2.0 * M * N * K / ms / 1e9))
        except Exception as e:
            row.append("%s: %s" % (algo, str(e)[:40]))
    print("M=%6d  " % M + "  |  ".join(row))
