#!/usr/bin/env python3
"""fp32 GEMM microbench (run on the GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "image-text-retrieval_amd"))
import torch
from itr_amd import ops
dev = torch.device("cuda:0")
for (M, N, K) in [(4096, 4096, 4096), (180000, 1024, 2048), (265000, 256, 1024), (265000, 256, 256), (20000, 256, 1024), (20000, 256, 256), (25000, 3072, 1024), (66000, 1024, 36)]:
    a = torch.randn(M, K, device=dev); b = torch.randn(N, K, device=dev); bias = torch.randn(N, device=dev)
    for _ in range(2): ops.linear(a, b, bias)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): ops.linear(a, b, bias)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print("M=%7d N=%5d K=%5d  %8.3f ms  %6.1f TF/s" % (M, N, K, ms, 2.0 * M * N * K / ms / 1e9))
