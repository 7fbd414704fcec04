#!/usr/bin/env python3
"""The recurrence GEMM of the text tower (h[0:n_act] x W_hh^T: N = 3072, K = 1024, M = the active prefix) per kernel choice
(itr_gemm_nt_algo): where does the library's own choice leave time?  Interleaved, two passes (the first pass of a new size runs
while the clock still ramps).  Run on the GPU box:  python tools/gemm_recur_bench.py [N K]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "image-text-retrieval_amd"))
import torch
from itr_amd import ops
dev = torch.device("cuda:0")
N, K = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (3072, 1024)
b = torch.randn(N, K, device=dev); bias = torch.randn(N, device=dev)


def t(a, algo, reps=20):
    for _ in range(3): ops.linear(a, b, bias, algo=algo)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): ops.linear(a, b, bias, algo=algo)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for M in (25000, 20000, 16000, 14000, 12000, 11000, 10000, 9000, 8000, 7000, 6000, 5000, 4000, 3000, 2000, 1000, 500):
    a = torch.randn(M, K, device=dev)
    best = {}
    for _pass in range(2):
        for algo in (None, "tile", "stream_plain"):
            try:
                ms = t(a, algo)
                best[algo] = min(best.get(algo, 1e9), ms)
            except Exception as e:
                best[algo] = float("nan")
    print("M=%6d  " % M + "  |  ".join("%s %.1f us %.0f TF" % (k or "auto", v * 1e3, 2.0 * M * N * K / v / 1e9) for k, v in best.items()))
