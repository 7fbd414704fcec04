#!/usr/bin/env python3
"""Split-bf16 GEMM study (STUDY_SPLIT_PRECISION.md): throughput and accuracy of gemm_bf16_kernel (terms 3 / 1) next to the exact fp32
MFMA GEMM on the tower / score shapes of the bench workloads.  Run on the GPU box."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "image-text-retrieval_amd"))
import torch
from itr_amd import ops

dev = torch.device("cuda", 0)
shapes = [("image tower fc  180000 x 1024 x 2048", 180000, 1024, 2048),
          ("cosine scores     5000 x 25000 x 1024", 5000, 25000, 1024),
          ("BERT FFN        32768 x 3072 x  768", 32768, 3072, 768),
          ("SGR projection  265000 x  256 x 256", 265000, 256, 256)]


def timed(fn, n=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


for name, M, N, K in shapes:
    torch.manual_seed(0)
    a = torch.randn(M, K, device=dev)
    b = torch.randn(N, K, device=dev)
    a = a / a.norm(dim=1, keepdim=True)
    b = b / b.norm(dim=1, keepdim=True)
    flop = 2.0 * M * N * K
    t32 = timed(lambda: ops.linear(a, b))
    pa, pb = ops.split_bf16(a), ops.split_bf16(b)
    t3 = timed(lambda: ops.gemm_nt_bf16(pa, pb, terms=3))
    t1 = timed(lambda: ops.gemm_nt_bf16(pa, pb, terms=1))
    ts = timed(lambda: (ops.split_bf16(a), ops.split_bf16(b)))
    qa, qb = ops.split_f16(a), ops.split_f16(b)
    t16 = timed(lambda: ops.gemm_nt_f16x3(qa, qb))
    ts16 = timed(lambda: (ops.split_f16(a), ops.split_f16(b)))
    ref = ops.linear(a, b)
    e3 = (ops.gemm_nt_bf16(pa, pb, terms=3) - ref).abs().max().item()
    e1 = (ops.gemm_nt_bf16(pa, pb, terms=1) - ref).abs().max().item()
    # a 4096-row fp64 sample as ground truth for both
    rows = torch.randperm(M, device=dev)[:2048]
    cols = torch.randperm(N, device=dev)[:min(N, 2048)]
    w64 = a[rows].double() @ b[cols].double().t()
    e32_64 = (ref[rows][:, cols].double() - w64).abs().max().item()
    e3_64 = (ops.gemm_nt_bf16(pa, pb, terms=3)[rows][:, cols].double() - w64).abs().max().item()
    e16_64 = (ops.gemm_nt_f16x3(qa, qb)[rows][:, cols].double() - w64).abs().max().item()
    print("   fp16x3 %7.2f ms %6.1f TF/s (x%.2f vs fp32), split %.2f ms, max|d| vs fp64 %.1e" % (t16 * 1e3, flop / t16 / 1e12, t32 / t16, ts16 * 1e3, e16_64))
    print("%s | fp32 %7.2f ms %6.1f TF/s | bf16x3 %7.2f ms %6.1f TF/s (x%.2f) | bf16 %7.2f ms %6.1f TF/s | split %.2f ms | "
          "max|d| vs fp64 (unit rows): fp32 %.1e, bf16x3 %.1e; vs fp32: bf16x3 %.1e, bf16 %.1e"
          % (name, t32 * 1e3, flop / t32 / 1e12, t3 * 1e3, flop / t3 / 1e12, t32 / t3, t1 * 1e3, flop / t1 / 1e12, ts * 1e3,
             e32_64, e3_64, e3, e1), flush=True)
