#!/usr/bin/env python3
"""Streaming GEMM (csrc/gemm_stream.hip) against the tile kernel and against float64, plus timing (run on the GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "image-text-retrieval_amd"))
import subprocess
import torch
if len(sys.argv) > 1 and sys.argv[1] == "child":
    from itr_amd import ops
    ALGO = sys.argv[2]
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    for (M, N, K, act) in [(128 * 2048 + 37, 256, 256, None), (128 * 2100, 256, 256, 'relu'), (128 * 4200, 128, 128, None), (128 * 40000, 256, 256, 'relu'), (265000, 256, 256, 'relu'),
                           (128 * 1030, 384, 128, None), (128 * 2500, 256, 512, 'relu')] + [(128 * 1400, 1024, 2048, None), (128 * 6250, 2304, 768, None),
                           (128 * 6250, 768, 3072, None), (128 * 2070, 256, 1024, None), (128 * 200, 3072, 1024, None),
                           (128 * 1024, 3072, 768, 'gelu'), (128 * 2100, 256, 256, 'gelu'), (128 * 1030 + 5, 384, 128, 'gelu')]:      # BERT's first feed-forward GEMM; short K
        a = torch.randn(M, K, device=dev); b = torch.randn(N, K, device=dev) * 0.1; bias = torch.randn(N, device=dev)
        got = ops.linear(a, b, bias, act=act, algo=ALGO)
        idx = torch.randint(0, M, (4000,), device=dev)
        idx[:3] = torch.tensor([0, M - 1, M // 128 * 128 - 1], device=dev)
        want = a[idx].double() @ b.double().t() + bias.double()
        if act == 'relu':
            want = want.clamp(min=0)
        if act == 'gelu':
            want = 0.5 * want * (1.0 + torch.erf(want / 2.0 ** 0.5))
        err = float((got[idx].double() - want).abs().max())
        for _ in range(2): ops.linear(a, b, bias, act=act, algo=ALGO)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): ops.linear(a, b, bias, act=act, algo=ALGO)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        print("  M=%7d N=%4d K=%4d act=%-5s %8.3f ms  %6.1f TF/s   max|d| vs fp64 %.2e   checksum %.6f" % (
            M, N, K, act, ms, 2.0 * M * N * K / ms / 1e9, err, float(got.double().sum())))
else:
    # the kernel is an explicit argument of the ABI (itr_gemm_nt_algo), not an environment switch
    for algo in ("stream_xcd", "stream_plain", "tile"):
        print("ITR_GEMM_STREAM=%s" % algo)
        sys.stdout.flush()
        subprocess.run([sys.executable, os.path.abspath(__file__), "child", algo])
