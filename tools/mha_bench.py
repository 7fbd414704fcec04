#!/usr/bin/env python3
"""Short-sequence attention kernel (csrc/transformer.hip mha_reg_kernel / mha_mfma_kernel) at BERT-base shapes of the SAEM / CAMERA text tower:
B sequences x L positions x 12 heads x 64, q / k / v column slices of one fused QKV buffer.  Prints time, effective bandwidth
(q + k + v read, context written) and a checksum; the result is checked against a float64 torch reference on the first sequences.

    python3 tools/mha_bench.py [B] [L]            (run on the GPU box; ITR_MHA_LDS=1: the LDS-staged kernel instead of the register-only one)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "image-text-retrieval_amd"))
import torch
from itr_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 3571
Ls = [int(sys.argv[2])] if len(sys.argv) > 2 else [12, 30, 45, 64]
heads, dk = 12, 64
H = heads * dk
dev = torch.device("cuda:0")
for L in Ls:
    torch.manual_seed(L)
    qkv = torch.randn(B * L, 3 * H, device=dev)
    mask = (torch.arange(L, device=dev)[None, :] < torch.randint(max(1, L // 2), L + 1, (B, 1), device=dev)).float()
    q, k, v = qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:]
    out = ops.mha_small(q, k, v, mask, B, L, heads, dk, 0.125)
    nb = 8
    qq, kk, vv = (t[:nb * L].double().view(nb, L, heads, dk).transpose(1, 2) for t in (q, k, v))
    sc = qq @ kk.transpose(-1, -2) * 0.125 + ((1.0 - mask[:nb].double()) * -10000.0)[:, None, None, :]
    want = (torch.softmax(sc, -1) @ vv).transpose(1, 2).reshape(nb * L, H)
    err = float((out[:nb * L].double() - want).abs().max())
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    for _ in range(3):
        ops.mha_small(q, k, v, mask, B, L, heads, dk, 0.125)
    ev[0].record()
    n = 20
    for _ in range(n):
        ops.mha_small(q, k, v, mask, B, L, heads, dk, 0.125)
    ev[1].record()
    torch.cuda.synchronize()
    us = ev[0].elapsed_time(ev[1]) * 1e3 / n
    gb = 4 * B * L * H * 4 / 1e9
    print("B %d L %2d  %8.1f us  %6.2f TB/s  max|err| %.2e  checksum %.9e" % (B, L, us, gb / us * 1e3, err, float(out.double().sum())))
