"""GPU: the split-bf16 GEMM study kernel (csrc/gemm_bf16x3.hip; opt-in, not on any default path): fragment / C-layout
correctness with asymmetric operands, ragged M / N, bias, and the error bounds that STUDY_SPLIT_PRECISION.md quotes."""
import numpy as np
import pytest
import torch

from itr_amd import ops

pytestmark = pytest.mark.gpu


def test_split_bf16_planes(dev):
    torch.manual_seed(0)
    x = torch.randn(5, 64, device=dev) * torch.logspace(-3, 3, 64, device=dev)
    il = ops.split_bf16(x)                                            # [5, 128]: per 32-chunk 32 hi then 32 lo
    assert il.shape == (5, 128) and il.dtype == torch.int16
    f = lambda p: (p.to(torch.int32) << 16).view(torch.float32)      # bf16 bits -> fp32
    v = il.view(5, 2, 2, 32)                                          # [row][chunk][hi | lo][32]
    hi, lo = f(v[:, :, 0]).reshape(5, 64), f(v[:, :, 1]).reshape(5, 64)
    assert torch.equal(hi, x.to(torch.bfloat16).float())             # round to nearest even, like torch
    assert ((x - hi - lo).abs() <= x.abs() * 2.0 ** -16).all()
    with pytest.raises(Exception):
        ops.split_bf16(torch.randn(4, 48, device=dev))               # K not a multiple of 32


@pytest.mark.parametrize("M,N,K", [(128, 128, 32), (200, 77, 64), (1, 1, 32), (300, 513, 1024)])
def test_gemm_bf16x3_vs_fp64(dev, M, N, K):
    torch.manual_seed(M + N)
    a = torch.randn(M, K) * (1.0 + torch.arange(M).float()[:, None] / M)        # asymmetric: a row / column swap fails
    b = torch.randn(N, K) * (0.5 + torch.arange(N).float()[:, None] / N)
    bias = torch.randn(N)
    want = a.double() @ b.double().t() + bias.double()
    scale = (a.abs().double() @ b.abs().double().t())                            # sum_k |a_k b_k|: the natural error scale
    pa, pb = ops.split_bf16(a.to(dev)), ops.split_bf16(b.to(dev))
    got3 = ops.gemm_nt_bf16(pa, pb, bias.to(dev), terms=3).cpu().double()
    got1 = ops.gemm_nt_bf16(pa, pb, bias.to(dev), terms=1).cpu().double()
    assert ((got3 - want).abs() / scale).max().item() <= 3e-5                   # worst case 3 * 2^-17 per product
    assert ((got3 - want).abs() / scale).mean().item() <= 1e-6
    assert ((got1 - want).abs() / scale).max().item() <= 8e-3                   # plain bf16: 2^-8 per product
    fp32 = ops.linear(a.to(dev), b.to(dev), bias.to(dev)).cpu().double()
    assert (got3 - fp32).abs().max().item() <= 2e-4 * max(1.0, want.abs().max().item())


def test_gemm_bf16_argument_errors(dev):
    pa = ops.split_bf16(torch.randn(4, 32, device=dev))
    pb = ops.split_bf16(torch.randn(6, 64, device=dev))
    with pytest.raises(ValueError):
        ops.gemm_nt_bf16(pa, pb)                                   # K mismatch
    with pytest.raises(Exception):
        ops.gemm_nt_bf16(pa, pa, terms=2)
    assert ops.gemm_nt_bf16(pa, pa, terms=1).shape == (4, 4)


@pytest.mark.parametrize("act", [None, 'relu', 'tanh', 'sigmoid', 'gelu', 'leaky_relu'])
def test_gemm_bf16x3_activation_epilogue(dev, act):
    torch.manual_seed(3)
    a, b, bias = torch.randn(70, 96, device=dev), torch.randn(45, 96, device=dev), torch.randn(45, device=dev)
    got = ops.gemm_nt_bf16(ops.split_bf16(a), ops.split_bf16(b), bias, terms=3, act=act)
    want = ops.linear(a, b, bias, act=act)
    scale = (a.abs() @ b.abs().t()).max().item()               # sum_k |a_k b_k|; activations here are 1-Lipschitz or flatter
    assert (got - want).abs().max().item() <= 4e-6 * scale


def test_gemm_bf16x3_overlapping_rows_and_env_switch(dev):
    """lda < K (SAEM's Conv2d over consecutive tokens as a GEMM) and the ITR_GEMM_BF16X3 routing of ops.linear /
    linear_strided / cosine_scores."""
    torch.manual_seed(4)
    base = torch.randn(40, 64, device=dev)                      # 40 tokens x 64
    w = torch.randn(24, 3 * 64, device=dev)
    want = ops.linear_strided(base, 64, 38, 192, w, None, act='relu')
    ops.BF16X3 = True
    try:
        got = ops.linear_strided(base, 64, 38, 192, w, None, act='relu')
        x = torch.randn(5, 7, 64, device=dev)
        w2, b2 = torch.randn(33, 64, device=dev), torch.randn(33, device=dev)
        y = ops.linear(x, w2, b2, act='gelu')
        s = ops.cosine_scores(base, base[:9].contiguous())
        w2.mul_(2.0)                                            # in-place update: the planes are rebuilt on every call
        y2 = ops.linear(x, w2, b2)
    finally:
        ops.BF16X3 = False
    tol = 4e-6 * 3 * 64 * 4.0        # 4e-6 of sum_k |a_k b_k| (K = 192 at most, |a_k b_k| rarely above 4)
    assert (got - want).abs().max().item() <= tol
    assert (y - ops.linear(x, w2 / 2.0, b2, act='gelu')).abs().max().item() <= tol and y.shape == (5, 7, 33)
    assert (s - ops.cosine_scores(base, base[:9].contiguous())).abs().max().item() <= tol
    assert (y2 - ops.linear(x, w2, b2)).abs().max().item() <= 2 * tol


@pytest.mark.parametrize("M,N,K,scale", [(200, 77, 64, 1.0), (300, 513, 1024, 1.0), (130, 140, 96, 3000.0), (64, 64, 32, 1e-6)])
def test_gemm_f16x3_fp32_level_accuracy(dev, M, N, K, scale):
    """fp16 planes with absmax scaling: error at the fp32 rounding level for operands of any magnitude (no fp16 overflow, no flushed
    subnormals), compared with the bf16x3 bound of 4e-6 and with the exact fp32 GEMM."""
    torch.manual_seed(M + K)
    a = torch.randn(M, K) * scale * (1.0 + torch.arange(M).float()[:, None] / M)
    b = torch.randn(N, K) * (0.5 + torch.arange(N).float()[:, None] / N)
    a[0, :5] = torch.tensor([1e-12, -3e-9, 7e-7, 0.0, 2e-15]) * scale           # tiny components next to large ones
    bias = torch.randn(N) * scale
    want = a.double() @ b.double().t() + bias.double()
    nat = (a.abs().double() @ b.abs().double().t())
    got = ops.gemm_nt_f16x3(ops.split_f16(a.to(dev)), ops.split_f16(b.to(dev)), bias.to(dev)).cpu().double()
    fp32 = ops.linear(a.to(dev), b.to(dev), bias.to(dev)).cpu().double()
    e16, e32 = ((got - want).abs() / nat).max().item(), ((fp32 - want).abs() / nat).max().item()
    assert bool(torch.isfinite(got).all())
    assert e16 <= 1e-6 and e16 <= 8 * e32 + 3e-7, (e16, e32)       # (bf16x3: 4e-6)
    act = ops.gemm_nt_f16x3(ops.split_f16(a.to(dev)), ops.split_f16(b.to(dev)), bias.to(dev), act='tanh').cpu().double()
    assert (act - torch.tanh(want)).abs().max().item() <= 1e-6 * max(1.0, nat.max().item())      # tanh is 1-Lipschitz


def test_fp16x3_env_switch(dev):
    torch.manual_seed(8)
    x, w, b = torch.randn(5, 7, 64, device=dev) * 30, torch.randn(33, 64, device=dev) * 0.05, torch.randn(33, device=dev)
    want = ops.linear(x, w, b, act='gelu')
    ops.FP16X3 = True
    try:
        got = ops.linear(x, w, b, act='gelu')
        s = ops.cosine_scores(w, w[:9].contiguous())
    finally:
        ops.FP16X3 = False
    assert (got - want).abs().max().item() <= 2e-5 and (s - ops.cosine_scores(w, w[:9].contiguous())).abs().max().item() <= 1e-7
