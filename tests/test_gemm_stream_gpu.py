"""GPU: the streaming short-K GEMM (csrc/gemm_stream.hip, generated asm body) against a float64 torch reference and, bit for bit,
against the tile kernel it replaces on those shapes (same MFMA order per accumulator => identical fp32 results)."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from itr_amd import ops

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("M,N,K,act,bias", [(128 * 2048 + 37, 256, 256, None, True), (128 * 2100, 256, 256, 'relu', True),
                                            (128 * 4200, 128, 64, None, False), (128 * 1030, 384, 128, 'relu', True),
                                            (128 * 2500, 256, 512, None, True), (128 * 2048, 256, 256, 'relu', False),
                                            # eight column tiles (XCD-aware map), K = 128 (the shortest tile the kernel takes);
                                            # two or three row tiles per workgroup, K = 192 (one trip through the generic chunk loop)
                                            (128 * 600, 1024, 128, 'relu', True), (128 * 520 + 5, 256, 192, None, True),
                                            # gelu epilogue (BERT's first feed-forward GEMM, bert.py:29-34): K = 768 / the shortest K / odd chunk count
                                            (128 * 1024, 3072, 768, 'gelu', True), (128 * 2100, 256, 128, 'gelu', True), (128 * 1100 + 9, 384, 192, 'gelu', False)])
def test_stream_gemm_vs_float64(dev, M, N, K, act, bias):
    torch.manual_seed(M % 1000 + K)
    a = torch.randn(M, K, device=dev)
    b = torch.randn(N, K, device=dev) * 0.1
    bv = torch.randn(N, device=dev) if bias else None
    got = ops.linear(a, b, bv, act=act)
    assert got.shape == (M, N) and bool(torch.isfinite(got).all())
    idx = torch.randint(0, M, (6000,), device=dev)
    idx[:6] = torch.tensor([0, 127, 128, M - 1, M // 128 * 128 - 1, M // 128 * 128 - 128], device=dev)   # tile edges, the remainder rows
    want = a[idx].double() @ b.double().t()
    if bias:
        want = want + bv.double()
    if act == 'relu':
        want = want.clamp(min=0)
    if act == 'gelu':
        want = 0.5 * want * (1.0 + torch.erf(want / 2.0 ** 0.5))
    scale = float((a[idx].double().abs() @ b.double().abs().t()).max())
    assert float((got[idx].double() - want).abs().max()) <= 4e-7 * scale + (5e-7 if act == 'gelu' else 0.0)   # fp32 fmaf chain of length K (+ the gelu fit)
    # every row tile was written exactly once: no row keeps the fill value
    sentinel = ops.linear(a[:128 * 3], b, bv, act=act)                           # small M: the tile kernel
    assert torch.equal(got[:128 * 3], sentinel)


def test_stream_gemm_equals_tile_kernel_bitwise(dev):
    """tools/gemm_stream_check.py runs the same shapes with the streaming kernel on and off and prints a checksum of each result."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gemm_stream_check.py")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    blocks = r.stdout.split("ITR_GEMM_STREAM=")           # streaming (XCD-aware map) | streaming (plain map) | tile kernel
    assert len(blocks) == 4, r.stdout[-2000:]
    sums = [[ln.split("checksum")[1].strip() for ln in blk.splitlines() if "checksum" in ln] for blk in blocks[1:]]
    assert len(sums[0]) >= 5 and sums[0] == sums[1] == sums[2], (sums, r.stdout[-1500:])


def test_stream_gemm_overlapping_rows_and_strided_output(dev):
    """The call shapes the plain-shape test does not reach: rows that OVERLAP (lda < K: SAEM's Conv2d(1, C, (k, 768))
    over a token sequence is a GEMM with lda = 768, K = k * 768 -- TextEncoder.py:115-152) and an output that is a column block
    of a wider matrix (ldc > N, base not 128-byte aligned), both large enough for the streaming kernel."""
    torch.manual_seed(5)
    lda, K, N = 256, 512, 256                       # row m = base[m * 256 : m * 256 + 512]: consecutive rows share 256 floats
    M = 128 * 2100
    base = torch.randn((M - 1) * lda + K, device=dev)
    w = torch.randn(N, K, device=dev) * 0.1
    bv = torch.randn(N, device=dev)
    wide = torch.full((M, N + 36), float("nan"), device=dev)
    out = wide[:, 20:20 + N]                        # ldc = N + 36, first element 80 bytes into the row
    got = ops.linear_strided(base, lda, M, K, w, bv, act='relu', out=out)
    assert got.data_ptr() == out.data_ptr()
    assert bool(torch.isnan(wide[:, :20]).all()) and bool(torch.isnan(wide[:, 20 + N:]).all())     # nothing written outside the block
    idx = torch.randint(0, M, (4000,), device=dev)
    idx[:4] = torch.tensor([0, 127, 128, M - 1], device=dev)
    rows = torch.stack([base[i * lda:i * lda + K] for i in idx.tolist()]).double()
    want = (rows @ w.double().t() + bv.double()).clamp(min=0)
    scale = float((rows.abs() @ w.double().abs().t()).max())
    assert float((out[idx].double() - want).abs().max()) <= 4e-7 * scale
    dense = ops.linear_strided(base, lda, M, K, w, bv, act='relu')                                   # the same call, dense output
    assert torch.equal(dense, out)


def test_gemm_residual_equals_accumulate_on_a_copy(dev):
    """itr_gemm_nt_residual (Rs_GCN's `W(y) + v`, vsrn_.py:64-67): the residual is read by the epilogue from its own matrix instead
    of being copied into the output first -- the same additions in the same order, bit for bit, on a ragged shape (both tile
    kernels) and against float64."""
    torch.manual_seed(9)
    for M, N, K in ((128 * 40 + 17, 2048, 2048), (77, 96, 40)):
        x, w, b, r = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev) * 0.05, torch.randn(N, device=dev), torch.randn(M, N, device=dev)
        got = ops.gemm_residual(x, w, b, r, act='relu')
        ref = r.clone()
        ops.gemm_acc(x, K, M, K, w, b, ref, act='relu')
        assert torch.equal(got, ref)
        want = (r.double() + x.double() @ w.double().t() + b.double()).clamp(min=0)
        assert float((got.double() - want).abs().max()) <= 4e-7 * float((x.double().abs() @ w.double().abs().t()).max())
    with pytest.raises(ValueError):
        ops.gemm_residual(x, w, b, r[:, :5])
