"""GPU: the weight-gradient GEMM dW = dY^T X with a split row reduction (csrc/gemm_tn.hip) against a float64 torch reference:
all three tile sizes, ragged edges (unaligned scalar path), one slice / many slices, accumulate, strided operands, and through
autograd.linear's backward (the caller: every dense layer of model.train_emb, Models.py:139-144)."""
import pytest
import torch

from itr_amd import autograd as ag

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("R,P,Q", [(295000, 32, 32),      # CAMERA's gate layers: one workgroup for 15.8 ms on the NT kernel
                                   (213000, 256, 1024),   # SGRAF sim_tranloc_w at batch 128
                                   (4608, 2048, 2048), (229376, 256, 256), (1664, 1024, 1024),
                                   (5000, 1, 256), (300, 12, 1024), (777, 300, 6), (129, 70, 50), (16, 64, 64), (1, 33, 17), (0, 8, 8)])
def test_gemm_tn_vs_float64(dev, R, P, Q):
    torch.manual_seed(R % 997 + P + Q)
    a = torch.randn(R, P, device=dev)
    b = torch.randn(R, Q, device=dev)
    got = ag._gemm_tn(a, b)
    want = a.double().t() @ b.double()
    scale = max(1.0, float(R) ** 0.5)
    assert got.shape == (P, Q)
    assert float((got.double() - want).abs().max()) <= 2e-5 * scale + 1e-6 * R ** 0.5
    # accumulate onto an existing matrix; bit-identical partial sums (deterministic slice order)
    base = torch.randn(P, Q, device=dev)
    acc = ag._gemm_tn(a, b, out=base.clone(), accumulate=True)
    assert float((acc.double() - base.double() - want).abs().max()) <= 2e-5 * scale + 1e-6 * R ** 0.5 + 1e-6
    assert torch.equal(ag._gemm_tn(a, b), got)


def test_gemm_tn_strided_operands(dev):
    torch.manual_seed(3)
    big_a = torch.randn(3000, 200, device=dev)
    big_b = torch.randn(3000, 400, device=dev)
    a, b = big_a[:, 8:136], big_b[:, 100:356]                  # row stride > width, 16-byte aligned start
    got = ag._gemm_tn(a, b)
    assert float((got.double() - a.double().t() @ b.double()).abs().max()) <= 2e-3
    a2, b2 = big_a[:, 3:70], big_b[:, 1:98]                    # unaligned start and ragged widths: the scalar path
    got2 = ag._gemm_tn(a2, b2)
    assert float((got2.double() - a2.double().t() @ b2.double()).abs().max()) <= 2e-3


@pytest.mark.parametrize("rows,K,N,bias", [(70000, 32, 32, True), (5000, 256, 1, True), (999, 300, 1024, False)])
def test_linear_backward_uses_the_split_reduction(dev, rows, K, N, bias):
    torch.manual_seed(rows + K)
    x = torch.randn(rows, K, device=dev, requires_grad=True)
    w = (torch.randn(N, K, device=dev) * 0.1).requires_grad_()
    bv = torch.randn(N, device=dev).requires_grad_() if bias else None
    y = ag.linear(x, w, bv)
    g = torch.randn_like(y)
    y.backward(g)
    xd, wd = x.detach().double().requires_grad_(), w.detach().double().requires_grad_()
    bd = bv.detach().double().requires_grad_() if bias else None
    yd = torch.nn.functional.linear(xd, wd, bd)
    yd.backward(g.double())
    assert float((y.detach().double() - yd.detach()).abs().max()) <= 1e-4
    assert float((w.grad.double() - wd.grad).abs().max()) <= 3e-5 * rows ** 0.5
    assert float((x.grad.double() - xd.grad).abs().max()) <= 1e-4
    if bias:
        assert float((bv.grad.double() - bd.grad).abs().max()) <= 3e-5 * rows ** 0.5


@pytest.mark.parametrize("M,N,K,bias,act", [(128, 1536, 512, True, None), (128, 512, 2560, True, 'tanh'), (37, 3072, 1024, False, None),
                                            (1, 64, 128, True, 'relu'), (128, 1000, 300, True, None), (100, 96, 132, False, 'sigmoid')])
def test_skinny_gemm_vs_float64(dev, M, N, K, bias, act):
    """itr_gemm_nt_algo(algo = 4): the <= 128-row products of the training tape (decoder steps, per-caption vector layers) on 16-column
    strips (csrc/gemm_skinny.hip), against float64 -- ragged N (not a multiple of 16), K not a multiple of the 64-wide chunk, M = 1."""
    from itr_amd import ops
    torch.manual_seed(M + N)
    a = torch.randn(M, K, device=dev)
    w = torch.randn(N, K, device=dev) * 0.1
    b = torch.randn(N, device=dev) if bias else None
    got = ops.linear(a, w, b, act=act, algo="skinny")
    want = a.double() @ w.double().t()
    if bias:
        want = want + b.double()
    want = {None: lambda t: t, 'tanh': torch.tanh, 'relu': torch.relu, 'sigmoid': torch.sigmoid}[act](want)
    assert float((got.double() - want).abs().max()) <= 2e-5 * K ** 0.5
    x = a.clone().requires_grad_()
    y = ag.linear(x, w.clone().requires_grad_(), b)                  # the tape routes M <= 128 through it
    y.sum().backward()
    assert float((y.detach().double() - (a.double() @ w.double().t() + (b.double() if bias else 0))).abs().max()) <= 2e-5 * K ** 0.5
    assert float((x.grad.double() - w.double().sum(0)).abs().max()) <= 2e-4 * N ** 0.5


@pytest.mark.parametrize("M,N,K", [(4608, 128, 10240), (4608, 256, 2048), (300, 128, 6144), (4608, 1024, 2048)])
def test_linear_splits_k_when_the_output_has_few_tiles(dev, M, N, K):
    """itr_gemm_nt_splitk through autograd.linear (CAMERA's dilated convolutions as GEMMs: 4 608 x 128 outputs over K = 6 144 / 10 240);
    the last shape has enough tiles and takes the plain kernel."""
    torch.manual_seed(N)
    x = torch.randn(M, K, device=dev)
    w = torch.randn(N, K, device=dev) * 0.05
    b = torch.randn(N, device=dev)
    y = ag.linear(x, w, b)
    idx = torch.randint(0, M, (257,), device=dev)
    want = x[idx].double() @ w.double().t() + b.double()
    assert float((y[idx].double() - want).abs().max()) <= 2e-5 * K ** 0.5
    assert torch.equal(y, ag.linear(x, w, b))            # slices are added in a fixed order


@pytest.mark.parametrize("M,N,K,bias", [(4608, 2048, 2048, True),       # 576 tiles: the last 4 row tiles peeled (VSRN's graph-convolution layers)
                                        (2048, 2304, 768, True),        # 288 tiles (BERT's fused Q/K/V at batch 64 x 32 tokens)
                                        (4700, 2000, 1024, False),      # ragged rows and columns, 592 tiles
                                        (2048, 3072, 768, True),        # 384 tiles: no plan pays, one launch
                                        (4096, 3000, 768, True)])       # 768 tiles: the last 8 COLUMN tiles peeled (ragged: 3 000 columns), BERT's FFN at 4 096 rows
def test_linear_peels_a_round_and_a_bit_of_tiles(dev, M, N, K, bias):
    """autograd._peel_plan: the main launch and the K-sliced tail together are the product (float64 reference on sampled rows that
    cover both parts), forward and input gradient, and the same bits on every call."""
    torch.manual_seed(M + N)
    x = torch.randn(M, K, device=dev, requires_grad=True)
    w = (torch.randn(N, K, device=dev) * 0.05).requires_grad_()
    b = torch.randn(N, device=dev).requires_grad_() if bias else None
    y = ag.linear(x, w, b)
    idx = torch.cat([torch.randint(0, M, (200,), device=dev), torch.arange(M - 130, M, device=dev), torch.arange(0, 3, device=dev)])
    want = x.detach()[idx].double() @ w.detach().double().t() + (b.detach().double() if bias else 0)
    assert float((y.detach()[idx].double() - want).abs().max()) <= 2e-5 * K ** 0.5
    assert torch.equal(y, ag.linear(x, w, b))
    g = torch.randn_like(y)
    y.backward(g)
    want_dx = g[idx].double() @ w.detach().double()
    assert float((x.grad[idx].double() - want_dx).abs().max()) <= 2e-5 * N ** 0.5


@pytest.mark.parametrize("M,N,K,bias", [(128, 512, 512, True), (128, 1536, 1024, True), (128, 812, 1536, False), (5, 64, 256, True), (128, 512, 300, False)])
def test_small_batch_layers_take_k_slices_and_one_sum(dev, M, N, K, bias):
    """autograd.linear for <= 128 rows with 64 <= N < 2048, K >= 256: itr_gemm_nt_splitk's skinny route (16-column strips x K slices + the
    slice sum with bias), against float64; the same bits on every call; ragged N and a K that is not a multiple of the 64-wide chunk."""
    torch.manual_seed(N + K)
    x = torch.randn(M, K, device=dev)
    w = torch.randn(N, K, device=dev) * 0.05
    b = torch.randn(N, device=dev) if bias else None
    y = ag.linear(x, w, b)
    want = x.double() @ w.double().t() + (b.double() if bias else 0)
    assert float((y.double() - want).abs().max()) <= 2e-5 * K ** 0.5
    assert torch.equal(y, ag.linear(x, w, b))


def test_linear_backward_with_an_output_width_that_is_not_a_multiple_of_four(dev):
    """_Linear.backward's transposed route (the vocabulary projection of the caption decoder, Fusionmodule.py:277-292: 9 487 words):
    dx, dW and db against float64 autograd."""
    torch.manual_seed(11)
    M, K, N = 1024, 512, 1303
    x = torch.randn(M, K, device=dev, requires_grad=True)
    w = (torch.randn(N, K, device=dev) * 0.05).requires_grad_()
    b = torch.randn(N, device=dev).requires_grad_()
    y = ag.linear(x, w, b)
    g = torch.randn_like(y)
    y.backward(g)
    xd, wd, bd = (t.detach().double().requires_grad_() for t in (x, w, b))
    torch.nn.functional.linear(xd, wd, bd).backward(g.double())
    assert float((x.grad.double() - xd.grad).abs().max()) <= 2e-5 * N ** 0.5
    assert float((w.grad.double() - wd.grad).abs().max()) <= 3e-5 * M ** 0.5
    assert float((b.grad.double() - bd.grad).abs().max()) <= 3e-5 * M ** 0.5
