"""GPU: training primitives of the transformer towers (csrc/train_bert.hip, itr_amd/autograd.py) against torch autograd run on the
CPU in float64: residual LayerNorm, gelu, short-sequence attention (mask, heads), relu + max-pool, mean over regions, dropout."""
import numpy as np
import pytest
import torch

from itr_amd import autograd as ag

pytestmark = pytest.mark.gpu


def _cmp(got, want, tol):
    assert float((got.detach().cpu().double() - want.detach()).abs().max()) <= tol


def test_add_layernorm_fwd_bwd(dev):
    torch.manual_seed(0)
    for rows, H, with_res in ((37, 64, True), (10, 768, False), (5, 200, True)):
        x, r = torch.randn(rows, H), torch.randn(rows, H)
        g, b = torch.rand(H) + 0.5, torch.randn(H) * 0.1
        dy = torch.randn(rows, H)
        xs = [t.double().requires_grad_(True) for t in (x, r, g, b)]
        z = xs[0] + (xs[1] if with_res else 0)
        u = z.mean(-1, keepdim=True)
        want = xs[2] * ((z - u) / torch.sqrt(((z - u) ** 2).mean(-1, keepdim=True) + 1e-12)) + xs[3]
        want.backward(dy.double())
        gs = [t.to(dev).requires_grad_(True) for t in (x, r, g, b)]
        out = ag.add_layernorm(gs[0], gs[1] if with_res else None, gs[2], gs[3], 1e-12)
        out.backward(dy.to(dev))
        _cmp(out, want, 5e-6)
        _cmp(gs[0].grad, xs[0].grad, 2e-5)
        if with_res:
            _cmp(gs[1].grad, xs[1].grad, 2e-5)
        _cmp(gs[2].grad, xs[2].grad, 5e-5)
        _cmp(gs[3].grad, xs[3].grad, 5e-5)


def test_gelu_fwd_bwd(dev):
    x = torch.linspace(-6, 6, 1001)
    a = x.double().requires_grad_(True)
    want = a * 0.5 * (1.0 + torch.erf(a / np.sqrt(2.0)))
    want.backward(torch.ones_like(want) * 0.7)
    gx = x.to(dev).requires_grad_(True)
    y = ag.gelu(gx)
    y.backward(torch.full_like(y, 0.7))
    _cmp(y, want, 1e-6)
    _cmp(gx.grad, a.grad, 1e-6)


@pytest.mark.parametrize("B,L,heads,dk,masked", [(3, 36, 4, 64, False), (2, 32, 12, 64, True), (4, 7, 2, 16, True), (1, 64, 1, 32, False)])
def test_mha_fwd_bwd(dev, B, L, heads, dk, masked):
    torch.manual_seed(B + L)
    A = heads * dk
    qkv = torch.randn(B * L, 3 * A) * 0.5
    dctx = torch.randn(B * L, A)
    mask = None
    if masked:
        mask = torch.ones(B, L)
        for b in range(B):
            mask[b, max(1, L - 1 - 2 * b):] = 0
    a = qkv.double().requires_grad_(True)
    q, k, v = (a[:, i * A:(i + 1) * A].view(B, L, heads, dk).permute(0, 2, 1, 3) for i in range(3))
    s = q @ k.transpose(-1, -2) / np.sqrt(dk)
    if masked:
        s = s + (1.0 - mask.double())[:, None, None, :] * -10000.0
    want = (torch.softmax(s, -1) @ v).permute(0, 2, 1, 3).reshape(B * L, A)
    want.backward(dctx.double())
    g = qkv.to(dev).requires_grad_(True)
    out = ag.mha(g, mask.to(dev) if masked else None, B, L, heads)
    out.backward(dctx.to(dev))
    _cmp(out, want, 5e-6)
    _cmp(g.grad, a.grad, 2e-5)


def test_relu_maxpool_and_mean_mid(dev):
    torch.manual_seed(3)
    x = torch.randn(5, 31, 70)
    x[0, :, 3] = -1.0                                   # a channel that never fires: output 0, no gradient
    dy = torch.randn(5, 70)
    a = x.double().requires_grad_(True)
    want = torch.relu(a).max(1).values
    want.backward(dy.double())
    g = x.to(dev).requires_grad_(True)
    out = ag.relu_maxpool(g)
    out.backward(dy.to(dev))
    _cmp(out, want, 0.0)
    _cmp(g.grad, a.grad, 0.0)
    a2 = x.double().requires_grad_(True)
    a2.mean(1).backward(dy.double())
    g2 = x.to(dev).requires_grad_(True)
    m = ag.mean_mid(g2)
    m.backward(dy.to(dev))
    _cmp(m, a2.detach().mean(1), 1e-6)
    _cmp(g2.grad, a2.grad, 1e-7)


def test_dropout_statistics_and_backward(dev):
    torch.manual_seed(5)
    seeds = ag.DropoutSeeds()
    seeds.new_step()
    x = torch.ones(400, 500, device=dev, requires_grad=True)
    y = ag.dropout(x, 0.1, seeds)
    keep = (y != 0).float().mean().item()
    assert abs(keep - 0.9) < 3e-3 and float(y.max()) == pytest.approx(1.0 / 0.9)
    y.sum().backward()
    assert torch.equal(x.grad, y.detach())                      # the same mask in the backward pass
    y2 = ag.dropout(x, 0.1, seeds)
    assert not torch.equal(y2, y)                                # the next site draws another mask
    assert ag.dropout(x, 0.0, seeds) is x and ag.dropout(x, 0.5, seeds, training=False) is x
    # attention dropout: P rows still sum to ~1 in expectation, backward consistent with a finite difference along dctx
    B, L, heads, dk = 2, 12, 2, 16
    qkv = (torch.randn(B * L, 3 * heads * dk, device=dev) * 0.3).requires_grad_(True)
    out = ag.mha(qkv, None, B, L, heads, 0.2, 1234)
    out2 = ag.mha(qkv, None, B, L, heads, 0.2, 1234)
    assert torch.equal(out, out2)
    w = torch.randn_like(out)
    (out * w).sum().backward()
    g = qkv.grad.clone()
    eps = 1e-2
    d = torch.randn_like(qkv)
    f = lambda t: float((ag.mha(t, None, B, L, heads, 0.2, 1234) * w).sum())
    fd = (f(qkv.detach() + eps * d) - f(qkv.detach() - eps * d)) / (2 * eps)
    assert fd == pytest.approx(float((g * d).sum()), rel=2e-2, abs=2e-3)
