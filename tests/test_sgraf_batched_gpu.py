"""GPU: the whole-batch SGRAF training kernels (csrc/sgraf_train.hip, itr_amd/sgraf_train.py).

* every ragged stage, forward and backward, against the same arithmetic written pair by pair with torch float64 autograd
  (the per-caption structure of EncoderSimilarity.forward, Fusionmodule.py:406-451, :632-664, :579-586, :613-618, :549-564);
* the batched step against the grouped restatement of the reference's per-caption loop (which G20 pins to the reference's own
  SGRAF.train_emb): similarity matrix, every gradient, BatchNorm running statistics -- SAF and SGR, ragged captions, C != B.
G20 itself (tests/test_sgraf_train_gpu.py) runs through the batched path."""
import numpy as np
import pytest
import torch

from itr_amd import autograd as ag
from itr_amd import sgraf_train as sgt
from itr_amd.modalmodule import Fusionmodule

pytestmark = pytest.mark.gpu


def _lens(rng, C, lo=1, hi=9):
    return sorted([int(x) for x in rng.randint(lo, hi + 1, size=C)], reverse=True)


def _close(got, want, tol, what):
    want = want.to(torch.float64)
    err = float((got.to(torch.float64) - want).abs().max())
    ref = max(1.0, float(want.abs().max()))
    assert err <= tol * ref, "%s: max error %.3g (scale %.3g)" % (what, err, ref)


@pytest.mark.parametrize("B,C,R,D", [(3, 4, 36, 32), (5, 7, 36, 1024), (2, 3, 20, 2048), (4, 2, 7, 8)])
def test_local_alignment_stages_vs_float64(dev, B, C, R, D):
    rng = np.random.RandomState(B * 100 + C)
    torch.manual_seed(B + D)
    lens = _lens(rng, C, 1, 20)
    lay = sgt.Layout(lens, dev)
    T = lay.T
    img = torch.nn.functional.normalize(torch.randn(B, R, D, device=dev), dim=-1).requires_grad_()
    words = torch.nn.functional.normalize(torch.randn(T, D, device=dev), dim=-1).requires_grad_()
    A = ag.cosine_scores(img.reshape(B * R, D), words)
    P = sgt.loc_attn(A, lay, B, R, 9.0)
    X = sgt.loc_ctx(P, img, words)
    gX = torch.randn_like(X)
    (X * gX).sum().backward()
    # float64, pair by pair (SCAN_attention + the squared difference)
    i64, w64 = img.detach().double().requires_grad_(), words.detach().double().requires_grad_()
    rows = [[None] * C for _ in range(B)]
    Pref = torch.zeros(B * T, R, dtype=torch.float64, device=dev)
    off = np.concatenate([[0], np.cumsum(lens)])
    for c in range(C):
        cap = w64[off[c]:off[c + 1]]
        for b in range(B):
            attn = torch.nn.functional.leaky_relu(i64[b] @ cap.t(), 0.1)                       # (R, W)
            attn = attn / (attn.pow(2).sum(1, keepdim=True).sqrt() + 1e-8)
            p = torch.softmax(attn.t() * 9.0, dim=1)                                           # (W, R)
            Pref[b * T + off[c]:b * T + off[c + 1]] = p.detach()
            ctx = p @ i64[b]
            ctx = ctx / (ctx.pow(2).sum(1, keepdim=True).sqrt() + 1e-8)
            rows[b][c] = (ctx - cap) ** 2
    Xref = torch.cat([torch.cat(r, 0) for r in rows], 0)
    (Xref * gX.double()).sum().backward()
    _close(P.detach(), Pref, 2e-5, "P")
    _close(X.detach(), Xref.detach(), 2e-5, "X")
    _close(img.grad, i64.grad, 2e-4, "d img")
    _close(words.grad, w64.grad, 2e-4, "d words")


@pytest.mark.parametrize("B,C,S", [(3, 5, 16), (4, 6, 256), (2, 2, 20)])
def test_graph_and_filtration_stages_vs_float64(dev, B, C, S):
    rng = np.random.RandomState(S + C)
    torch.manual_seed(S)
    lens = _lens(rng, C, 1, 22)
    lay = sgt.Layout(lens, dev)
    off = np.concatenate([[0], np.cumsum(lens)])
    glo = torch.randn(B * C, S, device=dev, requires_grad=True)
    loc = torch.randn(B * lay.T, S, device=dev, requires_grad=True)
    nodes = sgt.assemble_nodes(glo, loc, lay, B)
    q = (torch.randn(B * lay.NT, S, device=dev) * 0.3).requires_grad_()
    k = (torch.randn(B * lay.NT, S, device=dev) * 0.3).requires_grad_()
    Z = sgt.graph_attn(q, k, nodes, lay, B)
    bn = torch.nn.BatchNorm1d(1).to(dev)
    bn.weight.data.fill_(1.3)
    bn.bias.data.fill_(-0.2)
    bn.running_mean.fill_(0.05)
    bn.running_var.fill_(0.8)
    wa = (torch.randn(S, device=dev) * 0.2).requires_grad_()
    a = (Z * wa).sum(1)
    y = sgt.seg_bn_train(a, bn, lay, B)
    pooled = sgt.saf_pool(y, Z, lay, B)
    row0 = ag.gather_rows(Z, lay.node0_rows(B))
    g1, g2 = torch.randn_like(pooled), torch.randn_like(row0)
    ((pooled * g1).sum() + (row0 * g2).sum()).backward()

    d = lambda t: t.detach().double().requires_grad_()
    glo64, loc64, q64, k64, wa64 = d(glo), d(loc), d(q), d(k), d(wa)
    gam, bet = d(bn.weight), d(bn.bias)
    rm, rv = torch.tensor([0.05], dtype=torch.float64, device=dev), torch.tensor([0.8], dtype=torch.float64, device=dev)
    pooled_ref = [[None] * C for _ in range(B)]
    row0_ref = [[None] * C for _ in range(B)]
    Zs = {}
    for c in range(C):                                   # the reference's order: caption by caption, all images at once
        n = lens[c] + 1
        logits = []
        for b in range(B):
            x = torch.cat([glo64[b * C + c:b * C + c + 1], loc64[b * lay.T + off[c]:b * lay.T + off[c + 1]]], 0)       # (n, S)
            r0 = b * lay.NT + off[c] + c
            E = torch.softmax(q64[r0:r0 + n] @ k64[r0:r0 + n].t(), dim=-1)
            Zs[b] = E @ x
            row0_ref[b][c] = Zs[b][0]
            logits.append(Zs[b] @ wa64)
        lg = torch.stack(logits, 0)                      # (B, n): BatchNorm1d(1) over B * n values
        mu, var = lg.mean(), lg.var(unbiased=False)
        Nn = B * n
        rm = 0.9 * rm + 0.1 * mu.detach()
        rv = 0.9 * rv + 0.1 * var.detach() * Nn / max(Nn - 1, 1)
        yy = torch.sigmoid(gam * (lg - mu) / torch.sqrt(var + bn.eps) + bet)
        wgt = yy / (yy.abs().sum(1, keepdim=True) + 1e-8)
        for b in range(B):
            pooled_ref[b][c] = wgt[b] @ Zs[b]
    pr = torch.stack([torch.stack(r, 0) for r in pooled_ref], 0).reshape(B * C, S)
    rr = torch.stack([torch.stack(r, 0) for r in row0_ref], 0).reshape(B * C, S)
    ((pr * g1.double()).sum() + (rr * g2.double()).sum()).backward()
    _close(pooled.detach(), pr.detach(), 3e-5, "pooled")
    _close(row0.detach(), rr.detach(), 3e-5, "node 0")
    for name, got, want in (("d glo", glo.grad, glo64.grad), ("d loc", loc.grad, loc64.grad), ("d q", q.grad, q64.grad), ("d k", k.grad, k64.grad),
                            ("d w_attn", wa.grad, wa64.grad), ("d gamma", bn.weight.grad, gam.grad), ("d beta", bn.bias.grad, bet.grad)):
        _close(got, want, 3e-4, name)
    _close(bn.running_mean, rm, 1e-5, "running_mean")
    _close(bn.running_var, rv, 1e-5, "running_var")
    assert int(bn.num_batches_tracked) == C


@pytest.mark.parametrize("B,C,S", [(3, 5, 16), (4, 6, 256)])
def test_last_reasoning_step_asks_only_node_zero(dev, B, C, S):
    """row0 form (the last GraphReasoning step, read at [:, 0, :]) == node 0 of the full form: values and every gradient."""
    rng = np.random.RandomState(S)
    torch.manual_seed(S + 1)
    lens = _lens(rng, C, 1, 22)
    lay = sgt.Layout(lens, dev)
    rows0 = lay.node0_rows(B)
    res = []
    for row0 in (False, True):
        torch.manual_seed(4)
        x = torch.randn(B * lay.NT, S, device=dev, requires_grad=True)
        q = (torch.randn(B * lay.NT, S, device=dev) * 0.3).requires_grad_()
        k = (torch.randn(B * lay.NT, S, device=dev) * 0.3).requires_grad_()
        if row0:
            z0 = sgt.graph_attn(ag.gather_rows(q, rows0), k, x, lay, B, row0=True)
        else:
            z0 = ag.gather_rows(sgt.graph_attn(q, k, x, lay, B), rows0)
        g = torch.randn(B * C, S, device=dev, generator=torch.Generator(device=dev).manual_seed(8))
        (z0 * g).sum().backward()
        res.append((z0.detach(), x.grad, q.grad, k.grad))
    for name, a, b in zip(("Z0", "d x", "d q", "d k"), res[0], res[1]):
        _close(b, a, 1e-5, name)


@pytest.mark.parametrize("C,D", [(5, 32), (9, 1024), (1, 12)])
def test_text_self_attention_stages_vs_float64(dev, C, D):
    rng = np.random.RandomState(C)
    torch.manual_seed(C + D)
    lens = _lens(rng, C, 1, 30)
    lay = sgt.Layout(lens, dev)
    off = np.concatenate([[0], np.cumsum(lens)])
    words = torch.randn(lay.T, D, device=dev, requires_grad=True)
    capv = torch.randn(C, D, device=dev, requires_grad=True)
    logit = torch.randn(lay.T, device=dev, requires_grad=True)
    ig = torch.randn(4, D, device=dev, requires_grad=True)
    m = sgt.seg_mean(words, lay)
    sp = sgt.seg_spread(capv, lay)
    sm = sgt.seg_smry(logit, words, lay)
    sq = sgt.pair_sqdiff(ig, sm)
    g = [torch.randn_like(t) for t in (m, sp, sm, sq)]
    sum((t * gi).sum() for t, gi in zip((m, sp, sm, sq), g)).backward()
    d = lambda t: t.detach().double().requires_grad_()
    w64, c64, l64, i64 = d(words), d(capv), d(logit), d(ig)
    m_r = torch.stack([w64[off[c]:off[c + 1]].mean(0) for c in range(C)])
    sp_r = torch.cat([c64[c:c + 1].expand(lens[c], D) for c in range(C)])
    sm_r = torch.stack([torch.softmax(l64[off[c]:off[c + 1]], 0) @ w64[off[c]:off[c + 1]] for c in range(C)])
    sq_r = ((i64.unsqueeze(1) - sm_r.unsqueeze(0)) ** 2).reshape(4 * C, D)
    sum((t * gi.double()).sum() for t, gi in zip((m_r, sp_r, sm_r, sq_r), g)).backward()
    for name, got, want in (("mean", m, m_r), ("spread", sp, sp_r), ("smry", sm, sm_r), ("sqdiff", sq, sq_r)):
        _close(got.detach(), want.detach(), 1e-5, name)
    for name, got, want in (("d words", words.grad, w64.grad), ("d cap", capv.grad, c64.grad), ("d logit", logit.grad, l64.grad), ("d img_glo", ig.grad, i64.grad)):
        _close(got, want, 1e-4, name)


def _sim_enc(dev, D, S, mod, seed):
    torch.manual_seed(seed)
    m = Fusionmodule.EncoderSimilarity(D, S, mod, 3).to(dev)
    with torch.no_grad():
        for name, p in m.named_parameters():           # away from the all-zero biases / unit BatchNorm weights of the initialisation
            if p.dim() == 1:
                p.add_(torch.randn_like(p) * 0.1)
    for mm in m.modules():
        if isinstance(mm, torch.nn.Dropout):
            mm.p = 0.0
    m.train()
    return m


@pytest.mark.parametrize("mod,B,C,D,S,R", [("SAF", 6, 6, 32, 16, 36), ("SGR", 6, 6, 32, 16, 36), ("SAF", 5, 3, 64, 32, 36), ("SGR", 4, 7, 128, 64, 36),
                                            ("SGR", 8, 8, 1024, 256, 36), ("SAF", 8, 8, 1024, 256, 36)])
def test_batched_step_equals_the_per_caption_loop(dev, mod, B, C, D, S, R):
    rng = np.random.RandomState(B + C + D)
    lens = _lens(rng, C, 2, 14)
    off = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.int64)
    out = {}
    for batched in (False, True):
        m = _sim_enc(dev, D, S, mod, 5)
        torch.manual_seed(9)
        img = torch.nn.functional.normalize(torch.randn(B, R, D, device=dev), dim=-1).requires_grad_()
        words = torch.nn.functional.normalize(torch.randn(int(np.sum(lens)), D, device=dev), dim=-1).requires_grad_()
        seeds = ag.DropoutSeeds()
        sims = Fusionmodule.encoder_similarity_train(m, img, words, off, lens, seeds, True, batched=batched)
        assert sims.shape == (B, C)
        gs = torch.randn(B, C, device=dev, generator=torch.Generator(device=dev).manual_seed(3))
        (sims * gs).sum().backward()
        out[batched] = (sims.detach(), img.grad, words.grad, {n: p.grad for n, p in m.named_parameters()},
                        {n: b.clone() for n, b in m.named_buffers()})
    a, b = out[False], out[True]
    _close(b[0], a[0], 2e-5, "sims")
    _close(b[1], a[1], 3e-4, "d img")
    _close(b[2], a[2], 3e-4, "d words")
    for n in a[3]:
        assert (a[3][n] is None) == (b[3][n] is None), n
        if a[3][n] is not None:
            scale = max(float(a[3][n].abs().max()), 1e-3)
            assert float((a[3][n] - b[3][n]).abs().max()) <= 5e-4 * scale + 1e-6, (n, float((a[3][n] - b[3][n]).abs().max()), scale)
    for n in a[4]:
        if a[4][n].is_floating_point():
            _close(b[4][n], a[4][n], 1e-4, n)
        else:
            assert int(a[4][n]) == int(b[4][n]), n


def test_long_captions_take_the_grouped_path(dev):
    m = _sim_enc(dev, 32, 16, "SAF", 1)
    lens = [120, 5]
    img = torch.randn(2, 36, 32, device=dev)
    words = torch.randn(125, 32, device=dev)
    assert not sgt.supported(32, 36, 16, lens)
    with pytest.raises(NotImplementedError):
        Fusionmodule.encoder_similarity_train(m, img, words, [0, 120], lens, ag.DropoutSeeds(), True, batched=True)
    sims = Fusionmodule.encoder_similarity_train(m, img, words, [0, 120], lens, ag.DropoutSeeds(), True)
    assert sims.shape == (2, 2) and bool(torch.isfinite(sims).all())
