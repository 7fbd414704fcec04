#!/usr/bin/env python3
"""Randomised parity sweep over the kernels of round 3 (run on the GPU box; a few minutes): short-sequence attention, CAMERA's gate,
SCAN with any region count, the i2t epilogue on random caption-length mixes (tiles with and without far Gram blocks), partition
invariance of t2i.  Prints one line per family; exits non-zero on the first violation.

    python3 tools/fuzz_round3.py [seconds per family]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "image-text-retrieval_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import torch
import itr_oracle as O
from itr_amd import ops

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 40.0
dev = torch.device("cuda:0")
rng = np.random.RandomState(12345)


def family(name, fn):
    t0, n, worst = time.time(), 0, 0.0
    while time.time() - t0 < budget:
        worst = max(worst, fn())
        n += 1
    print("%-28s %4d cases, worst ratio to tolerance %.3f" % (name, n, worst))
    if worst > 1.0:
        sys.exit("FAILED: " + name)


def mha():
    B, L, heads, dk = int(rng.randint(1, 40)), int(rng.randint(1, 65)), int(rng.randint(1, 13)), int(rng.choice([16, 32, 64]))
    H = heads * dk
    qkv = torch.randn(B * L, 3 * H)
    mask = (torch.arange(L)[None, :] < torch.from_numpy(rng.randint(1, L + 1, size=(B, 1)))).float()
    use_mask = rng.rand() < 0.7
    scale = 1.0 / dk ** 0.5
    q, k, v = (qkv[:, i * H:(i + 1) * H].double().reshape(B, L, heads, dk).permute(0, 2, 1, 3) for i in range(3))
    sc = q @ k.transpose(-1, -2) * scale
    if use_mask:
        sc = sc + ((1.0 - mask.double()) * -10000.0)[:, None, None, :]
    want = (torch.softmax(sc, -1) @ v).permute(0, 2, 1, 3).reshape(B * L, H)
    d = qkv.to(dev)
    got = ops.mha_small(d[:, :H], d[:, H:2 * H], d[:, 2 * H:], mask.to(dev) if use_mask else None, B, L, heads, dk, scale)
    return float((got.cpu().double() - want).abs().max()) / 3e-6


def gate():
    dk, rows = int(rng.choice([16, 32])), int(rng.randint(1, 5000))
    q, k = torch.randn(rows, dk), torch.randn(rows, dk)
    lin = lambda o, i: (torch.randn(o, i) / i ** 0.5, torch.randn(o) * 0.1)
    fq, fk, fg = lin(dk, dk), lin(dk, dk), lin(2 * dk, dk)
    G = (q.double() @ fq[0].double().t() + fq[1].double()) * (k.double() @ fk[0].double().t() + fk[1].double())
    M = torch.sigmoid(G @ fg[0].double().t() + fg[1].double())
    qo, ko = ops.agsa_gate(q.to(dev), k.to(dev), *[(w.to(dev), b.to(dev)) for w, b in (fq, fk, fg)])
    return max(float((qo.cpu().double() - q.double() * M[:, :dk]).abs().max()), float((ko.cpu().double() - k.double() * M[:, dk:]).abs().max())) / 3e-6


def scan_any_r():
    R, Ni, Nc, D = int(rng.randint(1, 101)), int(rng.randint(1, 6)), int(rng.randint(1, 12)), int(rng.choice([32, 64, 96]))
    max_len = int(rng.randint(1, 97))
    lens = [max_len] + [int(x) for x in rng.randint(1, max_len + 1, size=Nc - 1)]
    xa = 't2i' if rng.rand() < 0.5 else 'i2t'
    norm = str(rng.choice(['clipped_l2norm', 'l2norm', 'softmax', 'no_norm', 'clipped', 'l1norm', 'clipped_l1norm']))
    agg = str(rng.choice(['LogSumExp', 'Max', 'Sum', 'Mean']))
    img = O.l2norm(torch.randn(Ni, R, D), -1)
    cap = torch.randn(Nc, max_len, D) * 0.5
    want = O.xattn_score(img, cap, lens, xa, norm, agg)
    got = ops.scan_xattn_padded(img.to(dev), cap.to(dev), lens, cross_attn=xa, raw_feature_norm=norm, agg_func=agg)
    return float((got.cpu() - want).abs().max()) / (2e-5 * (max(R, max_len) if agg == 'Sum' else 1))


def scan_i2t_mix():
    Ni, Nc, D = int(rng.randint(1, 10)), int(rng.randint(1, 60)), int(rng.choice([32, 64, 256]))
    hi = int(rng.choice([8, 20, 40, 64]))
    lens = [int(x) for x in rng.randint(1, hi + 1, size=Nc)]
    xa = 'i2t' if rng.rand() < 0.7 else 't2i'
    norm = str(rng.choice(['clipped_l2norm', 'l2norm', 'softmax', 'no_norm', 'clipped']))
    agg = str(rng.choice(['LogSumExp', 'Max', 'Sum', 'Mean']))
    img = O.l2norm(torch.randn(Ni, 36, D), -1)
    cap = torch.randn(Nc, max(lens), D) * 0.5
    want = O.xattn_score(img, cap, lens, xa, norm, agg)
    got = ops.scan_xattn_padded(img.to(dev), cap.to(dev), lens, cross_attn=xa, raw_feature_norm=norm, agg_func=agg)
    return float((got.cpu() - want).abs().max()) / (2e-5 * (36 if agg == 'Sum' else 1))


def t2i_partition():
    Ni, Nc, D = int(rng.randint(1, 12)), int(rng.randint(2, 400)), int(rng.choice([32, 128]))
    lens = rng.randint(1, int(rng.choice([10, 30, 64])) + 1, size=Nc).astype(np.int32)
    off = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.int64)
    n_rows = int(lens.sum())
    norm = str(rng.choice(['clipped_l2norm', 'l2norm', 'softmax', 'no_norm', 'clipped', 'l1norm', 'clipped_l1norm']))
    agg = str(rng.choice(['LogSumExp', 'Max', 'Sum', 'Mean']))
    img = ops.l2norm(torch.randn(Ni, 36, D, device=dev))
    words = torch.randn(n_rows, D, device=dev) * 0.3
    full = ops.scan_xattn_scores(img, words, ops.ScanPlan(off, lens, n_rows, dev), raw_feature_norm=norm, agg_func=agg)
    c0 = int(rng.randint(0, Nc - 1)); c1 = int(rng.randint(c0 + 1, Nc + 1))
    r0, r1 = int(off[c0]), int(off[c1 - 1] + lens[c1 - 1])
    sub = ops.scan_xattn_scores(img, words[r0:r1].contiguous(), ops.ScanPlan(off[c0:c1] - r0, lens[c0:c1], r1 - r0, dev), raw_feature_norm=norm, agg_func=agg)
    return 0.0 if torch.equal(sub, full[:, c0:c1]) else 2.0


def _sgraf_w(D, S, steps):
    g = torch.Generator().manual_seed(int(rng.randint(1 << 30)))
    w = {}
    def lin(name, o, i):
        r = float(np.sqrt(6.0 / (i + o)))
        w[name + ".weight"] = (torch.rand(o, i, generator=g) * 2 - 1) * r
        w[name + ".bias"] = torch.randn(o, generator=g) * 0.02
    def bn(name, n):
        w[name + ".weight"] = torch.rand(n, generator=g) * 0.4 + 0.8; w[name + ".bias"] = torch.randn(n, generator=g) * 0.05
        w[name + ".running_mean"] = torch.randn(n, generator=g) * 0.1; w[name + ".running_var"] = torch.rand(n, generator=g) + 0.5
    lin("v_global_w.embedding_local.0", D, D); bn("v_global_w.embedding_local.1", 36)
    lin("v_global_w.embedding_global.0", D, D); bn("v_global_w.embedding_global.1", D)
    lin("v_global_w.embedding_common.0", 1, D)
    lin("t_global_w.embedding_local.0", D, D); lin("t_global_w.embedding_global.0", D, D); lin("t_global_w.embedding_common.0", 1, D)
    lin("sim_tranloc_w", S, D); lin("sim_tranglo_w", S, D); lin("sim_eval_w", 1, S)
    lin("SAF_module.attn_sim_w", 1, S); bn("SAF_module.bn", 1)
    for k in range(steps):
        for nm in ("graph_query_w", "graph_key_w", "sim_graph_w"):
            lin("SGR_module.sgr%d.%s" % (k, nm), S, S)
    return w


def sgraf():
    mod = 'SAF' if rng.rand() < 0.4 else 'SGR'
    Ni, Nc, D, S = int(rng.randint(1, 20)), int(rng.randint(1, 40)), int(rng.choice([32, 64, 96])), int(rng.choice([256, 256, 64]))
    steps = int(rng.randint(1, 5)) if mod == 'SGR' else 3
    hi = int(rng.choice([6, 20, 63, 80]))
    lens = [int(x) for x in rng.randint(1, hi + 1, size=Nc)]
    img = O.l2norm(torch.randn(Ni, 36, D), -1)
    cap = O.l2norm(torch.randn(Nc, max(lens), D), -1)
    w = _sgraf_w(D, S, steps)
    want = O.sgraf_similarity(w, img, cap, lens, mod, steps)
    got = ops.sgraf_padded(img.to(dev), cap.to(dev), lens, {k: v.to(dev) for k, v in w.items()}, mod, steps)
    return float((got.cpu() - want).abs().max()) / 5e-6


def bigru():
    V, E, D = int(rng.randint(5, 300)), int(rng.choice([16, 64, 300])), int(rng.choice([32, 128, 256]))
    B = int(rng.choice([1, 3, 40, 1100]))
    hi = int(rng.randint(1, 12))
    lengths = sorted([int(x) for x in rng.randint(1, hi + 1, size=B)], reverse=True)
    ids = torch.from_numpy(rng.randint(0, V, size=(B, max(lengths))))
    bi = bool(rng.rand() < 0.7)
    rnn = torch.nn.GRU(E, D, 1, batch_first=True, bidirectional=bi)
    w = {'embed.weight': torch.empty(V, E).uniform_(-0.1, 0.1)}
    w.update({'rnn.' + k: v.detach() for k, v in rnn.state_dict().items()})
    toks = torch.cat([ids[b, :l] for b, l in enumerate(lengths)]).to(dev)
    off = torch.tensor(np.concatenate([[0], np.cumsum(lengths)[:-1]]), dtype=torch.int64, device=dev)
    last = bool(rng.rand() < 0.5)
    got = ops.gru_encode(toks, off, lengths, {k: v.to(dev) for k, v in w.items()}, bi, gather_last=last, batch_invariant=bool(rng.rand() < 0.5))
    n = min(B, 12)
    want, _ = O.encoder_text(ids[:n], lengths[:n], w, bi, False, False, None)
    worst, o = 0.0, 0
    for b in range(n):
        if last:
            worst = max(worst, float((got[b].cpu() - want[b, lengths[b] - 1]).abs().max()))
        else:
            worst = max(worst, float((got[o:o + lengths[b]].cpu() - want[b, :lengths[b]]).abs().max()))
            o += lengths[b]
    return worst / 5e-6


def gemm():
    M, N, K = int(rng.choice([1, 77, 128 * 3 + 5, 128 * 40, 128 * 700 + 3])), int(rng.choice([1, 64, 96, 128, 256, 384])), int(rng.choice([4, 40, 64, 128, 192, 256, 768]))
    act = rng.choice([None, 'relu', 'gelu', 'tanh', 'sigmoid'])
    x, wt = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev) * 0.1
    b = torch.randn(N, device=dev) if rng.rand() < 0.7 else None
    got = ops.linear(x, wt, b, act=act).double()
    idx = torch.randint(0, M, (min(M, 2000),), device=dev)
    want = x[idx].double() @ wt.double().t() + (b.double() if b is not None else 0)
    want = {None: lambda t: t, 'relu': lambda t: t.clamp(min=0), 'gelu': lambda t: 0.5 * t * (1 + torch.erf(t / 2 ** 0.5)), 'tanh': torch.tanh,
            'sigmoid': torch.sigmoid}[act](want)
    scale = float((x[idx].double().abs() @ wt.double().abs().t()).max()) + 1.0
    return float((got[idx] - want).abs().max()) / (6e-7 * scale)


torch.manual_seed(7)
family("SGRAF SAF / SGR", sgraf)
family("GRU", bigru)
family("linear (+ activations)", gemm)
family("attention (mha_small)", mha)
family("CAMERA gate (agsa_gate)", gate)
family("SCAN, any region count", scan_any_r)
family("SCAN, caption-length mixes", scan_i2t_mix)
family("SCAN t2i partition invariance", t2i_partition)
print("ok")
