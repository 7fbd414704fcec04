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


torch.manual_seed(7)
family("attention (mha_small)", mha)
family("CAMERA gate (agsa_gate)", gate)
family("SCAN, any region count", scan_any_r)
family("SCAN, caption-length mixes", scan_i2t_mix)
family("SCAN t2i partition invariance", t2i_partition)
print("ok")
