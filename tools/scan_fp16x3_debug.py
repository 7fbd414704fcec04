#!/usr/bin/env python3
"""Debug helper of the fp16x3 SCAN variant: where do fp16x3 and fp32 scores differ, and by how much."""
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "image-text-retrieval_amd"))
from itr_amd import ops
dev = torch.device("cuda", 0)
for n_img in (24, 200, 1000):
    rng = np.random.RandomState(11); n_cap = 5 * n_img; D = 1024
    lens = rng.randint(6, 21, size=n_cap).astype(np.int64); off = np.concatenate([[0], np.cumsum(lens)[:-1]])
    g = torch.Generator(device=dev); g.manual_seed(11)
    img = ops.l2norm(torch.randn(n_img, 36, D, device=dev, generator=g))
    words = ops.l2norm(torch.randn(int(lens.sum()), D, device=dev, generator=g))
    plan = ops.ScanPlan(off, lens, words.shape[0], dev)
    S0 = ops.scan_xattn_scores(img, words, plan)
    S2 = ops.scan_xattn_scores(img, words, plan, precision='fp16x3')
    bad = ~torch.isfinite(S2)
    d = (S2 - S0).abs()
    d[bad] = 0
    print(n_img, "non-finite:", int(bad.sum()), " finite max|d| %.2e mean %.2e  >1e-6: %d" % (d.max().item(), d.mean().item(), int((d > 1e-6).sum())))
    if bad.any():
        cols = torch.nonzero(bad.any(0)).flatten().tolist()
        print("   bad columns:", cols[:12], "their lens:", [int(lens[c]) for c in cols[:12]], " rows bad per col:", [int(bad[:, c].sum()) for c in cols[:6]])
        c = cols[0]
        w = words[off[c]:off[c] + lens[c]]
        print("   caption", c, "max|w| %.3e  S2 col sample" % w.abs().max().item(), S2[:4, c].tolist(), "S0", S0[:4, c].tolist())
