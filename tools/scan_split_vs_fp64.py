#!/usr/bin/env python3
"""Which SCAN kernel is closer to the float64 truth?  Scores of the fp32, bf16x3 and fp16x3 kernels on a 1 000 x 5 000 problem against
the oracle evaluated in float64 on scattered sub-blocks (STUDY_SPLIT_PRECISION.md)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "image-text-retrieval_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
from itr_amd import ops
import itr_oracle as O
dev = torch.device("cuda", 0)
rng = np.random.RandomState(11); n_img, n_cap, D = 1000, 5000, 1024
lens = rng.randint(6, 21, size=n_cap).astype(np.int64); off = np.concatenate([[0], np.cumsum(lens)[:-1]])
g = torch.Generator(device=dev); g.manual_seed(11)
img = ops.l2norm(torch.randn(n_img, 36, D, device=dev, generator=g))
words = ops.l2norm(torch.randn(int(lens.sum()), D, device=dev, generator=g))
plan = ops.ScanPlan(off, lens, words.shape[0], dev)
ws = ops.scan_prepare(img, words, plan, 't2i')
S = {p: ops.scan_xattn_scores(img, words, plan, workspace=ws, precision=p).cpu().double() for p in ('fp32', 'bf16x3', 'fp16x3')}
err = {p: [] for p in S}
for rep in range(6):
    ri = np.sort(rng.choice(n_img, 16, replace=False)); ci = np.sort(rng.choice(n_cap, 30, replace=False))
    L = int(lens[ci].max()); cap = torch.zeros(len(ci), L, D, dtype=torch.float64)
    for k, c in enumerate(ci):
        cap[k, :lens[c]] = words[off[c]:off[c] + lens[c]].cpu().double()
    want = O.xattn_score(img[ri].cpu().double(), cap, [int(lens[c]) for c in ci], 't2i', 'clipped_l2norm', 'LogSumExp', 6.0, 9.0)
    for p in S:
        err[p].append((S[p][ri][:, ci] - want).abs())
for p in S:
    e = torch.cat([x.flatten() for x in err[p]])
    print("%-7s vs float64 oracle: max %.2e  mean %.2e  (%d pairs)" % (p, e.max().item(), e.mean().item(), e.numel()))
