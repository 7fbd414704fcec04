import sys, os, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "image-text-retrieval_amd"))
from itr_amd import ops
dev = torch.device("cuda", 0)
n_img = 1000
rng = np.random.RandomState(11); n_cap = 5 * n_img; D = 1024
lens = rng.randint(6, 21, size=n_cap).astype(np.int64); off = np.concatenate([[0], np.cumsum(lens)[:-1]])
g = torch.Generator(device=dev); g.manual_seed(11)
img = ops.l2norm(torch.randn(n_img, 36, D, device=dev, generator=g))
words = ops.l2norm(torch.randn(int(lens.sum()), D, device=dev, generator=g))
plan = ops.ScanPlan(off, lens, words.shape[0], dev)
ws = ops.scan_prepare(img, words, plan, 't2i')
S0 = ops.scan_xattn_scores(img, words, plan, workspace=ws)
for abl in ("0",):
    os.environ["ITR_SCAN_BF16_ABLATE"] = abl
    for rep in range(2):
        S3 = ops.scan_xattn_scores(img, words, plan, workspace=ws, precision='fp16x3')
        bad = ~torch.isfinite(S3)
        d = (S3 - S0).abs(); d[bad] = 0
        print("ablate", abl, "fp16x3: non-finite", int(bad.sum()), " finite max|d| %.2e" % d.max().item())
