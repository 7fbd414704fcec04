#!/usr/bin/env python3
"""Write a synthetic precomp dataset in the reference's file layout (README.md:395-441, data_loader.py:52-80) with the
SURVEY 8d statistics: <out>/data/<name>/test_ims.npy (N, 36, 2048) fp32 l2-normalised rows, test_caps.txt (5 N lines of
4..18 random vocabulary words = 6..20 tokens with <start>/<end>), <out>/vocab/<name>_vocab.json.  Used by
`bench.py --from-files <out>` to time the file -> rank path (mmap -> pinned -> HBM, tokenise, encode, score, rank).

    python tools/make_synth_precomp.py /tmp/itr_synth --n-img 5000
"""
import argparse
import json
import os

import numpy as np

ap = argparse.ArgumentParser()
ap.add_argument("out")
ap.add_argument("--n-img", type=int, default=5000)
ap.add_argument("--vocab", type=int, default=11353)
ap.add_argument("--name", default="coco_precomp")
ap.add_argument("--seed", type=int, default=0)
a = ap.parse_args()

d = os.path.join(a.out, "data", a.name)
os.makedirs(d, exist_ok=True)
os.makedirs(os.path.join(a.out, "vocab"), exist_ok=True)
rng = np.random.RandomState(a.seed)
ims = np.lib.format.open_memmap(os.path.join(d, "test_ims.npy"), mode="w+", dtype=np.float32, shape=(a.n_img, 36, 2048))
try:
    import torch
    gpu = torch.cuda.is_available()
except ImportError:
    gpu = False
for r0 in range(0, a.n_img, 250):
    r1 = min(a.n_img, r0 + 250)
    if gpu:
        g = torch.Generator(device="cuda")
        g.manual_seed(a.seed * 100003 + r0)
        x = torch.randn(r1 - r0, 36, 2048, device="cuda", generator=g)
        x = x / (x.pow(2).sum(-1, keepdim=True).sqrt() + 1e-8)
        ims[r0:r1] = x.cpu().numpy()
    else:
        x = rng.standard_normal((r1 - r0, 36, 2048)).astype(np.float32)
        ims[r0:r1] = x / (np.sqrt((x * x).sum(-1, keepdims=True)) + 1e-8)
ims.flush()
del ims
words = ["<pad>", "<start>", "<end>", "<unk>"] + ["w%d" % i for i in range(4, a.vocab)]
json.dump({"word2idx": {w: i for i, w in enumerate(words)}, "idx2word": {str(i): w for i, w in enumerate(words)}, "idx": len(words)},
          open(os.path.join(a.out, "vocab", "%s_vocab.json" % a.name), "w"))
lens = rng.randint(4, 19, size=5 * a.n_img)
with open(os.path.join(d, "test_caps.txt"), "w") as f:
    for n in lens:
        f.write(" ".join("w%d" % t for t in rng.randint(4, a.vocab, size=int(n))) + "\n")
print("wrote %s: %d images (%.2f GB), %d captions, %d tokens incl. <start>/<end>" % (
    d, a.n_img, a.n_img * 36 * 2048 * 4 / 1e9, 5 * a.n_img, int(lens.sum()) + 10 * a.n_img))
