#!/usr/bin/env python3
"""VSE++ text tower (5 000 captions, last state) per number of interleaved caption chains (ITR_GRU_CHAINS).  Run on the GPU box."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "image-text-retrieval_amd"))
import numpy as np, torch
import bench
from itr_amd import ops
dev = torch.device("cuda:0")
for n_cap, vocab in ((5000, 8481), (25000, 11353)):
    wi, wt = bench.make_weights(vocab)
    wt = {k: v.to(dev) for k, v in wt.items()}
    lengths, tokens = bench.make_captions(n_cap, vocab)
    toks, tok_off, lens_sorted, order = bench.shard_captions(lengths, tokens, 0, n_cap, dev)
    ref = None
    for ch in (1, 2, 3, 4, 0):
        for _ in range(3): out = ops.gru_encode(toks, tok_off, lens_sorted, wt, True, gather_last=True, chains=ch)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): out = ops.gru_encode(toks, tok_off, lens_sorted, wt, True, gather_last=True, chains=ch)
        e1.record(); torch.cuda.synchronize()
        ref = out if ref is None else ref
        print("captions %d  chains %d: %.3f ms  bit-identical to 1 chain: %s" % (n_cap, ch, e0.elapsed_time(e1) / 10, torch.equal(ref, out)))
