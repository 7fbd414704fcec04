#!/usr/bin/env python3
"""Host/device time split of one bench step (default workload): where do the milliseconds outside
scan_xattn_kernel go?  Run on the GPU box:  python tools/step_breakdown.py [--workload W] [--shard P]
--shard P runs rank 0's share of a P-way row/caption sharding (what one of P GPUs would do, minus the collectives)."""
import argparse
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "image-text-retrieval_amd"))
import numpy as np
import torch
import bench
from itr_amd import evalpipe, ops

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="scan_t2i_coco5k")
ap.add_argument("--shard", type=int, default=1)
args = ap.parse_args()
wl = bench.WORKLOADS[args.workload]
dev = torch.device("cuda", 0)
n_img, n_cap = wl["n_img"], 5 * wl["n_img"]
cfg = dict(wl, bi_gru=True, no_txtnorm=True, no_imgnorm=False)
wi, wt = bench.make_weights(wl["vocab"])
g = torch.Generator(device=dev); g.manual_seed(0)
feats = ops.l2norm(torch.randn(n_img, 36, 2048, device=dev, generator=g))
lengths, tokens = bench.make_captions(n_cap, wl["vocab"])
P = args.shard
i0, i1 = evalpipe.block_range(n_img, P, 0, 4)
c0, c1 = evalpipe.block_range(n_cap, P, 0)
model = evalpipe.GruModelEval({k: v.to(dev) for k, v in wi.items()}, {k: v.to(dev) for k, v in wt.items()}, cfg)
feats_local = feats[i0:i1].contiguous()
toks, tok_off, lens_sorted, order = bench.shard_captions(lengths, tokens, c0, c1, dev)
toks_all, tok_off_all, lens_all, order_all = bench.shard_captions(lengths, tokens, 0, n_cap, dev)


def sync():
    torch.cuda.synchronize()
    return time.perf_counter()


def staged():
    t = [sync()]
    img = model.encode_images(feats_local); t.append(sync())
    words = model.encode_captions(toks, tok_off, lens_sorted); t.append(sync())
    # stand-in for the all-gather: the full packed word matrix (encoded once outside the timing)
    plan = ops.ScanPlan(cap_off_all, cap_len_all, words_all.shape[0], dev); t.append(sync())
    ws = ops.scan_prepare(img, words_all, plan, "t2i"); t.append(sync())
    S = ops.scan_xattn_scores(img, words_all, plan, cross_attn="t2i", workspace=ws); t.append(sync())
    r = evalpipe.finalize_ranks(evalpipe.Comm(), S, 0, n_img if P == 1 else (i1 - i0)); t.append(sync())
    return np.diff(t) * 1e3


words_all = model.encode_captions(toks_all, tok_off_all, lens_all)
ls = np.asarray(lens_all, np.int64)
off_sorted = np.concatenate([[0], np.cumsum(ls)[:-1]])
cap_len_all = np.zeros(n_cap, np.int64); cap_off_all = np.zeros(n_cap, np.int64)
cap_len_all[np.asarray(order_all)] = ls; cap_off_all[np.asarray(order_all)] = off_sorted
if P > 1:   # rank-local GT layout is only meaningful for P == 1; ranks here are timing-only
    pass
staged()
for _ in range(2):
    d = staged()
    print("P=%d  img %.1f  text %.1f  plan(host) %.1f  prepare %.1f  scan %.1f  rank %.1f   total %.1f ms" % ((P,) + tuple(d) + (d.sum(),)))

if P == 1:
    def step():
        model.scan_eval(feats_local, toks, tok_off, lens_sorted, order, n_img, n_cap)
    step()
    t0 = sync(); step(); t1 = sync()
    print("unstaged step: %.1f ms" % ((t1 - t0) * 1e3))
    pr = cProfile.Profile(); pr.enable(); step(); torch.cuda.synchronize(); pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(14)
