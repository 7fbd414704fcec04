#!/usr/bin/env python3
"""Host/device time split of one bench step (default workload): where do the milliseconds outside
scan_xattn_kernel go?  Run on the GPU box:  python tools/step_breakdown.py [--workload W] [--shard P]
--shard P runs rank 0's share of a P-way sharding (image rows by count, captions by TOKEN count -- evalpipe.caption_ranges)
the way evalpipe.scan_eval orders it: towers -> [all-gather of the packed words starts] -> the columns of the rank's own
captions are scored -> [wait] -> the other ranks' columns.  The collective itself cannot run on a 1-GPU box; the tool prints
the time of the own-column launch next to the modelled exchange time it has to cover."""
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
is_sgraf = "sgraf" in wl                  # BASELINE config [4]: SGRAF (the 8-GPU config) -- the same order of work, its own scorer
cfg = dict(wl, bi_gru=True, no_txtnorm=not is_sgraf, no_imgnorm=False)
sim_w = {k: v.to(dev) for k, v in bench.make_sgraf_weights(wl["sgraf"]).items()} if is_sgraf else None


def score(img, words_t, off, lens, out):
    plan = ops.ScanPlan(off, lens, words_t.shape[0], dev)
    if is_sgraf:
        ops.sgraf_scores(img, words_t, plan, sim_w, wl["sgraf"], 3, out=out)
    else:
        ws = ops.scan_prepare(img, words_t, plan, "t2i")
        ops.scan_xattn_scores(img, words_t, plan, cross_attn="t2i", workspace=ws, out=out)
wi, wt = bench.make_weights(wl["vocab"])
g = torch.Generator(device=dev); g.manual_seed(0)
feats = ops.l2norm(torch.randn(n_img, 36, 2048, device=dev, generator=g))
lengths, tokens = bench.make_captions(n_cap, wl["vocab"])
P = args.shard
i0, i1 = evalpipe.block_range(n_img, P, 0, 4)
ranges = evalpipe.caption_ranges(n_cap, P, lengths)
c0, c1 = ranges[0]
if is_sgraf:
    cfg.update(module_name=wl["sgraf"], sgr_step=3)
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
    S = torch.empty(i1 - i0, n_cap, device=dev)
    # own columns: scored from the local word matrix while the exchange would be in flight
    ls_loc = np.zeros(c1 - c0, np.int64); off_loc = np.zeros(c1 - c0, np.int64)
    lsrt = np.asarray(lens_sorted, np.int64)
    ls_loc[np.asarray(order)] = lsrt; off_loc[np.asarray(order)] = np.concatenate([[0], np.cumsum(lsrt)[:-1]])
    score(img, words, off_loc, ls_loc, S[:, c0:c1]); t.append(sync())
    # stand-in for the gathered buffer: the full packed word matrix (encoded once outside the timing)
    if c1 < n_cap:
        score(img, words_all, cap_off_all[c1:], cap_len_all[c1:], S[:, c1:])
    t.append(sync())
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
tok_sums = [int(lengths[lo:hi].sum()) for lo, hi in ranges]
recv_gb = (sum(tok_sums) - tok_sums[0]) * 1024 * 4 / 1e9
xgmi_ms = recv_gb / (7 * 153.0 * 0.5) * 1e3          # 7 links x 153 GB/s per direction, half of peak
for _ in range(2):
    d = staged()
    print("P=%d  img %.1f  text %.1f (%d of %d tokens)  own columns %.1f  other columns %.1f  rank %.1f   total %.1f ms" % (
        (P, d[0], d[1], tok_sums[0], sum(tok_sums), d[2], d[3], d[4], d.sum())))
if P > 1:
    print("      exchange: %.2f GB received per GPU, ~%.1f ms at half of the xGMI peak -- %s the %.1f ms own-column launch it runs under; "
          "token sums per rank min %d max %d (count-balanced shards: min %d max %d)" % (
              recv_gb, xgmi_ms, "covered by" if xgmi_ms <= d[2] else "LONGER than", d[2], min(tok_sums), max(tok_sums),
              min(int(lengths[evalpipe.block_range(n_cap, P, q)[0]:evalpipe.block_range(n_cap, P, q)[1]].sum()) for q in range(P)),
              max(int(lengths[evalpipe.block_range(n_cap, P, q)[0]:evalpipe.block_range(n_cap, P, q)[1]].sum()) for q in range(P))))

if P == 1:
    def step():
        model.scan_eval(feats_local, toks, tok_off, lens_sorted, order, n_img, n_cap, sgraf_weights=sim_w)
    step()
    t0 = sync(); step(); t1 = sync()
    print("unstaged step: %.1f ms" % ((t1 - t0) * 1e3))
    pr = cProfile.Profile(); pr.enable(); step(); torch.cuda.synchronize(); pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(14)
