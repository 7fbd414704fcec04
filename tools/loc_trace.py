#!/usr/bin/env python3
"""sgraf_loc_kernel: where a workgroup's time goes OUTSIDE the generated D loop.

ITR_LOC_TRACE=<file> makes csrc/sgraf_loc.hip record, per workgroup, its hardware id (XCC / SE / SH / CU) and s_memtime at
entry, before the asm statement (prologue + loop), after it, and after its last store has left.  This tool rebuilds every CU's
timeline from the records of ONE launch: phase lengths per workgroup, how long a CU had 0 / 1 / 2 workgroups inside the asm
statement, and the dispatch gap between a workgroup's end and its successor's entry.

    python3 tools/loc_trace.py [D] [n_img]          (run on the GPU box)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "image-text-retrieval_amd"))
TRACE = "/tmp/itr_loc_trace.bin"
os.environ["ITR_LOC_TRACE"] = TRACE
import numpy as np
import torch
import bench
from itr_amd import ops

D = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
n_img = int(sys.argv[2]) if len(sys.argv) > 2 else 16
dev = torch.device("cuda:0")
torch.manual_seed(0)
lengths, _ = bench.make_captions(5000, 8481)
off = np.concatenate([[0], np.cumsum(lengths)[:-1]]).astype(np.int64)
n_rows = int(lengths.sum())
img = ops.l2norm(torch.randn(n_img, 36, D, device=dev))
words = ops.l2norm(torch.randn(n_rows, D, device=dev))
w = {k: v.to(dev) for k, v in bench.make_sgraf_weights("SAF", D=D).items()}
plan = ops.ScanPlan(off, lengths, n_rows, dev)
for _ in range(2):
    ops.sgraf_scores(img, words, plan, w, "SAF", 3)
torch.cuda.synchronize()

rec = np.fromfile(TRACE, dtype=np.uint64).reshape(-1, 5)
rec = rec[rec[:, 4] != 0]                       # workgroups past the last tile leave at once
hw = rec[:, 0]
cu = ((hw >> np.uint64(32)) << np.uint64(8)) | ((hw >> np.uint64(8)) & np.uint64(0xff))   # xcc | se, sh, cu
t = rec[:, 1:].astype(np.int64)
pro, loop, epi = t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2]
life = t[:, 3] - t[:, 0]
print("D %d  images %d  tiles %d  workgroups %d  CUs seen %d" % (D, n_img, plan.n_tiles, len(rec), len(np.unique(cu))))
for name, v in (("before the asm statement", pro), ("asm statement (its prologue + D loop)", loop), ("epilogue", epi), ("lifetime", life)):
    print("  %-40s mean %8.0f  median %8.0f  p10 %8.0f  p90 %8.0f cycles" % (name, v.mean(), np.median(v), np.percentile(v, 10), np.percentile(v, 90)))

occ = np.zeros(4)            # time with k workgroups resident
inasm = np.zeros(4)          # time with k workgroups inside the asm statement
gaps = []
span_tot = 0
for c in np.unique(cu):
    m = t[cu == c]
    ev = []
    for a, b, c2, d in m:
        ev += [(a, 0, +1), (d, 0, -1), (b, 1, +1), (c2, 1, -1)]
    ev.sort()
    k = [0, 0]
    last = ev[0][0]
    for tt, which, dlt in ev:
        occ[min(k[0], 3)] += tt - last
        inasm[min(k[1], 3)] += tt - last
        last = tt
        k[which] += dlt
    span_tot += ev[-1][0] - ev[0][0]
    # successor gap: every end is matched with the next entry on this CU
    ends = np.sort(m[:, 3]); starts = np.sort(m[:, 0])
    for e in ends[:-2]:
        nxt = starts[np.searchsorted(starts, e)] if np.searchsorted(starts, e) < len(starts) else None
        if nxt is not None:
            gaps.append(nxt - e)
print("  per-CU time with k workgroups RESIDENT     : " + "  ".join("k=%d %.1f%%" % (k, 100 * occ[k] / span_tot) for k in range(3)))
print("  per-CU time with k workgroups IN THE ASM   : " + "  ".join("k=%d %.1f%%" % (k, 100 * inasm[k] / span_tot) for k in range(3)))
gaps = np.asarray(gaps)
print("  end -> next entry on the same CU            : mean %.0f  median %.0f  p90 %.0f cycles" % (gaps.mean(), np.median(gaps), np.percentile(gaps, 90)))
nk = D // 32
print("  asm statement per slice (incl. its prologue): %.0f cycles; two co-resident workgroups at the matrix peak would take %.0f"
      % (loop.mean() / nk, 2 * (64 * 64 + 18 * 32)))
