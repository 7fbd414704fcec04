#!/usr/bin/env python3
"""sgr_fused_kernel: cycles per phase and workgroup (ITR_SGR_TRACE=<file>, csrc/sgr_fused.hip): load, then for every graph step
P1 (query projection), P2 (attention units), P3 (graph projection), each up to its closing barrier.  Also the MFMA-bound floor of
every phase for the group's shape, so the table shows where the kernel is above it.

    python3 tools/sgr_trace.py [n_img]          (run on the GPU box)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "image-text-retrieval_amd"))
TRACE = "/tmp/itr_sgr_trace.bin"
os.environ["ITR_SGR_TRACE"] = TRACE
import numpy as np
import torch
import bench
from itr_amd import ops

n_img = int(sys.argv[1]) if len(sys.argv) > 1 else 16
D = 1024
dev = torch.device("cuda:0")
torch.manual_seed(0)
lengths, _ = bench.make_captions(5000, 8481)
off = np.concatenate([[0], np.cumsum(lengths)[:-1]]).astype(np.int64)
n_rows = int(lengths.sum())
img = ops.l2norm(torch.randn(n_img, 36, D, device=dev))
words = ops.l2norm(torch.randn(n_rows, D, device=dev))
w = {k: v.to(dev) for k, v in bench.make_sgraf_weights("SGR", D=D).items()}
ROWS = int(os.environ.get("ITR_SGR_GROUP_ROWS", "64"))      # which class of groups: 64 (default) or 32 (two workgroups per CU)
from itr_amd.settings import SETTINGS                        # (the package reads no environment variable: this tool sets the switch)
SETTINGS.sgr_group_rows = ROWS
plan = ops.ScanPlan(off, lengths, n_rows, dev)
for _ in range(2):
    ops.sgraf_scores(img, words, plan, w, "SGR", 3)
torch.cuda.synchronize()

rec = np.fromfile("%s.%d" % (TRACE, ROWS), dtype=np.uint64).reshape(-1, 20)
rec = rec[rec[:, 2] != 0]
shape = rec[:, 1]
nrows = (shape >> np.uint64(32)).astype(np.int64)
nunit = ((shape >> np.uint64(8)) & np.uint64(0xff)).astype(np.int64)
ncap = (shape & np.uint64(0xff)).astype(np.int64)
t = rec[:, 2:17].astype(np.int64)          # entry, loaded, 3 x (P1, P2 scores + softmax, P2 values, P3), end
names = ["load"] + ["step %d %s" % (k, p) for k in range(3) for p in ("P1", "P2e", "P2y", "P3")] + ["final"]
d = np.diff(t, axis=1)
print("workgroups %d   groups %d   mean rows %.1f  captions %.1f  P2 units %.1f" % (len(rec), plan.node_groups()[2], nrows.mean(), ncap.mean(), nunit.mean()))
ng = (nrows + 15) // 16
floor_proj = 2 * 2 * ng * 64 * 32          # per SIMD: 2 waves x (2 o-tiles x ng x 64 k-quads) MFMAs of 32 cycles
floor_last = 2 * 2 * 1 * 64 * 32
print("%-12s %10s %10s %10s   %s" % ("phase", "mean", "p10", "p90", "MFMA floor (per SIMD, mean)"))
for k, nm in enumerate(names):
    fl = ""
    if "P1" in nm or "P3" in nm:
        fl = "%.0f" % (floor_last if nm.startswith("step 2") else floor_proj.mean())
    print("%-12s %10.0f %10.0f %10.0f   %s" % (nm, d[:, k].mean(), np.percentile(d[:, k], 10), np.percentile(d[:, k], 90), fl))
life = t[:, -1] - t[:, 0]
print("%-12s %10.0f %10.0f %10.0f" % ("lifetime", life.mean(), np.percentile(life, 10), np.percentile(life, 90)))
hw = rec[:, 0]
cu = ((hw >> np.uint64(32)) << np.uint64(8)) | ((hw >> np.uint64(8)) & np.uint64(0xff))
# per CU: wall time covered by its workgroups and the MFMA floor of the work it did (two workgroups per CU overlap for ROWS = 32)
span = busy_floor = 0.0
for c in np.unique(cu):
    sel = cu == c
    m = t[sel]
    span += m[:, -1].max() - m[:, 0].min()
    busy_floor += (4 * floor_proj[sel] + 2 * floor_last).sum()
print("CUs seen %d; sum over CUs of (last end - first entry) %.3g cycles; projection MFMA floor of their work %.3g = %.1f%%" % (
    len(np.unique(cu)), span, busy_floor, 100 * busy_floor / span))
print("workgroup lifetimes overlapping on a CU: mean resident workgroups = %.2f" % (life.sum() / span))

# attention phases against the group's shape: extra = number of (caption, query tile) units beyond one per caption, i.e. how many
# second / third / fourth 16-node tiles the group's captions have (0: every caption has <= 16 nodes)
extra = nunit - ncap
print("%-22s %8s %10s %10s %10s" % ("units - captions", "groups", "P2e step0", "P2y step0", "P1 step0"))
for x in sorted(set(extra.tolist())):
    m = extra == x
    print("%-22d %8d %10.0f %10.0f %10.0f" % (x, int(m.sum()), d[m, 2].mean(), d[m, 3].mean(), d[m, 1].mean()))
print("%-22s %8s %10s %10s" % ("units", "groups", "P2e step0", "P2y step0"))
for x in sorted(set(nunit.tolist())):
    m = nunit == x
    print("%-22d %8d %10.0f %10.0f" % (x, int(m.sum()), d[m, 2].mean(), d[m, 3].mean()))
