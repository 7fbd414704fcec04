#!/usr/bin/env python3
"""Split-bf16 ("bf16x3") study on whole workloads (STUDY_SPLIT_PRECISION.md): encode + score a sample of a pooled workload twice in ONE
process -- exact fp32 GEMMs, then ITR_GEMM_BF16X3 routing of linear / linear_strided / cosine_scores -- and report the score
deviation, the rank agreement and the time of both.  Run on the GPU box:  python tools/bf16x3_study.py SAEM|CAMERA|VSRN|VSE++"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "image-text-retrieval_amd"))
import numpy as np
import torch
import bench
from itr_amd import config as C, evalpipe, ops
from itr_amd.modalmodule import get_model

kind = sys.argv[1] if len(sys.argv) > 1 else "SAEM"
n_img = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
n_cap = 5 * n_img
dev = torch.device("cuda", 0)
torch.manual_seed(0)
if kind in ("SAEM", "CAMERA"):
    cfg_file, ckpt, trans = bench.bert_files(os.path.join("/tmp", "itr_bench_bert"))
    cfg = C.build_config(['with', kind, 'data_name=coco_precomp'])
    cfg.update(bert_config_file=cfg_file, init_checkpoint=ckpt, trans_cfg=trans, vocab_size=30522)
    model = get_model(cfg)
    model.val_start()
    feats, boxes, imgs_wh, ids, mask, types, lengths = bench.pooled_inputs(n_img, n_cap, kind, dev)
    pe = evalpipe.PooledModelEval(model, evalpipe.Comm(), batch=1024)

    def run():
        return pe.eval(feats, boxes, imgs_wh, ids, mask, types, [int(x) for x in lengths], n_img, n_cap)
else:
    raise SystemExit("kinds: SAEM, CAMERA")


def timed():
    run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    S, ranks = run()
    torch.cuda.synchronize()
    return S, ranks, time.perf_counter() - t0


ops.BF16X3 = False
S0, r0, t0 = timed()
which = os.environ.get("STUDY_VARIANT", "bf16x3")          # or fp16x3
setattr(ops, "FP16X3" if which == "fp16x3" else "BF16X3", True)
S1, r1, t1 = timed()
d = (S1 - S0).abs()
agree_i = float((np.asarray(r0[0]) == np.asarray(r1[0])).mean())
agree_t = float((np.asarray(r0[2]) == np.asarray(r1[2])).mean())
rec0 = ops.recall_from_ranks(r0[0])[:3] + ops.recall_from_ranks(r0[2])[:3]
rec1 = ops.recall_from_ranks(r1[0])[:3] + ops.recall_from_ranks(r1[2])[:3]
print(("%s %d x %d: fp32 %.1f ms, " + which + " %.1f ms (x%.2f); max|dS| %.2e mean|dS| %.2e (|S| max %.3f); identical ranks i2t %.2f%% t2i %.2f%%; "
      "max |dRecall@K| %.3f") % (kind, n_img, n_cap, t0 * 1e3, t1 * 1e3, t0 / t1, d.max().item(), d.mean().item(), S0.abs().max().item(),
                               100 * agree_i, 100 * agree_t, max(abs(a - b) for a, b in zip(rec0, rec1))))
