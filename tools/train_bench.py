#!/usr/bin/env python3
"""Time model.train_emb at the reference's training shape (batch 128, 36 x 2048 regions, coco vocabulary, word_dim 300,
embed 1024, bi-GRU): ms per step and the split forward / backward / optimizer.  Run on the GPU box."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "image-text-retrieval_amd"))
import numpy as np
import torch
from itr_amd import config as C, ops
from itr_amd.modalmodule import get_model
from itr_amd.metricmodule.evaluation import LogCollector

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="SCAN", choices=["SCAN", "VSE_PP", "SGRAF", "SAEM", "CAMERA", "VSRN"])
ap.add_argument("--module", default="SAF", choices=["SAF", "SGR"])
ap.add_argument("--batch", type=int, default=128)
ap.add_argument("--steps", type=int, default=10)
a = ap.parse_args()
dev = torch.device("cuda", 0)
cfg = C.build_config(['with', a.model, 'data_name=coco_precomp', 'bi_gru=True', 'max_violation=True'] + (['module_name=' + a.module] if a.model == 'SGRAF' else []))
cfg['vocab_size'] = 11353
cfg['img_dim'] = 2048        # precomp region features
BERT = a.model in ('SAEM', 'CAMERA')
if BERT:
    sys.path.insert(0, ROOT)
    import bench
    cfg_file, ckpt, trans = bench.bert_files(os.path.join("/tmp", "itr_bench_bert"))
    cfg.update(bert_config_file=cfg_file, init_checkpoint=ckpt, trans_cfg=trans, vocab_size=30522, batch_size=a.batch)
torch.manual_seed(0)
model = get_model(cfg)
model.train_start()
model.logger = LogCollector()
rng = np.random.RandomState(0)
B = a.batch


def batch():
    lens = sorted([int(x) for x in rng.randint(6, 21, size=B)], reverse=True)
    ids = torch.zeros(B, max(lens), dtype=torch.long)
    for b, l in enumerate(lens):
        ids[b, :l] = torch.from_numpy(rng.randint(4, 11353, size=l))
    feats = ops.l2norm(torch.randn(B, 36, 2048, device=dev))
    if a.model == 'VSRN':          # the loader's VSRN layout: every caption max_len + 1 = 61 ids (data_loader.py:117-125)
        vid = torch.zeros(B, 61, dtype=torch.long)
        for b, l in enumerate(lens):
            vid[b, :l] = ids[b, :l]
        vmask = torch.zeros(B, 61)
        vmask[:, :60] = 1
        return (feats, None, None, vid.to(dev), [61] * B, list(range(B)), vmask.to(dev), None)
    if BERT:
        L = 32
        bid = torch.from_numpy(rng.randint(1000, 30522, size=(B, L)))
        mask = torch.zeros(B, L, dtype=torch.long)
        for b, l in enumerate(lens):
            mask[b, :l] = 1
            bid[b, l:] = 0
        x1y1 = torch.rand(B, 36, 2) * 300
        boxes = torch.cat([x1y1, x1y1 + 20 + torch.rand(B, 36, 2) * 150], 2)
        return (feats, boxes.to(dev), torch.tensor([[640., 480.]]).repeat(B, 1).to(dev), bid.to(dev), lens, list(range(B)), mask.to(dev),
                torch.zeros(B, L, dtype=torch.long, device=dev))
    return (feats, None, None, ids.to(dev), lens, list(range(B)), None, None)


batches = [batch() for _ in range(4)]
for i in range(3):
    model.train_emb(batches[i % 4])
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(a.steps):
    model.train_emb(batches[i % 4])
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.steps
n_tok = sum(batches[0][4])
print("%s train_emb  batch %d (%d words): %.2f ms/step  (%.0f pairs/s, %.0f img-cap pairs scored per step); loss %.4f" % (
    a.model + ("-" + a.module if a.model == 'SGRAF' else ""), B, n_tok, dt * 1e3, B * B / dt, B * B,
    float(model.logger.meters['Loss' if 'Loss' in model.logger.meters else 'Loss1'].val)))
