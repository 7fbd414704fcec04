#!/usr/bin/env python3
"""Worker of tests/test_dp_train_gpu.py: a few `model.train_emb` steps on seeded global batches, alone or as one rank of
a torch.distributed job (backend from ITR_DIST_BACKEND; gloo = several ranks on ONE GPU, host-staged collectives).
Rank 0 writes the per-step losses, gradient norms and the final parameters to --out (npz)."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "image-text-retrieval_amd"))
import numpy as np
import torch
import torch.distributed as dist

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="SCAN")
ap.add_argument("--cross-attn", default="t2i")
ap.add_argument("--module-name", default="SAF")
ap.add_argument("--live-dropout", action="store_true")
ap.add_argument("--batch", type=int, default=24)
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--out", required=True)
a = ap.parse_args()

world = int(os.environ.get("WORLD_SIZE", "1"))
backend = os.environ.get("ITR_DIST_BACKEND", "nccl")
local = 0 if backend == "gloo" else int(os.environ.get("LOCAL_RANK", "0"))
torch.cuda.set_device(local)
forced = os.environ.get("ITR_FORCE_COLLECTIVES") == "1"      # a 1-rank group whose collectives still run (RCCL on a 1-GPU box)
if world > 1 or forced:
    if forced and world == 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29655")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    dist.init_process_group(backend)

from itr_amd import config as C                                   # noqa: E402
from itr_amd.settings import SETTINGS                             # noqa: E402
SETTINGS.force_collectives = forced
from itr_amd.modalmodule import get_model                         # noqa: E402
from itr_amd.metricmodule.evaluation import LogCollector          # noqa: E402

over = ['with', a.model, 'data_name=coco_precomp', 'bi_gru=True', 'max_violation=True', 'embed_size=256', 'word_dim=64']
if a.model == 'SCAN':
    over.append('cross_attn=' + a.cross_attn)
if a.model == 'SGRAF':
    over.append('module_name=' + a.module_name)
cfg = C.build_config(over)
cfg['vocab_size'] = 500
cfg['img_dim'] = 128
if a.model == 'SGRAF':
    cfg.update(embed_size=32, word_dim=16, sim_dim=16, sgr_step=3, learning_rate=2e-3)
if a.model in ('SAEM', 'CAMERA'):
    # a tiny random BERT + transformer config (dropout 0: the shards of a data-parallel run draw other masks than one process)
    import json
    from itr_amd.modalmodule import bert
    d = a.out + ".files_rank%s" % os.environ.get("RANK", "0")
    os.makedirs(d, exist_ok=True)
    bc = dict(vocab_size=500, hidden_size=32, num_hidden_layers=2, num_attention_heads=2, intermediate_size=64, max_position_embeddings=40,
              type_vocab_size=2, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    json.dump(bc, open(os.path.join(d, 'bert_config.json'), 'w'))
    json.dump(dict(bc, num_hidden_layers=1, vocab_size=10), open(os.path.join(d, 'trans_cfg.json'), 'w'))
    torch.manual_seed(99)
    bm = bert.BertModel(bert.BertConfig.from_dict(bc))
    for p_ in bm.parameters():
        p_.data.normal_(0, 0.05)
    torch.save(bm.state_dict(), os.path.join(d, 'pytorch_model.bin'))
    cfg.update(bert_config_file=os.path.join(d, 'bert_config.json'), init_checkpoint=os.path.join(d, 'pytorch_model.bin'),
               trans_cfg=os.path.join(d, 'trans_cfg.json'), final_dims=32, embed_size=32, learning_rate=1e-3)
    if a.model == 'CAMERA':
        cfg.update(img_dim=24, head=2, smry_k=12, drop=0.0, smry_lamda=0.01)
if a.model == 'VSRN':
    cfg.update(img_dim=20, embed_size=32, word_dim=12, vocab_size=40, dim_vid=32, dim_hidden=16, dim_word=10, max_len=8, input_dropout_p=0.0,
               rnn_dropout_p=0.0, learning_rate=2e-3)
torch.manual_seed(1234)
model = get_model(cfg)
if a.model == 'VSRN':
    model.caption_model.cuda()
if a.model == 'SGRAF' and not a.live_dropout:
    # the reference hard-codes p = 0.4 dropout sites; the shards of a data-parallel run draw other masks than one process
    model.txt_enc.dropout_p = 0.0
    for m_ in model.sim_enc.modules():
        if isinstance(m_, torch.nn.Dropout):
            m_.p = 0.0
model.train_start()
model.logger = LogCollector()
rng = np.random.RandomState(7)
B = a.batch
losses, gnorms, grads1 = [], [], []


def keep_grads():
    if not grads1:     # the (summed, unclipped) gradient of the first step, as Adam.step saw it
        grads1.append(torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).detach().reshape(-1) for p in model.params
                                 if p.requires_grad]).cpu().numpy())


for step in range(a.steps):
    lens = sorted([int(x) for x in rng.randint(3, 15, size=B)], reverse=True)
    ids = torch.zeros(B, max(lens), dtype=torch.long)
    for b, l in enumerate(lens):
        ids[b, :l] = torch.from_numpy(rng.randint(4, 500, size=l))
    feats = torch.from_numpy(rng.randn(B, 36, 128).astype(np.float32))
    feats = feats / feats.norm(dim=-1, keepdim=True)
    if a.model == 'VSRN':
        # the VSRN loader's layout: every caption padded to max_len + 1 ids, all "lengths" equal (data_loader.py:100-121)
        L = cfg['max_len'] + 1
        ids = torch.zeros(B, L, dtype=torch.long)
        mask = torch.zeros(B, L, dtype=torch.long)
        for b in range(B):
            l = int(rng.randint(3, L))
            ids[b, :l] = torch.from_numpy(rng.randint(1, 40, size=l))
            mask[b, :l] = 1
        feats = torch.from_numpy(rng.randn(B, 36, 20).astype(np.float32))
        model.train_emb((feats, None, None, ids, [L] * B, list(range(B)), mask, None))
        losses.append(float(model.logger.meters['Loss'].val))
        gnorms.append(float(model.optimizer.last_grad_norm[0]))
        keep_grads()
        continue
    if a.model == 'CAMERA':
        L = 14
        ids = torch.from_numpy(rng.randint(1, 500, size=(B, L)))
        mask = torch.zeros(B, L, dtype=torch.long)
        for b, l in enumerate(lens):
            mask[b, :l] = 1
            ids[b, l:] = 0
        feats = torch.from_numpy(rng.randn(B, 36, 24).astype(np.float32))
        wh = torch.from_numpy(rng.randint(200, 640, size=(B, 2)).astype(np.float32))
        xy = rng.rand(B, 36, 2, 2).astype(np.float32)
        xy.sort(axis=2)                                                  # x0 <= x1, y0 <= y1 as fractions of the image
        boxes = torch.from_numpy(np.concatenate([xy[:, :, 0], xy[:, :, 1]], -1)) * torch.cat([wh, wh], 1)[:, None, :]
        model.train_emb((feats, boxes, wh, ids, lens, list(range(B)), mask, torch.zeros(B, L, dtype=torch.long)))
        losses.append(float(model.logger.meters['Loss'].val))
        gnorms.append(float(model.optimizer.last_grad_norm[0]))
        keep_grads()
        continue
    if a.model == 'SAEM':
        L = 14
        ids = torch.from_numpy(rng.randint(1, 500, size=(B, L)))
        mask = torch.zeros(B, L, dtype=torch.long)
        for b, l in enumerate(lens):
            mask[b, :l] = 1
            ids[b, l:] = 0
        model.train_emb((feats, None, None, ids, lens, list(range(B)), mask, torch.zeros(B, L, dtype=torch.long)))
        losses.append(float(model.logger.meters['Loss1'].val) + float(model.logger.meters['Loss2'].val))
        gnorms.append(float(model.optimizer.last_grad_norm[0]))
        keep_grads()
        continue
    model.train_emb((feats, None, None, ids, lens, list(range(B)), None, None))
    losses.append(float(model.logger.meters['Loss'].val))
    gnorms.append(float(model.optimizer.last_grad_norm[0]))
    keep_grads()
torch.cuda.synchronize()
if world == 1 or dist.get_rank() == 0:
    flat = torch.cat([p.detach().reshape(-1) for p in model.params if p.requires_grad]).cpu().numpy()
    np.savez(a.out, dp_on=int(model.optimizer.comm is not None), dp_world=(model.optimizer.comm.world if model.optimizer.comm is not None else 1), params=flat, grads1=grads1[0], losses=np.asarray(losses), gnorms=np.asarray(gnorms), lr=cfg['learning_rate'])
if dist.is_initialized():
    dist.barrier()
    dist.destroy_process_group()
