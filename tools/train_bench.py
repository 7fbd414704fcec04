#!/usr/bin/env python3
"""Time model.train_emb (the reference times every training batch: itr/utils.py:80-102) at the reference's training shapes:
batch 128 (SAEM 64), 36 x 2048 region features, coco vocabulary, word_dim 300, embed 1024, bi-GRU / BERT-base 32 tokens.

    python tools/train_bench.py --model SGRAF --module SGR            one family, one text line
    python tools/train_bench.py --all --json                          every family, ONE JSON line (bench.py's `train_configs`)

Per family: ms per step (wall, synchronised), the forward / backward / optimizer split from HIP events on the stream the kernels run
on, pairs scored per second, a flop model (3 x the forward flop of SURVEY 8(d) for the trained layers, 1 x for the frozen BERT) as a
fraction of the fp32 MFMA peak, and -- for the families whose training step oracle/itr_oracle.py restates (VSE++, SCAN, SGRAF) -- the
oracle's step on the host cores beside it.  Run on the GPU box."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (os.path.join(ROOT, "image-text-retrieval_amd"), ROOT):
    if _p not in sys.path:
        sys.path.insert(0, _p)
import numpy as np
import torch

PEAK_F32_TFLOPS = 157.3          # MI355X fp32 MFMA = vector rate (/opt/skills/guides/MI355X_MICROARCH.md)
FAMILIES = [("VSE_PP", None, 128), ("SCAN", None, 128), ("SGRAF", "SAF", 128), ("SGRAF", "SGR", 128), ("SAEM", None, 64), ("CAMERA", None, 128),
            ("VSRN", None, 128)]
V_COCO = 11353


def make_batches(model_name, B, n, rng, dev):
    from itr_amd import ops
    out = []
    bert = model_name in ("SAEM", "CAMERA")
    for _ in range(n):
        lens = sorted([int(x) for x in rng.randint(6, 21, size=B)], reverse=True)
        ids = torch.zeros(B, max(lens), dtype=torch.long)
        for b, l in enumerate(lens):
            ids[b, :l] = torch.from_numpy(rng.randint(4, V_COCO, size=l))
        feats = ops.l2norm(torch.randn(B, 36, 2048, device=dev))
        if model_name == "VSRN":      # the loader's VSRN layout: every caption max_len + 1 = 61 ids (data_loader.py:117-125)
            vid = torch.zeros(B, 61, dtype=torch.long)
            for b, l in enumerate(lens):
                vid[b, :l] = ids[b, :l]
            vmask = torch.zeros(B, 61)
            vmask[:, :60] = 1
            out.append((feats, None, None, vid.to(dev), [61] * B, list(range(B)), vmask.to(dev), None))
        elif bert:
            L = 32
            bid = torch.from_numpy(rng.randint(1000, 30522, size=(B, L)))
            mask = torch.zeros(B, L, dtype=torch.long)
            for b, l in enumerate(lens):
                mask[b, :l] = 1
                bid[b, l:] = 0
            x1y1 = torch.rand(B, 36, 2) * 300
            boxes = torch.cat([x1y1, x1y1 + 20 + torch.rand(B, 36, 2) * 150], 2)
            out.append((feats, boxes.to(dev), torch.tensor([[640., 480.]]).repeat(B, 1).to(dev), bid.to(dev), lens, list(range(B)), mask.to(dev),
                        torch.zeros(B, L, dtype=torch.long, device=dev)))
        else:
            out.append((feats, None, None, ids.to(dev), lens, list(range(B)), None, None))
    return out


def flop_model(model_name, module, B, lens, D=1024, S=256):
    """Forward flop of one step by SURVEY 8(d)'s per-unit figures; trained layers count 3 x (forward, dX, dW), the frozen BERT 1 x."""
    T = float(sum(lens))
    W = np.asarray(lens, dtype=np.float64)
    img = 2.0 * 36 * 2048 * D * B
    gru = 16.27e6 * T
    if model_name == "VSE_PP":
        fwd = 2.0 * 2048 * D * B + gru + 2.0 * D * B * B
        return 3 * fwd
    if model_name == "SCAN":
        return 3 * (img + gru + B * 153600.0 * T)
    if model_name == "SGRAF":
        pair = 4.0 * 36 * W * D + 2.0 * W * D * S + 2.0 * D * S + 2.0 * S
        if module == "SGR":
            pair = pair + 3 * (6.0 * (W + 1) * S * S + 4.0 * (W + 1) ** 2 * S)
        else:
            pair = pair + 4.0 * (W + 1) * S
        return 3 * (img + gru + B * float(pair.sum()))
    bert = 5.47e9 * B
    if model_name == "SAEM":
        return bert + 3 * (B * 36 * (2.0 * 2048 * 256 + 12.0 * 256 * 256) + 2.0 * 256 * B * B)
    if model_name == "CAMERA":
        return bert + 3 * (1.55e9 * B + 1.45e9 * B + 2.0 * 12 * 2048 * B * B)
    return None


def oracle_cpu_step(model_name, batch, cfg, model=None):
    """The oracle's own train_emb restatement (oracle/itr_oracle.py: gru_model_train_step; SGRAF: sgraf_model_train_grads, pinned by G20)
    on the host cores, same shapes."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import itr_oracle as O
    if model_name == "SGRAF":      # the model's own weights on the host; forward + backward + clip (the Adam update is noise next to them)
        cpu = lambda m: {k: v.detach().cpu() for k, v in m.state_dict().items()}
        wi, wt, ws = cpu(model.img_enc), cpu(model.txt_enc), cpu(model.sim_enc)
        ocfg = dict(bi_gru=True, margin=0.2, max_violation=True, grad_clip=2.0, module_name=cfg['module_name'], sgr_step=cfg.get('sgr_step', 3))
        feats, ids, lens = batch[0].cpu(), batch[3].cpu(), batch[4]
        t0 = time.perf_counter()
        O.sgraf_model_train_grads(wi, wt, ws, feats, ids, lens, ocfg)
        return time.perf_counter() - t0
    kind = {"VSE_PP": "VSE++", "SCAN": "SCAN"}[model_name]
    torch.manual_seed(0)
    D, E, F_ = 1024, 300, 2048
    wi = {'fc.weight': torch.empty(D, F_).uniform_(-0.03, 0.03), 'fc.bias': torch.zeros(D)}
    rnn = torch.nn.GRU(E, D, 1, batch_first=True, bidirectional=True)
    wt = {'embed.weight': torch.empty(V_COCO, E).uniform_(-0.1, 0.1)}
    wt.update({'rnn.' + k: v.detach() for k, v in rnn.state_dict().items()})
    feats, ids, lens = batch[0].cpu(), batch[3].cpu(), batch[4]
    ocfg = dict(bi_gru=True, margin=0.2, max_violation=True, learning_rate=2e-4, grad_clip=2.0, cross_attn='t2i', raw_feature_norm='clipped_l2norm',
                agg_func='LogSumExp', lambda_lse=6.0, lambda_softmax=9.0)
    t0 = time.perf_counter()
    O.gru_model_train_step(kind, wi, wt, feats, ids, lens, ocfg)
    return time.perf_counter() - t0


def run(model_name, module=None, batch=128, steps=10, warmup=3, cpu=False, dev=None):
    from itr_amd import config as C
    from itr_amd.modalmodule import get_model
    from itr_amd.metricmodule.evaluation import LogCollector
    dev = dev or torch.device("cuda", 0)
    cfg = C.build_config(['with', model_name, 'data_name=coco_precomp', 'bi_gru=True', 'max_violation=True'] +
                         (['module_name=' + module] if model_name == 'SGRAF' else []))
    cfg['vocab_size'] = V_COCO
    cfg['img_dim'] = 2048        # precomp region features
    if model_name in ('SAEM', 'CAMERA'):
        import bench
        cfg_file, ckpt, trans = bench.bert_files(os.path.join("/tmp", "itr_bench_bert"))
        cfg.update(bert_config_file=cfg_file, init_checkpoint=ckpt, trans_cfg=trans, vocab_size=30522, batch_size=batch)
    torch.manual_seed(0)
    model = get_model(cfg)
    model.train_start()
    model.logger = LogCollector()
    rng = np.random.RandomState(0)
    batches = make_batches(model_name, batch, 4, rng, dev)
    # forward | backward | optimizer boundaries: events on torch's current stream (the stream every kernel of the step is launched on)
    marks = {}
    ev = lambda: torch.cuda.Event(enable_timing=True)
    orig_step, orig_opt = model._step, model.optimizer.step

    def _step(loss, *a, **k):
        marks['fwd_end'] = ev(); marks['fwd_end'].record()
        return orig_step(loss, *a, **k)

    def _opt(*a, **k):
        marks['bwd_end'] = ev(); marks['bwd_end'].record()
        return orig_opt(*a, **k)
    model._step, model.optimizer.step = _step, _opt
    for i in range(warmup):
        model.train_emb(batches[i % 4])
    torch.cuda.synchronize()
    split = np.zeros(3)
    t0 = time.perf_counter()
    for i in range(steps):
        e0, e3 = ev(), ev()
        e0.record()
        model.train_emb(batches[i % 4])
        e3.record()
        marks['pairs'] = marks.get('pairs', []) + [(e0, marks['fwd_end'], marks['bwd_end'], e3)]
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    for e0, e1, e2, e3 in marks['pairs']:
        split += np.array([e0.elapsed_time(e1), e1.elapsed_time(e2), e2.elapsed_time(e3)])
    split /= steps
    lens = batches[0][4] if model_name != 'VSRN' else [61] * batch
    fm = flop_model(model_name, module, batch, batches[0][4])
    name = model_name + ("-" + module if module else "")
    meter = model.logger.meters['Loss' if 'Loss' in model.logger.meters else 'Loss1']
    row = {"family": name, "batch": batch, "words": int(sum(lens)), "steps": steps, "warmup": warmup, "ms_per_step": round(dt * 1e3, 3),
           "pairs_per_s": round(batch * batch / dt, 1), "forward_ms": round(float(split[0]), 3), "backward_ms": round(float(split[1]), 3),
           "optimizer_ms": round(float(split[2]), 3), "loss": round(float(meter.val), 4),
           "reference": "Models.py train_emb (:115-145, :205-225, :444-464, :518-546, :606-645); utils.py:80-102 times each batch"}
    if fm is not None:
        row["flop_model"] = fm
        row["flop_model_note"] = "3 x SURVEY 8(d) forward flop of the trained layers (+ 1 x the frozen BERT forward)"
        row["tflops"] = round(fm / dt / 1e12, 2)
        row["frac_of_fp32_mfma_peak"] = round(fm / dt / 1e12 / PEAK_F32_TFLOPS, 4)
    if cpu:
        if model_name in ("VSE_PP", "SCAN", "SGRAF"):
            # The CPU leg runs AFTER every family's GPU timing (main): a 4 ms step is ~300 launches, and the OpenMP threads of a CPU leg
            # keep spinning on the host cores for a while after it ends (SCAN measured 5.5 ms right after VSE++'s CPU leg, 3.95 alone).
            class _Frozen(object):      # what oracle_cpu_step reads of the model, moved to the host now (the model itself is released)
                pass
            frozen = None
            if model_name == "SGRAF":
                frozen = _Frozen()
                for part in ("img_enc", "txt_enc", "sim_enc"):
                    sd = {k: v.detach().cpu() for k, v in getattr(model, part).state_dict().items()}
                    holder = _Frozen()
                    holder.state_dict = (lambda sd_=sd: sd_)
                    setattr(frozen, part, holder)
            batch0 = tuple(t.cpu() if torch.is_tensor(t) else t for t in batches[0])

            def cpu_leg(row=row, dt=dt):
                torch.set_num_threads(min(32, os.cpu_count() or 1))
                s = oracle_cpu_step(model_name, batch0, cfg, frozen)
                fn = "sgraf_model_train_grads (forward, backward, clip)" if model_name == "SGRAF" else "gru_model_train_step"
                row["cpu_oracle"] = {"ms_per_step": round(s * 1e3, 1), "threads": torch.get_num_threads(),
                                     "kind": "port (oracle/itr_oracle.py %s)" % fn, "sample": "1 step, same shapes", "speedup": round(s / dt, 1)}
            row["_cpu_leg"] = cpu_leg
        else:
            row["cpu_oracle"] = None
            row["cpu_oracle_note"] = "oracle/ restates this family's evaluation path only; its training goldens (G18, G19, G21) are the imported reference's own train_emb"
    return row


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="SCAN", choices=["SCAN", "VSE_PP", "SGRAF", "SAEM", "CAMERA", "VSRN"])
    ap.add_argument("--module", default="SAF", choices=["SAF", "SGR"])
    ap.add_argument("--batch", type=int, default=None)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--all", action="store_true", help="every model family at the reference's batch size")
    ap.add_argument("--json", action="store_true")
    ap.add_argument("--cpu", action="store_true", help="time the oracle's training step on the host cores beside VSE++ / SCAN")
    ap.add_argument("--budget", type=float, default=1e9, help="--all: seconds after which no further family is started")
    a = ap.parse_args()
    t_start = time.perf_counter()
    rows = {}
    fams = FAMILIES if a.all else [(a.model, a.module if a.model == 'SGRAF' else None, a.batch or (64 if a.model == 'SAEM' else 128))]
    for model_name, module, batch in fams:
        name = model_name + ("-" + module if module else "")
        if time.perf_counter() - t_start > a.budget:
            rows[name] = {"error": "not started: the time budget of the run was spent"}
            continue
        try:
            r = run(model_name, module, a.batch or batch, a.steps, a.warmup, cpu=a.cpu)
        except Exception as e:      # noqa: BLE001  (one family must not take the others' rows with it)
            r = {"family": name, "error": "%s: %s" % (type(e).__name__, e)}
        rows[name] = r
        if not a.json:
            if "error" in r:
                print("%s train_emb: %s" % (name, r["error"]))
            else:
                print("%s train_emb  batch %d (%d words): %.2f ms/step  (fwd %.2f | bwd %.2f | opt %.2f ms; %.0f pairs/s%s); loss %.4f" % (
                    name, r["batch"], r["words"], r["ms_per_step"], r["forward_ms"], r["backward_ms"], r["optimizer_ms"], r["pairs_per_s"],
                    "; %.1f TFLOP/s = %.3f of the fp32 MFMA peak" % (r["tflops"], r["frac_of_fp32_mfma_peak"]) if "tflops" in r else "", r["loss"]))
        torch.cuda.empty_cache()
    for name, r in rows.items():          # the CPU legs, after every GPU timing
        leg = r.pop("_cpu_leg", None) if isinstance(r, dict) else None
        if leg is not None and time.perf_counter() - t_start <= a.budget:
            leg()
            if not a.json:
                print("%s: oracle on the host cores %.0f ms per step (%d threads): %.0f x" % (name, r["cpu_oracle"]["ms_per_step"], r["cpu_oracle"]["threads"],
                                                                                       r["cpu_oracle"]["speedup"]))
    if a.json:
        print(json.dumps({"train_configs": rows, "wall_s": round(time.perf_counter() - t_start, 1)}))


if __name__ == "__main__":
    main()
