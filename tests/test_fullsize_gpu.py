"""GPU: the SCAN scorer + count ranker at BASELINE.json's FULL sizes (MS-COCO 5k images x 25k captions,
f30k 1k x 5k), where the CPU oracle would need hours.  Parity is asserted through size-independent properties
of the path plus scattered oracle spot checks:

  * spot check: a random 24-image x 40-caption sub-problem picked out of the full matrix equals the oracle
    (scores of one pair depend on that pair only);
  * shard invariance: scoring a row block alone gives BIT-identical rows (the multi-GPU row sharding contract);
  * caption permutation: scoring the captions in another order (=> another tile packing) permutes the columns;
  * ranker: ranks of sampled rows / columns equal a host argsort; the rank histogram is a permutation-count
    checksum (sum over all columns of #greater equals the number of strictly-greater pairs)."""
import numpy as np
import pytest
import torch

import itr_oracle as O
from itr_amd import ops

pytestmark = pytest.mark.gpu


def _problem(n_img, seed, dev, D=1024):
    rng = np.random.RandomState(seed)
    n_cap = 5 * n_img
    lens = rng.randint(6, 21, size=n_cap).astype(np.int64)      # SURVEY 8d caption lengths
    off = np.concatenate([[0], np.cumsum(lens)[:-1]])
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    img = ops.l2norm(torch.randn(n_img, 36, D, device=dev, generator=g))
    words = torch.randn(int(lens.sum()), D, device=dev, generator=g) * 0.3
    return img, words, lens, off


@pytest.mark.parametrize("n_img,xa", [(5000, 't2i'), (1000, 't2i'), (1000, 'i2t'), (5000, 'i2t')])
def test_scan_full_size(dev, n_img, xa):
    img, words, lens, off = _problem(n_img, 11, dev)
    n_cap = len(lens)
    plan = ops.ScanPlan(off, lens, words.shape[0], dev)
    kw = dict(cross_attn=xa, lambda_lse=6.0 if xa == 't2i' else 20.0, lambda_softmax=9.0 if xa == 't2i' else 4.0)
    S = ops.scan_xattn_scores(img, words, plan, **kw)
    assert S.shape == (n_img, n_cap) and bool(torch.isfinite(S).all())

    # -- oracle spot check on scattered rows / columns
    rng = np.random.RandomState(5)
    ri = np.sort(rng.choice(n_img, 24, replace=False))
    ci = np.sort(rng.choice(n_cap, 40, replace=False))
    L = int(lens[ci].max())
    cap = torch.zeros(len(ci), L, words.shape[1])
    for k, c in enumerate(ci):
        cap[k, :lens[c]] = words[off[c]:off[c] + lens[c]].cpu()
    want = O.xattn_score(img[ri].cpu(), cap, [int(lens[c]) for c in ci], xa, 'clipped_l2norm', 'LogSumExp',
                         kw['lambda_lse'], kw['lambda_softmax'])
    got = S[ri][:, ci].cpu()
    assert float((got - want).abs().max()) <= 2e-5        # fp32 tolerance of the small-size parity tests

    # -- shard invariance: a row block scored alone (row0 a multiple of the 4-image workgroup) is bit-identical
    r0, r1 = (n_img // 8) * 3 // 4 * 4, (n_img // 8) * 3 // 4 * 4 + n_img // 8
    Sb = ops.scan_xattn_scores(img[r0:r1].contiguous(), words, plan, **kw)
    assert torch.equal(Sb, S[r0:r1])

    # -- caption permutation: other packing order, same scores (summation order inside a caption is unchanged,
    #    the position of the caption inside its 64-column tile is not -> allow 1e-6)
    perm = rng.permutation(n_cap)
    lens_p = lens[perm]
    off_p = np.concatenate([[0], np.cumsum(lens_p)[:-1]])
    idx = torch.from_numpy(np.concatenate([np.arange(off[c], off[c] + lens[c]) for c in perm])).to(dev)
    words_p = words[idx]
    plan_p = ops.ScanPlan(off_p, lens_p, words_p.shape[0], dev)
    n_chk = min(n_img, 512)
    Sp = ops.scan_xattn_scores(img[:n_chk].contiguous(), words_p, plan_p, **kw)
    assert float((Sp - S[:n_chk][:, torch.from_numpy(perm).to(dev)]).abs().max()) <= 1e-6


def test_ranker_full_size(dev):
    """Count ranker on a 5 000 x 25 000 score matrix: sampled rows / columns against a host argsort, plus the
    global checksum  sum_j t2i_rank[j] == #{(i, j): S[i, j] > S[gt(j), j]}  (ties broken towards lower index)."""
    n_img, n_cap = 5000, 25000
    g = torch.Generator(device=dev)
    g.manual_seed(3)
    S = torch.randn(n_img, n_cap, device=dev, generator=g)
    i_rank, i_top, t_rank, t_best, _ = ops.rank_counts(S)
    s_gt = ops.gather_gt(S)
    i_rank, i_top, t_rank = i_rank.cpu().numpy(), i_top.cpu().numpy(), t_rank.cpu().numpy()
    t_top = (t_best & 0xffffffff).cpu().numpy()
    rng = np.random.RandomState(0)
    for i in rng.choice(n_img, 40, replace=False):
        row = S[i].cpu().numpy()
        order = np.argsort(row)[::-1]
        pos = np.empty(n_cap, np.int64)
        pos[order] = np.arange(n_cap)
        assert i_rank[i] == pos[5 * i:5 * i + 5].min()      # evaluation.py:172-178
        assert i_top[i] == order[0]
    for j in rng.choice(n_cap, 40, replace=False):
        col = S[:, j].cpu().numpy()
        order = np.argsort(col)[::-1]
        assert t_rank[j] == int(np.where(order == j // 5)[0][0])   # evaluation.py:208-212
        assert t_top[j] == order[0]
    gt = S[torch.arange(n_cap, device=dev) // 5, torch.arange(n_cap, device=dev)]
    assert torch.equal(gt, s_gt)
    greater = int((S > gt[None, :]).sum().item())            # continuous random scores: no exact ties
    assert int(t_rank.astype(np.int64).sum()) == greater
    assert ops.recall_from_ranks(i_rank)[:3] == tuple(100.0 * int((i_rank < k).sum()) / n_img for k in (1, 5, 10))


def test_scan_repeated_launches_are_bit_identical(dev):
    """Stress for the hand-counted loads of the main loop (two workgroups per CU, 8 launches back to back): a load that
    lands in a register after it was reused shows up as a launch-to-launch difference or a non-finite score."""
    img, words, lens, off = _problem(1000, 23, dev)
    plan = ops.ScanPlan(off, lens, words.shape[0], dev)
    ws = ops.scan_prepare(img, words, plan)
    ref = ops.scan_xattn_scores(img, words, plan, workspace=ws).clone()
    assert bool(torch.isfinite(ref).all())
    for _ in range(8):
        again = ops.scan_xattn_scores(img, words, plan, workspace=ws)
        assert torch.equal(again, ref)


@pytest.mark.parametrize("mod,n_img", [('SAF', 1000), ('SGR', 1000), ('SAF', 5000), ('SGR', 5000)])
def test_sgraf_full_size(dev, mod, n_img):
    """BASELINE config 5 shape (embed 1024, sim_dim 256, sgr_step 3) on the f30k-size problem 1k x 5k AND at the size the
    config states, 5 000 x 25 000 (325 000 words, 5 100 column tiles, 313 blocks of 16 images): oracle spot check on
    24 scattered images x 40 scattered captions (a pair's score depends on that pair only: BatchNorms run on running
    statistics), and row-shard invariance (the multi-GPU contract of SURVEY 8e) -- Fusionmodule.py:406-451."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "helpers"))
    import sgraf_weights
    D, S_ = 1024, 256
    img, words, lens, off = _problem(n_img, 17, dev, D)
    words = ops.l2norm(words)                               # SGRAF's text tower normalises the word vectors
    n_cap = len(lens)
    w = sgraf_weights.make(D, S_)
    wd = {k: v.to(dev) for k, v in w.items()}
    plan = ops.ScanPlan(off, lens, words.shape[0], dev)
    S = ops.sgraf_scores(img, words, plan, wd, mod, 3)
    assert S.shape == (n_img, n_cap) and bool(torch.isfinite(S).all())
    assert float(S.min()) > 0.0 and float(S.max()) < 1.0   # sigmoid outputs
    rng = np.random.RandomState(7)
    ri = np.sort(rng.choice(n_img, 24, replace=False))
    ci = np.sort(rng.choice(n_cap, 40, replace=False))
    L = int(lens[ci].max())
    cap = torch.zeros(len(ci), L, D)
    for k, c in enumerate(ci):
        cap[k, :lens[c]] = words[off[c]:off[c] + lens[c]].cpu()
    want = O.sgraf_similarity(w, img[ri].cpu(), cap, [int(lens[c]) for c in ci], mod, 3)
    got = S[ri][:, ci].cpu()
    assert float((got - want).abs().max()) <= 5e-6         # the tolerance of the small-size SGRAF parity tests
    # row block scored alone (a multiple of the 16-image block): identical rows
    r0, r1 = 112, 240
    Sb = ops.sgraf_scores(img[r0:r1].contiguous(), words, plan, wd, mod, 3)
    assert torch.equal(Sb, S[r0:r1])


@pytest.mark.parametrize("kind", ["SAEM", "CAMERA"])
def test_pooled_bert_models_full_size(dev, kind):
    """BASELINE config 4 at MS-COCO size: SAEM (BERT-base + cnn head, D = 256, pdist_cos) and CAMERA (BERT-base + AGSA,
    12 views x 2048, MultiViewMatching) through the sharded evaluator (evalpipe.PooledModelEval) on 5 000 images x
    25 000 captions -- Models.py:600-645, Fusionmodule.py:674-692, Objectives.py:310-323.  Checked: the towers + scorer
    against the CPU oracle on the first 6 images / 30 captions (the 12-layer BERT stack included), scattered rows /
    columns of the rank vectors against a host argsort of the same matrix, row-block invariance."""
    import os
    import bench
    from itr_amd import config as C, evalpipe
    from itr_amd.modalmodule import get_model
    n_img, n_cap = 5000, 25000
    cfg_file, ckpt, trans = bench.bert_files(os.path.join("/tmp", "itr_bench_bert"))
    cfg = C.build_config(['with', kind, 'data_name=coco_precomp'])
    cfg.update(bert_config_file=cfg_file, init_checkpoint=ckpt, trans_cfg=trans, vocab_size=30522)
    torch.manual_seed(0)
    model = get_model(cfg)
    for m in model.modules():
        if isinstance(m, torch.nn.BatchNorm1d):
            m.running_mean.normal_(0, 0.1)
            m.running_var.uniform_(0.5, 1.5)
    model.val_start()
    feats, boxes, imgs_wh, ids, mask, types, lengths = bench.pooled_inputs(n_img, n_cap, kind, dev)
    pe = evalpipe.PooledModelEval(model, evalpipe.Comm(), batch=1024)
    lens = [int(x) for x in lengths]
    S, ranks = pe.eval(feats, boxes, imgs_wh, ids, mask, types, lens, n_img, n_cap)
    assert S.shape == (n_img, n_cap) and bool(torch.isfinite(S).all())
    # ---- oracle: towers + scorer on a corner of the problem
    ns, ncs = 6, 30
    wi = {k: v.detach().cpu() for k, v in model.img_enc.state_dict().items() if "num_batches_tracked" not in k}
    wt = {k: v.detach().cpu() for k, v in model.txt_enc.state_dict().items() if "num_batches_tracked" not in k}
    with torch.no_grad():
        if kind == "CAMERA":
            img_o, _ = O.camera_image(wi, feats[:ns].cpu(), boxes[:ns].cpu(), imgs_wh[:ns].cpu(), cfg["head"])
            cap_o = O.camera_text(wt, ids[:ncs].cpu(), mask[:ncs].cpu(), types[:ncs].cpu(), 12, 12, cfg["head"])
            S_o = O.multi_view_matching(img_o, cap_o)
        else:
            img_o = O.saem_image(wi, feats[:ns].cpu(), 4)
            cap_o = O.saem_text(wt, cfg["txt_stru"], ids[:ncs].cpu(), mask[:ncs].cpu(), types[:ncs].cpu(), 12, 12, 4)
            S_o = O.pdist_cos(img_o, cap_o)
    assert float((S[:ns, :ncs].cpu() - S_o).abs().max()) <= 2e-5
    # ---- rank vectors vs a host argsort of the same matrix (evaluation.py:156-222)
    i_rank, i_top, t_rank, t_top = [np.asarray(r) for r in ranks]
    rng = np.random.RandomState(1)
    cidx, rows_idx = np.arange(n_cap), np.arange(n_img)
    for i in rng.choice(n_img, 24, replace=False):
        row = S[i].cpu().numpy()
        lo = min(int((row > row[g_]).sum()) for g_ in range(5 * i, 5 * i + 5))       # exact ties (random-init BERT: near-identical
        hi = min(int((row >= row[g_]).sum()) - 1 for g_ in range(5 * i, 5 * i + 5))  # captions): the rank lies in the tie band
        assert lo <= i_rank[i] <= hi
        # ... and inside the band the pinned tie rule (G12 / G22): among equal scores the higher index ranks first
        assert i_rank[i] == min(int((row > row[g_]).sum() + ((row == row[g_]) & (cidx > g_)).sum()) for g_ in range(5 * i, 5 * i + 5))
        assert i_top[i] == n_cap - 1 - int(np.argmax(row[::-1]))
    for j in rng.choice(n_cap, 24, replace=False):
        col = S[:, j].cpu().numpy()
        assert int((col > col[j // 5]).sum()) <= t_rank[j] <= int((col >= col[j // 5]).sum()) - 1
        assert t_rank[j] == int((col > col[j // 5]).sum() + ((col == col[j // 5]) & (rows_idx > j // 5)).sum())
        assert t_top[j] == n_img - 1 - int(np.argmax(col[::-1]))
    # ---- a row block of images encoded and scored alone gives the same rows
    r0, r1 = 1024, 1024 + 256
    img_b, _ = pe.encode(feats[r0:r1], None if boxes is None else boxes[r0:r1], None if imgs_wh is None else imgs_wh[r0:r1],
                         ids[:8], mask[:8], types[:8], lens[:8])
    img_all, cap_all = pe.encode(feats[r0:r1], None if boxes is None else boxes[r0:r1], None if imgs_wh is None else imgs_wh[r0:r1],
                                 ids[:2048], mask[:2048], types[:2048], lens[:2048])
    assert torch.equal(img_b, img_all)
    Sb = pe._score(img_all, cap_all)
    assert float((Sb - S[r0:r1, :2048]).abs().max()) <= 1e-6


def test_vsepp_full_size(dev):
    """BASELINE config 2 at its stated size (Flickr30k test fold: 1 000 images x 5 000 captions, bi-GRU text tower, embed 1024,
    mean-pooled regions): image projection + text tower + cosine scores against the CPU oracle on a 40-image x 200-caption
    corner, and the rank vectors against a host argsort of the GPU's matrix -- ImgEncoder.py:133-147, TextEncoder.py:38-70,
    Objectives.py:18-21, evaluation.py:156-222."""
    import bench
    n_img, n_cap, vocab = 1000, 5000, 8481
    wi, wt = bench.make_weights(vocab)
    wid = {k: v.to(dev) for k, v in wi.items()}
    wtd = {k: v.to(dev) for k, v in wt.items()}
    g = torch.Generator(device=dev)
    g.manual_seed(0)
    feats = ops.l2norm(torch.randn(n_img, 36, 2048, device=dev, generator=g))
    lengths, tokens = bench.make_captions(n_cap, vocab)
    toks, tok_off, lens_sorted, order = bench.shard_captions(lengths, tokens, 0, n_cap, dev)
    img = ops.proj_l2norm(ops.mean_mid(feats), wid['fc.weight'], wid['fc.bias'])
    cap_sorted = ops.gru_encode(toks, tok_off, lens_sorted, wtd, True, gather_last=True)
    cap = torch.empty_like(cap_sorted)
    cap[torch.from_numpy(np.ascontiguousarray(order)).to(dev)] = cap_sorted
    S = ops.cosine_scores(img, cap)
    assert S.shape == (n_img, n_cap) and bool(torch.isfinite(S).all())
    # ---- oracle on the corner (the captions of the corner are encoded by the oracle in ITS batch: a caption's embedding
    # depends on that caption only)
    ns, ncs = 40, 200
    lens = lengths[:ncs]
    order_s = np.argsort(-lens, kind="stable")
    ids_s = torch.zeros(ncs, int(lens.max()), dtype=torch.long)
    for r, i in enumerate(order_s):
        ids_s[r, :lens[i]] = torch.from_numpy(tokens[int(i)])
    with torch.no_grad():
        img_o = O.encoder_image_precomp(feats[:ns].cpu().mean(1), wi["fc.weight"], wi["fc.bias"])
        cs, _ = O.encoder_text(ids_s, [int(lens[i]) for i in order_s], wt, True, False, False, "VSE++")
        cap_o = torch.zeros_like(cs)
        cap_o[torch.as_tensor(order_s)] = cs
        S_o = O.cosine_sim(img_o, cap_o)
    assert float((img[:ns].cpu() - img_o).abs().max()) <= 2e-6
    assert float((cap[:ncs].cpu() - cap_o).abs().max()) <= 5e-6
    assert float((S[:ns, :ncs].cpu() - S_o).abs().max()) <= 5e-6
    # ---- ranks of the whole fold vs a host argsort of the same matrix (continuous scores: no exact ties)
    i_rank, i_top, t_rank, t_best, _ = ops.rank_counts(S)
    Sh = S.cpu().numpy()
    i_rank, i_top, t_rank = i_rank.cpu().numpy(), i_top.cpu().numpy(), t_rank.cpu().numpy()
    t_top = (t_best & 0xffffffff).cpu().numpy()
    want = O.rank_counts(Sh)
    for got, ref in zip((i_rank, i_top, t_rank, t_top), want):
        assert (np.asarray(got) == np.asarray(ref)).all()
    for i in range(0, n_img, 97):
        order_r = np.argsort(Sh[i])[::-1]
        pos = np.empty(n_cap, np.int64)
        pos[order_r] = np.arange(n_cap)
        assert i_rank[i] == pos[5 * i:5 * i + 5].min() and i_top[i] == order_r[0]
