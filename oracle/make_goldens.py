#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY -- generate tests/golden/*.npz from the *reference itself*.

Runs only in the build container (needs /root/reference): imports the reference under
`oracle/ref_shim.py`, feeds it seeded synthetic inputs + seeded weights, and stores
inputs, weights and the reference's outputs as small .npz fixtures (data only -- no
reference source travels).  While generating, every fixture is also cross-checked against
the CPU restatement in `oracle/itr_oracle.py`; a mismatch aborts.

    python oracle/make_goldens.py            # regenerate everything
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
GOLD = os.path.join(ROOT, "tests", "golden")

import ref_shim  # noqa: E402

if not ref_shim.reference_available():
    print("reference tree absent -- nothing to do")
    sys.exit(0)

Objectives, ImgEncoder, TextEncoder, Fusionmodule, Models, evaluation, mutils = \
    ref_shim.import_reference()
import itr_oracle as O  # noqa: E402

torch.set_grad_enabled(False)


def npd(d):
    return {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in d.items()}


def save(name, **arrays):
    path = os.path.join(GOLD, name + ".npz")
    np.savez_compressed(path, **npd(arrays))
    print("wrote %-28s %8.1f KB" % (name + ".npz", os.path.getsize(path) / 1024))


def check(name, got, want, tol=1e-5):
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    err = float(np.abs(got - want).max()) if got.size else 0.0
    assert got.shape == want.shape, (name, got.shape, want.shape)
    assert err <= tol, "oracle != reference for %s: max abs err %.3e" % (name, err)
    print("   oracle==reference %-34s max|d|=%.2e" % (name, err))


def ragged_lengths(rng, n, lo, hi):
    """descending lengths like collate_fn (data_loader.py:146)."""
    return sorted([int(x) for x in rng.randint(lo, hi + 1, size=n)], reverse=True)


def sd(module, prefix=''):
    return {prefix + k: v.clone() for k, v in module.state_dict().items()}


# ------------------------------------------------------------------ G1 norms
def g1():
    g = torch.Generator().manual_seed(1)
    x = torch.randn(7, 5, 33, generator=g)
    x[2, 3] = 0.0  # zero row: eps keeps it finite
    out = dict(x=x, l2_last=mutils.l2norm(x, dim=-1), l2_dim1=mutils.l2norm(x, dim=1),
               l1_last=mutils.l1norm(x, dim=-1), l1_dim2=mutils.l1norm(x, dim=2))
    check('l2norm', O.l2norm(x, -1), out['l2_last'], 0)
    check('l1norm', O.l1norm(x, -1), out['l1_last'], 0)
    save('g1_norms', **out)


# ------------------------------------------------------------------ G2 image tower
def g2():
    torch.manual_seed(2)
    enc = ImgEncoder.EncoderImagePrecomp(256, 96, no_imgnorm=False).eval()
    enc.fc.bias.data.uniform_(-0.05, 0.05)
    x = mutils.l2norm(torch.randn(6, 36, 256), dim=-1)
    y3 = enc(x)
    y2 = enc(x.mean(1))                    # VSE++ build decision: pooled 2-D input
    enc_nn = ImgEncoder.EncoderImagePrecomp(256, 96, no_imgnorm=True, use_abs=True).eval()
    enc_nn.load_state_dict(enc.state_dict())
    y_abs = enc_nn(x)
    w, b = enc.fc.weight, enc.fc.bias
    check('img_precomp_3d', O.encoder_image_precomp(x, w, b), y3)
    check('img_precomp_2d', O.encoder_image_precomp(x.mean(1), w, b), y2)
    check('img_precomp_abs', O.encoder_image_precomp(x, w, b, True, True), y_abs)
    save('g2_img_precomp', images=x, fc_weight=w, fc_bias=b, out_3d=y3, out_2d=y2, out_nonorm_abs=y_abs)


# ------------------------------------------------------------------ G3 text tower
def g3():
    rng = np.random.RandomState(3)
    V, E, D, B = 60, 20, 48, 9
    lengths = ragged_lengths(rng, B, 1, 11)
    L = max(lengths)
    ids = torch.zeros(B, L, dtype=torch.long)
    for b, l in enumerate(lengths):
        ids[b, :l] = torch.from_numpy(rng.randint(4, V, size=l))
    out = dict(ids=ids, lengths=np.array(lengths))
    for bi in (False, True):
        for method in (None, 'VSE++'):
            for no_norm in (False, True):
                torch.manual_seed(30 + bi)
                enc = TextEncoder.EncoderText(V, E, D, 1, use_bi_gru=bi, no_txtnorm=no_norm,
                                              method_name=method).eval()
                cap, cap_len = enc(ids, lengths)
                tag = "%s_%s_%s" % ('bi' if bi else 'uni', 'last' if method else 'seq',
                                    'raw' if no_norm else 'l2')
                w = sd(enc)
                got, _ = O.encoder_text(ids, lengths, w, bi, no_norm, False, method)
                check('text_' + tag, got, cap, 2e-6)
                out['out_' + tag] = cap
                for k, v in w.items():
                    out["w_%s_%s" % ('bi' if bi else 'uni', k)] = v
                assert [int(x) for x in cap_len] == lengths
    save('g3_text_gru', **out)


# ------------------------------------------------------------------ G4 cosine + hinge (cfg 1 shape)
def g4():
    torch.manual_seed(4)
    B, D = 128, 1024
    im = mutils.l2norm(torch.randn(B, D), dim=-1)
    s = mutils.l2norm(torch.randn(B, D) + 0.35 * im, dim=-1)   # some signal so hinges are mixed
    scores = Objectives.cosine_sim(im, s)
    check('cosine_sim', O.cosine_sim(im, s), scores, 1e-6)
    out = dict(im=im, s=s, scores=scores)
    cfg = {'name': 'VSE++'}
    for mv in (False, True):
        crit = Objectives.ContrastiveLoss(cfg, margin=0.2, measure='cosine', max_violation=mv)
        with torch.enable_grad():
            im_g = im.clone().requires_grad_(True)
            s_g = s.clone().requires_grad_(True)
            sc = Objectives.cosine_sim(im_g, s_g)
            sc.retain_grad()
            loss = crit(im_g, s_g)
            # loss on an explicit score leaf, to capture dL/dS
            sleaf = scores.clone().requires_grad_(True)
            trip = Objectives.TripletLoss(margin=0.2, max_violation=mv)
            l2 = trip(sleaf)
            l2.backward()
            loss.backward()
        tag = 'maxviol' if mv else 'sum'
        ol, og = O.hinge_loss_and_grad(scores, 0.2, mv)
        check('hinge_' + tag, ol, l2, 1e-4)
        check('hinge_grad_' + tag, og, sleaf.grad, 0)
        assert abs(float(loss) - float(l2)) < 1e-4
        out['loss_' + tag] = loss
        out['dscores_' + tag] = sleaf.grad
        out['dim_' + tag] = im_g.grad
        out['ds_' + tag] = s_g.grad
    save('g4_cosine_hinge', **out)


# ------------------------------------------------------------------ G5 SCAN cross attention
def g5():
    rng = np.random.RandomState(5)
    torch.manual_seed(5)
    Ni, Nc, R, D = 8, 12, 36, 64
    lens = ragged_lengths(rng, Nc, 2, 9)
    L = max(lens)
    img = mutils.l2norm(torch.randn(Ni, R, D), dim=-1)
    cap = torch.randn(Nc, L, D) * 0.7
    for c, l in enumerate(lens):
        cap[c, l:] = 0
    out = dict(images=img, captions=cap, cap_lens=np.array(lens))
    base = dict(name='SCAN', lambda_lse=6.0, lambda_softmax=9.0)
    for xa, fn in (('t2i', Objectives.xattn_score_t2i), ('i2t', Objectives.xattn_score_i2t)):
        for agg in ('LogSumExp', 'Mean', 'Max', 'Sum'):
            for norm in ('clipped_l2norm', 'l2norm', 'softmax', 'no_norm', 'clipped'):
                cfg = dict(base, cross_attn=xa, agg_func=agg, raw_feature_norm=norm)
                import warnings
                with warnings.catch_warnings():
                    warnings.simplefilter('ignore')
                    ref = fn(img, cap, lens, cfg)
                got = O.xattn_score(img, cap, lens, xa, norm, agg, 6.0, 9.0)
                check('xattn_%s_%s_%s' % (xa, agg, norm), got, ref, 2e-5)
                out['sim_%s_%s_%s' % (xa, agg, norm)] = ref
    save('g5_scan_xattn', **out)


# ------------------------------------------------------------------ G6 SGRAF similarity
def g6():
    rng = np.random.RandomState(6)
    Ni, Nc, R, D, S = 8, 12, 36, 64, 32
    lens = ragged_lengths(rng, Nc, 2, 9)
    L = max(lens)
    torch.manual_seed(6)
    img = mutils.l2norm(torch.randn(Ni, R, D), dim=-1)
    cap = mutils.l2norm(torch.randn(Nc, L, D), dim=-1)
    for c, l in enumerate(lens):
        cap[c, l:] = 0
    out = dict(images=img, captions=cap, cap_lens=np.array(lens))
    for mod in ('SAF', 'SGR'):
        torch.manual_seed(60)
        enc = Fusionmodule.EncoderSimilarity(D, S, mod, 3)
        # non-trivial BN running stats / affine + non-zero biases
        for m in enc.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.running_mean.normal_(0, 0.1)
                m.running_var.uniform_(0.5, 1.5)
                m.weight.data.uniform_(0.8, 1.2)
                m.bias.data.normal_(0, 0.05)
            if isinstance(m, torch.nn.Linear):
                m.bias.data.normal_(0, 0.02)
        enc.eval()
        ref = enc(img, cap, lens)
        w = {k: v for k, v in sd(enc).items() if 'num_batches_tracked' not in k}
        got = O.sgraf_similarity(w, img, cap, lens, mod, 3)
        check('sgraf_' + mod, got, ref, 2e-6)
        out['sim_' + mod] = ref
        for k, v in w.items():
            out['w_%s_%s' % (mod, k)] = v
    save('g6_sgraf', **out)


# ------------------------------------------------------------------ G7/G8 MVM, pdist_cos
def g78():
    torch.manual_seed(7)
    imgs = torch.nn.functional.normalize(torch.randn(10, 12, 40), dim=-1)
    caps_sq = torch.nn.functional.normalize(torch.randn(10, 40), dim=-1)
    caps_ns = torch.nn.functional.normalize(torch.randn(23, 40), dim=-1)
    mvm = Fusionmodule.MultiViewMatching()
    r_sq, r_ns = mvm(imgs, caps_sq), mvm(imgs, caps_ns)
    check('mvm_square', O.multi_view_matching(imgs, caps_sq), r_sq, 1e-6)
    check('mvm_nonsquare', O.multi_view_matching(imgs, caps_ns), r_ns, 1e-6)
    x1 = torch.randn(9, 24)
    x2 = torch.randn(14, 24)
    x2[3] = 0  # zero row -> NaN -> 0
    pc = Objectives.pdist_cos(x1, x2)
    check('pdist_cos', O.pdist_cos(x1, x2), pc, 1e-6)
    save('g7_mvm_pdist', imgs=imgs, caps_sq=caps_sq, caps_ns=caps_ns, mvm_sq=r_sq, mvm_ns=r_ns,
         x1=x1, x2=x2, pdist_cos=pc)


# ------------------------------------------------------------------ G9 TripletLoss on non-cosine scores
def g9():
    torch.manual_seed(9)
    sc = torch.rand(33, 33)
    out = dict(scores=sc)
    for mv in (False, True):
        with torch.enable_grad():
            leaf = sc.clone().requires_grad_(True)
            l = Objectives.TripletLoss(margin=0.2, max_violation=mv)(leaf)
            l.backward()
        ol, og = O.hinge_loss_and_grad(sc, 0.2, mv)
        check('triplet_%d' % mv, ol, l, 1e-5)
        check('triplet_grad_%d' % mv, og, leaf.grad, 0)
        out['loss_%d' % mv] = l
        out['grad_%d' % mv] = leaf.grad
    save('g9_triplet', **out)


# ------------------------------------------------------------------ G11 eval harness (SCAN, Q1 quirk)
class _FakeLoader:
    """Minimal stand-in for the DataLoader: yields the collate_fn 8-tuple (data_loader.py:178)
    with `ids` listified."""

    def __init__(self, images, ids_tok, lengths, batch):
        self.dataset = list(range(len(lengths)))
        self.batches = []
        n = len(lengths)
        for b0 in range(0, n, batch):
            idx = list(range(b0, min(b0 + batch, n)))
            idx.sort(key=lambda i: -lengths[i])
            lens = [lengths[i] for i in idx]
            tok = torch.zeros(len(idx), max(lens), dtype=torch.long)
            for r, i in enumerate(idx):
                tok[r, :lengths[i]] = ids_tok[i][:lengths[i]]
            self.batches.append((images[idx], None, None, tok, lens, idx, None, None))

    def __iter__(self):
        return iter(self.batches)


def g11():
    rng = np.random.RandomState(11)
    torch.manual_seed(11)
    NI, F_, D, E, V = 40, 48, 32, 16, 50
    n = NI * 5
    feats = mutils.l2norm(torch.randn(NI, 36, F_), dim=-1)
    images = feats.repeat_interleave(5, 0)                  # each image stored once per caption
    lengths = [int(x) for x in rng.randint(3, 10, size=n)]
    ids_tok = [torch.from_numpy(rng.randint(4, V, size=12)) for _ in range(n)]
    cfg = dict(name='SCAN', grad_clip=2.0, img_dim=F_, embed_size=D, precomp_enc_type='basic',
               no_imgnorm=False, vocab_size=V, word_dim=E, num_layers=1, bi_gru=True,
               no_txtnorm=True, margin=0.2, measure='cosine', max_violation=False,
               learning_rate=2e-4, cross_attn='t2i', raw_feature_norm='clipped_l2norm',
               agg_func='LogSumExp', lambda_lse=6.0, lambda_softmax=9.0, data_name='coco_precomp')
    import io
    import contextlib
    with contextlib.redirect_stdout(io.StringIO()):
        model = Models.SCAN(cfg)
    loader = _FakeLoader(images, ids_tok, lengths, 32)
    img_embs, cap_embs, cap_lens = evaluation.encode_data(model, loader, islength=True)
    img_u = img_embs[::5]
    with contextlib.redirect_stdout(io.StringIO()):
        sims_q1 = evaluation.cal_sims(model, img_u, cap_embs, cap_lens, shard_size=100)
        sims_ok = evaluation.cal_sims(model, img_u, cap_embs, cap_lens, shard_size=10 ** 9)
    (ri, (ranks_i, top_i)) = evaluation.i2t(sims_ok, True)
    (rt, (ranks_t, top_t)) = evaluation.t2i(sims_ok, True)

    # oracle cross-check
    wi = sd(model.img_enc)
    wt = sd(model.txt_enc)
    o_img = O.encoder_image_precomp(feats, wi['fc.weight'], wi['fc.bias'])
    check('harness_img_embs', o_img, img_u, 2e-6)
    fn = lambda a, b, l: O.xattn_score(a, b, l, 't2i', 'clipped_l2norm', 'LogSumExp', 6.0, 9.0)
    check('harness_sims_correct', O.cal_sims(fn, img_u, cap_embs, cap_lens, 10 ** 9), sims_ok, 2e-5)
    check('harness_sims_q1', O.cal_sims(fn, img_u, cap_embs, cap_lens, 100, True), sims_q1, 2e-5)
    check('harness_sims_sliced', O.cal_sims(fn, img_u, cap_embs, cap_lens, 100, False), sims_ok, 2e-5)
    oi = O.i2t_argsort(sims_ok, True)
    ot = O.t2i_argsort(sims_ok, True)
    assert oi[0] == ri and ot[0] == rt
    assert (oi[1][0] == ranks_i).all() and (ot[1][0] == ranks_t).all()
    c = O.rank_counts(sims_ok)
    assert (c[0] == ranks_i).all() and (c[2] == ranks_t).all()
    assert (c[1] == top_i).all() and (c[3] == top_t).all()
    print("   oracle==reference harness ranks/top1 exact")
    out = dict(features=feats, lengths=np.array(lengths), token_ids=torch.stack(ids_tok),
               img_embs=img_u, cap_embs=cap_embs, cap_lens=cap_lens,
               sims_q1_shard100=sims_q1.astype(np.float32), sims=sims_ok.astype(np.float32),
               i2t=np.array(ri), t2i=np.array(rt), i2t_ranks=ranks_i, i2t_top1=top_i,
               t2i_ranks=ranks_t, t2i_top1=top_t)
    for k, v in wi.items():
        out['wimg_' + k] = v
    for k, v in wt.items():
        out['wtxt_' + k] = v
    save('g11_harness_scan', **out)


# ------------------------------------------------------------------ G12 ranker only (+ ties)
def g12():
    rng = np.random.RandomState(12)
    sims = rng.randn(100, 500)
    (ri, (ranks_i, top_i)) = evaluation.i2t(sims, True)
    (rt, (ranks_t, top_t)) = evaluation.t2i(sims, True)
    c = O.rank_counts(sims)
    assert (c[0] == ranks_i).all() and (c[2] == ranks_t).all()
    assert (c[1] == top_i).all() and (c[3] == top_t).all()
    # tie case: quantised scores -> many exact ties
    tie = np.round(rng.randn(30, 150) * 2) / 2
    (tri, (tranks_i, ttop_i)) = evaluation.i2t(tie, True)
    (trt, (tranks_t, ttop_t)) = evaluation.t2i(tie, True)
    ct = O.rank_counts(tie)
    n_i = int((ct[0] != tranks_i).sum())
    n_t = int((ct[2] != tranks_t).sum())
    print("   tie case: rank_counts vs reference argsort: %d/%d i2t, %d/%d t2i rows differ "
          "(numpy argsort is unstable; see SURVEY Q8)" % (n_i, len(tranks_i), n_t, len(tranks_t)))
    zeros = np.zeros((6, 30))
    (_, (zr_i, zt_i)) = evaluation.i2t(zeros, True)
    (_, (zr_t, zt_t)) = evaluation.t2i(zeros, True)
    cz = O.rank_counts(zeros)
    assert (cz[0] == zr_i).all() and (cz[2] == zr_t).all(), "all-zeros tie rule"
    save('g12_ranker', sims=sims, i2t=np.array(ri), t2i=np.array(rt), i2t_ranks=ranks_i,
         i2t_top1=top_i, t2i_ranks=ranks_t, t2i_top1=top_t,
         tie_sims=tie, tie_i2t_ranks=tranks_i, tie_t2i_ranks=tranks_t, tie_i2t=np.array(tri),
         tie_t2i=np.array(trt), zeros_i2t_ranks=zr_i, zeros_t2i_ranks=zr_t)


# ------------------------------------------------------------------ G22 ranker on float64 (ensemble) matrices
sys.path.insert(0, os.path.join(ROOT, 'tests', 'helpers'))
from rank_matrices import ensemble_sigmoid_matrix, half_ulp_matrix  # noqa: E402  (one recipe for generator and tests)


def g22():
    import rank_matrices
    out = {}
    # seeds 35 / 26: the fp32 cast of these two changes one i2t / one t2i rank (most seeds change none at this size)
    for name, make in rank_matrices.CASES.items():
        S = make()
        assert S.dtype == np.float64
        (ri, (ranks_i, top_i)) = evaluation.i2t(S, True)
        (rt, (ranks_t, top_t)) = evaluation.t2i(S, True)
        c = O.rank_counts(S)
        assert (c[0] == ranks_i).all() and (c[2] == ranks_t).all(), "oracle counts != reference argsort (float64)"
        assert (c[1] == top_i).all() and (c[3] == top_t).all()
        # how much an fp32 cast of the same matrix would change (documents that the fixture is sensitive)
        c32 = O.rank_counts(S.astype(np.float32))
        n_i, n_t = int((c32[0] != ranks_i).sum()), int((c32[2] != ranks_t).sum())
        print("   %s: fp32 truncation would change %d/%d i2t and %d/%d t2i ranks" % (name, n_i, len(ranks_i), n_t, len(ranks_t)))
        assert n_i + n_t > 0, "fixture does not separate float64 from float32 ranking"
        out.update({name + '_sha256': rank_matrices.sha256_u8(S),
                    name + '_shape': np.array(S.shape), name + '_i2t': np.array(ri), name + '_t2i': np.array(rt),
                    name + '_i2t_ranks': ranks_i.astype(np.int32), name + '_i2t_top1': top_i.astype(np.int32),
                    name + '_t2i_ranks': ranks_t.astype(np.int32), name + '_t2i_top1': top_t.astype(np.int32),
                    name + '_fp32_changed': np.array([n_i, n_t])})
    save('g22_ranker_f64', **out)


# ------------------------------------------------------------------ G10 BERT + SAEM towers
def g10():
    import json
    import tempfile
    from itr.modalmodule import bert as rbert
    rng = np.random.RandomState(10)
    out = {}

    def rand_init(m, seed):
        torch.manual_seed(seed)
        for p_ in m.parameters():
            p_.data.normal_(0, 0.05)
        for name, p_ in m.named_parameters():
            if name.endswith('gamma'):
                p_.data.uniform_(0.8, 1.2)

    # tiny BERT: 2 layers, hidden 64, 4 heads (dk 16)
    cfg_d = dict(vocab_size=100, hidden_size=64, num_hidden_layers=2, num_attention_heads=4, intermediate_size=128,
                 max_position_embeddings=40, type_vocab_size=2, hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1)
    cfg = rbert.BertConfig.from_dict(cfg_d)
    model = rbert.BertModel(cfg).eval()
    rand_init(model, 100)
    B, L = 5, 12
    lens = [12, 9, 7, 4, 2]
    ids = torch.from_numpy(rng.randint(1, 100, size=(B, L)))
    mask = torch.zeros(B, L, dtype=torch.long)
    for b, l in enumerate(lens):
        mask[b, :l] = 1
        ids[b, l:] = 0
    types = torch.zeros(B, L, dtype=torch.long)
    types[:, 6:] = 1
    layers, pooled = model(ids, types, mask)
    w = sd(model)
    ol, op = O.bert_model(w, ids, types, mask, 2, 4)
    for n in range(2):
        check('bert_layer%d' % n, ol[n], layers[n], 2e-5)
    check('bert_pooled', op, pooled, 2e-5)
    out.update(bert_ids=ids, bert_types=types, bert_mask=mask, bert_layer0=layers[0], bert_layer1=layers[1],
               bert_pooled=pooled, bert_cfg=json.dumps(cfg_d))
    for k, v in w.items():
        out['wbert_' + k] = v

    # one wider layer: hidden 256, 4 heads (dk 64 like BERT-base), intermediate 512
    cfg2_d = dict(vocab_size=10, hidden_size=256, num_hidden_layers=1, num_attention_heads=4, intermediate_size=512,
                  max_position_embeddings=40, type_vocab_size=2)
    lay = rbert.BERTLayer(rbert.BertConfig.from_dict(cfg2_d)).eval()
    rand_init(lay, 101)
    x = torch.randn(3, 36, 256)
    m2 = torch.ones(3, 36)
    m2[1, 30:] = 0
    ext = ((1.0 - m2) * -10000.0)[:, None, None, :]
    y = lay(x, ext)
    w2 = sd(lay, 'layer.')
    check('bert_wide_layer', O.bert_layer(w2, 'layer.', x, m2, 4), y, 2e-5)
    out.update(wide_x=x, wide_mask=m2, wide_y=y, wide_cfg=json.dumps(cfg2_d))
    for k, v in w2.items():
        out['wwide_' + k] = v

    # SAEM heads on the tiny BERT
    tmp = tempfile.mkdtemp()
    json.dump(cfg_d, open(os.path.join(tmp, 'bert_config.json'), 'w'))
    torch.save(model.state_dict(), os.path.join(tmp, 'pytorch_model.bin'))
    trans_d = dict(vocab_size=10, hidden_size=64, num_hidden_layers=1, num_attention_heads=4, intermediate_size=128,
                   max_position_embeddings=40, type_vocab_size=2)
    json.dump(trans_d, open(os.path.join(tmp, 'trans_cfg.json'), 'w'))
    out['trans_cfg'] = json.dumps(trans_d)
    for stru in ('cnn', 'pooling', 'trans'):
        scfg = dict(bert_config_file=os.path.join(tmp, 'bert_config.json'), init_checkpoint=os.path.join(tmp, 'pytorch_model.bin'),
                    txt_stru=stru, final_dims=64, trans_cfg=os.path.join(tmp, 'trans_cfg.json'), bi_gru=False, embed_size=64,
                    num_layers=1)
        torch.manual_seed(110)
        tm = TextEncoder.BertMapping(scfg).eval()
        for name, p_ in tm.named_parameters():
            if not name.startswith('bert.'):
                p_.data.normal_(0, 0.05)
        code = tm(ids, mask, types, lens)
        wt = sd(tm)
        check('saem_text_' + stru, O.saem_text(wt, stru, ids, mask, types, 2, 4, 4), code, 2e-5)
        out['saem_text_' + stru] = code
        for k, v in wt.items():
            if not k.startswith('bert.'):
                out['wsaem_%s_%s' % (stru, k)] = v
    icfg = dict(trans_cfg=os.path.join(tmp, 'trans_cfg.json'), img_dim=96, final_dims=64)
    torch.manual_seed(111)
    im = ImgEncoder.TransformerMapping(icfg).eval()
    for p_ in im.parameters():
        p_.data.normal_(0, 0.05)
    xi = mutils.l2norm(torch.randn(4, 36, 96), dim=-1)
    yi = im(xi)
    wi = sd(im)
    check('saem_image', O.saem_image(wi, xi, 4), yi, 2e-5)
    out.update(saem_img_x=xi, saem_img_y=yi)
    for k, v in wi.items():
        out['wsaemimg_' + k] = v
    save('g10_bert_saem', **out)


# ------------------------------------------------------------------ G13 CAMERA towers
def g13():
    import json
    import tempfile
    from itr.modalmodule import bert as rbert
    rng = np.random.RandomState(13)
    cfg_d = dict(vocab_size=100, hidden_size=64, num_hidden_layers=2, num_attention_heads=4, intermediate_size=128,
                 max_position_embeddings=40, type_vocab_size=2, hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1)
    bm = rbert.BertModel(rbert.BertConfig.from_dict(cfg_d)).eval()
    torch.manual_seed(130)
    for p_ in bm.parameters():
        p_.data.normal_(0, 0.05)
    tmp = tempfile.mkdtemp()
    json.dump(cfg_d, open(os.path.join(tmp, 'bert_config.json'), 'w'))
    torch.save(bm.state_dict(), os.path.join(tmp, 'pytorch_model.bin'))

    def randomize(m, seed):
        torch.manual_seed(seed)
        for name, p_ in m.named_parameters():
            if not name.startswith('bert.'):
                p_.data.normal_(0, 0.08)
        for mod in m.modules():
            if isinstance(mod, torch.nn.BatchNorm1d):
                mod.running_mean.normal_(0, 0.1)
                mod.running_var.uniform_(0.5, 1.5)
                mod.weight.data.uniform_(0.8, 1.2)
                mod.bias.data.normal_(0, 0.05)

    D, H, K = 64, 4, 12
    # image tower
    ie = ImgEncoder.EncoderImagePrecompSelfAttn(96, D, H, K, drop=0.0)
    randomize(ie, 131)
    ie.eval()
    B, R = 5, 36
    images = mutils.l2norm(torch.randn(B, R, 96), dim=-1)
    x1 = torch.from_numpy(rng.uniform(0, 400, size=(B, R))).float()
    y1 = torch.from_numpy(rng.uniform(0, 300, size=(B, R))).float()
    bw = torch.from_numpy(rng.uniform(20, 200, size=(B, R))).float()
    bh = torch.from_numpy(rng.uniform(20, 200, size=(B, R))).float()
    boxes = torch.stack([x1, y1, x1 + bw, y1 + bh], -1)
    wh = torch.tensor([[640., 480.]]).repeat(B, 1)
    img_emb, smry = ie(images, boxes, wh)
    wi = {k: v for k, v in sd(ie).items() if 'num_batches_tracked' not in k}
    oi, osm = O.camera_image(wi, images, boxes, wh, H)
    check('camera_img_emb', oi, img_emb, 2e-5)
    check('camera_smry', osm, smry, 2e-5)
    # text tower
    te = TextEncoder.CAMERAEncoderText(os.path.join(tmp, 'bert_config.json'), os.path.join(tmp, 'pytorch_model.bin'), D, H, drop=0.0)
    randomize(te, 132)
    te.eval()
    Bc, L = 7, 12
    lens = [12, 10, 9, 7, 5, 3, 2]
    ids = torch.from_numpy(rng.randint(1, 100, size=(Bc, L)))
    mask = torch.zeros(Bc, L, dtype=torch.long)
    for b, l in enumerate(lens):
        mask[b, :l] = 1
        ids[b, l:] = 0
    types = torch.zeros(Bc, L, dtype=torch.long)
    cap = te(ids, mask, types)
    wt = {k: v for k, v in sd(te).items() if 'num_batches_tracked' not in k}
    check('camera_text', O.camera_text(wt, ids, mask, types, 2, 4, H), cap, 2e-5)
    # similarity + losses
    mvm = Fusionmodule.MultiViewMatching()
    sim = mvm(img_emb, cap)
    check('camera_mvm', O.multi_view_matching(img_emb, cap), sim, 1e-6)
    torch.manual_seed(133)
    a_im = torch.nn.functional.normalize(torch.randn(9, 32), dim=1)
    a_s = torch.nn.functional.normalize(torch.randn(9, 32), dim=1)
    ang = Objectives.AngularLoss()(a_im, a_s, None, list(range(9)))
    check('angular_loss', O.angular_loss(a_im, a_s), ang, 1e-4)
    div = Objectives.DiversityRegularization(K, B)(smry)
    check('camera_divreg', O.diversity_regularization(smry), div, 1e-4)
    out = dict(images=images, boxes=boxes, imgs_wh=wh, img_emb=img_emb, smry_mat=smry, ids=ids, mask=mask, types=types,
               cap_emb=cap, sim=sim, div_reg=div, bert_cfg=json.dumps(cfg_d), ang_im=a_im, ang_s=a_s, ang_loss=ang)
    for k, v in wi.items():
        out['wimg_' + k] = v
    for k, v in wt.items():
        out['wtxt_' + k] = v
    save('g13_camera', **out)


# ------------------------------------------------------------------ G14 data layer
G14_CAPTIONS = [
    "A man riding a wave on top of a surfboard .", "a dog, running; through the GRASS!", "Two children play.",
    "the quick brown fox", "A woman in a red dress is standing near a café table with a zebra",
    "people", "An old man sits on a bench and reads a newspaper while pigeons gather around his feet",
    "a b c d e f g", "Snow-covered mountains behind a small wooden hut", "a cat sleeps on the sofa",
    "A man riding a horse", "the dog runs", "children play in the park near a fountain .", "a plate of food with broccoli",
    "Two dogs", "a skateboarder does a trick on a ramp", "the woman is reading", "A bus drives down the street , past a stop sign",
    "a man", "a giraffe eats leaves from a tall tree", "sheep graze", "a red dress", "A table with a laptop and a cup of coffee",
    "the park", "a horse and a dog run through the grass", "zebra", "A small hut", "the man reads a newspaper on a bench",
    "a child", "pigeons gather near the fountain in the park",
]
G14_BERT_VOCAB = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]", "a", "the", "man", "dog", "rid", "##ing", "##s", "run", "##ning",
                  "wave", "on", "top", "of", "surf", "##board", ".", ",", ";", "!", "-", "grass", "through", "two", "child",
                  "##ren", "play", "quick", "brown", "fox", "woman", "in", "red", "dress", "is", "stand", "near", "cafe",
                  "table", "with", "zebra", "people", "snow", "covered", "mountain", "behind", "small", "wood", "##en", "hut",
                  "cat", "sleep", "sofa", "un", "##aff", "##able", "'"]


def g14():
    import json
    import tempfile
    from itr.datamodule import data_loader as rdl, vocab as rvocab, tokenization as rtok
    rng = np.random.RandomState(14)
    root = tempfile.mkdtemp()
    name = 'toy_precomp'
    d = os.path.join(root, name)
    os.makedirs(d)
    n_img = 6
    ims = rng.randn(n_img, 36, 8).astype(np.float32)
    boxes = rng.uniform(0, 300, size=(n_img, 36, 4)).astype(np.float32)
    sizes = np.tile(np.array([[640., 480.]], np.float32), (n_img, 1))
    caps_blob = ("\n".join(G14_CAPTIONS) + "\n").encode('utf-8')
    for split in ('train', 'dev', 'test'):
        np.save(os.path.join(d, '%s_ims.npy' % split), ims)
        np.save(os.path.join(d, '%s_boxes.npy' % split), boxes)
        np.save(os.path.join(d, '%s_img_sizes.npy' % split), sizes)
        open(os.path.join(d, '%s_caps.txt' % split), 'wb').write(caps_blob)
    # -- vocabulary: the reference's build_vocab (threshold 2) + JSON round trip
    v = rvocab.build_vocab(root, name, caption_file={name: ['train_caps.txt']}, threshold=2)
    vdir = os.path.join(root, 'vocab')
    os.makedirs(vdir)
    rvocab.serialize_vocab(v, os.path.join(vdir, '%s_vocab.json' % name))
    vocab_json = open(os.path.join(vdir, '%s_vocab.json' % name)).read()
    out = dict(ims=ims, boxes=boxes, img_sizes=sizes, caps_blob=np.frombuffer(caps_blob, np.uint8),
               vocab_json=np.frombuffer(vocab_json.encode(), np.uint8), vocab_len=len(v),
               tokenizer_note="nltk absent: word_tokenize = regex  \\w+|[^\\w\\s]  (oracle/ref_shim.py)")
    # -- GRU dataset (the reference tokenises str(bytes): SURVEY Q6) + collate_fn
    cfg = {'use_bbox': False, 'text_encoder': 'gru', 'vocab_path': vdir, 'data_name': name, 'vocab_type': 'json', 'name': 'SCAN'}
    ds = rdl.PrecompDataset(d, 'test', cfg)
    out.update(test_len=len(ds), test_im_div=ds.im_div, dev_len=len(rdl.PrecompDataset(d, 'dev', cfg)))
    ids_all = [ds[i][3].numpy() for i in range(len(ds))]
    out['gru_ids_concat'] = np.concatenate(ids_all)
    out['gru_ids_len'] = np.array([len(x) for x in ids_all])
    out['item7_image'] = ds[7][0].numpy()
    out['item7_meta'] = np.array([ds[7][4], ds[7][5]])
    pick = [3, 17, 4, 29, 8, 0, 11, 5]
    batch = rdl.collate_fn([ds[i] for i in pick])
    out.update(pick=np.array(pick), col_images=batch[0], col_ids=batch[3], col_lengths=np.array(batch[4]), col_index=np.array(batch[5]))
    assert batch[1] == (None,) * len(pick) and batch[6] == (None,) * len(pick)
    # -- VSRN caption layout (data_loader.py:117-125): max_len small enough that some captions take the truncation branch
    dsv = rdl.PrecompDataset(d, 'test', dict(cfg, name='VSRN', max_len=9))
    out['vsrn_ids'] = np.stack([dsv[i][3].numpy() for i in range(len(dsv))]).astype(np.int64)
    out['vsrn_mask'] = np.stack([dsv[i][6].numpy() for i in range(len(dsv))])
    vb = rdl.collate_fn([dsv[i] for i in pick])
    out.update(vcol_ids=vb[3], vcol_lengths=np.array(vb[4]), vcol_mask=vb[6], vcol_index=np.array(vb[5]))
    # -- BERT tokenizer + features (+ bbox branch of collate_fn)
    vfile = os.path.join(root, 'bert_vocab.txt')
    open(vfile, 'w').write("\n".join(G14_BERT_VOCAB) + "\n")
    tk = rtok.FullTokenizer(vocab_file=vfile, do_lower_case=True)
    sentences = G14_CAPTIONS[:10] + ["unaffable", "x" * 101 + " dog", "café\tdog\x00s  run", "", "dog's", "running.surfboard"]
    toks = [tk.tokenize(sn) for sn in sentences]
    out['bert_vocab'] = np.frombuffer(("\n".join(G14_BERT_VOCAB) + "\n").encode(), np.uint8)
    out['bert_sentences'] = np.frombuffer(("\n".join(sentences)).encode('utf-8'), np.uint8)
    out['bert_tokens'] = np.frombuffer(("\n".join(" ".join(t) for t in toks)).encode('utf-8'), np.uint8)
    feats = [rdl.convert_to_feature(sn.encode('utf-8'), 12, tk) for sn in sentences]
    out['bert_input_ids'] = np.array([f[1] for f in feats])
    out['bert_input_mask'] = np.array([f[2] for f in feats])
    out['bert_type_ids'] = np.array([f[3] for f in feats])
    cfgb = {'use_bbox': True, 'text_encoder': 'bert', 'max_words': 12, 'vocab_file': vfile, 'data_name': name, 'name': 'CAMERA'}
    dsb = rdl.PrecompDataset(d, 'test', cfgb)
    bb = rdl.collate_fn([dsb[i] for i in pick])
    out.update(bcol_images=bb[0], bcol_boxes=bb[1], bcol_wh=bb[2], bcol_ids=bb[3], bcol_lengths=np.array([int(x) for x in bb[4]]),
               bcol_index=np.asarray(bb[5]), bcol_mask=bb[6], bcol_types=bb[7])
    save('g14_data_layer', **out)


# ------------------------------------------------------------------ G15 training step (a14 train_emb)
def _train_batch(rng, B, V, F_):
    lens = sorted([int(x) for x in rng.randint(2, 12, size=B)], reverse=True)
    ids = torch.zeros(B, max(lens), dtype=torch.long)
    for b, l in enumerate(lens):
        ids[b, :l] = torch.from_numpy(rng.randint(4, V, size=l))
    feats = mutils.l2norm(torch.randn(B, 36, F_), dim=-1)
    return feats, ids, lens


def g15():
    """The reference's own SCAN.train_emb (Models.py:198-225) run twice on CPU: loss, clipped gradients and the
    parameters after each Adam step; the oracle's restated step must agree.  VSE++ is non-functional in the
    reference, so its step is assembled from the reference components + torch's own
    clip_grad_norm_ / Adam."""
    rng = np.random.RandomState(15)
    torch.manual_seed(15)
    V, F_, D, E, B = 60, 24, 32, 16, 10
    cfg = dict(name='SCAN', img_dim=F_, embed_size=D, precomp_enc_type='basic', no_imgnorm=False, vocab_size=V, word_dim=E,
               num_layers=1, bi_gru=True, no_txtnorm=False, margin=0.2, measure='cosine', max_violation=True, learning_rate=2e-3,
               grad_clip=2.0, cross_attn='t2i', raw_feature_norm='clipped_l2norm', agg_func='LogSumExp', lambda_lse=6.0,
               lambda_softmax=9.0)
    out = {}
    with torch.enable_grad():
        model = Models.SCAN(cfg)
        model.train_start()
        model.logger = evaluation.LogCollector()
        w0_img, w0_txt = sd(model.img_enc), sd(model.txt_enc)
        for k, v in w0_img.items():
            out['w0_img_' + k] = v
        for k, v in w0_txt.items():
            out['w0_txt_' + k] = v
        state, wi, wt = None, w0_img, w0_txt
        for step in (1, 2):
            feats, ids, lens = _train_batch(rng, B, V, F_)
            batch = (feats, None, None, ids, lens, list(range(B)), None, None)
            model.train_emb(batch)
            loss = float(model.logger.meters['Loss'].val)
            out.update({'s%d_feats' % step: feats, 's%d_ids' % step: ids, 's%d_lens' % step: np.array(lens), 's%d_loss' % step: loss})
            grads = {('txt.' + n): p.grad.detach().clone() for n, p in model.txt_enc.named_parameters()}
            grads.update({('img.' + n): p.grad.detach().clone() for n, p in model.img_enc.named_parameters()})
            for n, gten in grads.items():
                out['s%d_grad_%s' % (step, n)] = gten
            for k, v in sd(model.img_enc).items():
                out['s%d_img_%s' % (step, k)] = v
            for k, v in sd(model.txt_enc).items():
                out['s%d_txt_%s' % (step, k)] = v
            # ---- oracle cross-check
            o_loss, o_grads, wi, wt, state = O.gru_model_train_step('SCAN', wi, wt, feats, ids, lens, cfg, state)
            check('train loss step %d' % step, o_loss, loss, 1e-5)
            for n in grads:
                check('grad %s (step %d)' % (n, step), o_grads[n], grads[n], 2e-6)
            for k, v in sd(model.img_enc).items():
                check('img %s after step %d' % (k, step), wi[k], v, 5e-5)   # Adam amplifies 1e-7 gradient noise where |g| ~ eps
            for k, v in sd(model.txt_enc).items():
                check('txt %s after step %d' % (k, step), wt[k], v, 5e-5)
    out['cfg_json'] = np.frombuffer(__import__('json').dumps(cfg).encode(), np.uint8)
    # ---- VSE++ from components (uni-GRU, sum of violations)
    torch.manual_seed(16)
    cfgv = dict(cfg, name='VSE++', bi_gru=False, max_violation=False, learning_rate=1e-3)
    with torch.enable_grad():
        ie = ImgEncoder.EncoderImagePrecomp(F_, D)
        te = TextEncoder.EncoderText(V, E, D, 1, use_bi_gru=False, method_name='VSE++')
        crit = Objectives.ContrastiveLoss(config=cfgv, margin=0.2, measure='cosine', max_violation=False)
        params = list(te.parameters()) + list(ie.fc.parameters())
        opt = torch.optim.Adam(params, lr=cfgv['learning_rate'])
        wi, wt = sd(ie), sd(te)
        for k, v in wi.items():
            out['v_w0_img_' + k] = v
        for k, v in wt.items():
            out['v_w0_txt_' + k] = v
        feats, ids, lens = _train_batch(rng, B, V, F_)
        opt.zero_grad()
        cap, _ = te(ids, lens)
        loss = crit(ie(feats.mean(1)), cap)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(params, 2.0)
        opt.step()
        out.update(v_feats=feats, v_ids=ids, v_lens=np.array(lens), v_loss=float(loss))
        o_loss, o_grads, nwi, nwt, _ = O.gru_model_train_step('VSE++', wi, wt, feats, ids, lens, cfgv, None)
        check('vse++ train loss', o_loss, float(loss), 1e-5)
        for k, v in sd(ie).items():
            out['v_s1_img_' + k] = v
            check('vse++ img %s' % k, nwi[k], v, 5e-5)
        for k, v in sd(te).items():
            out['v_s1_txt_' + k] = v
            check('vse++ txt %s' % k, nwt[k], v, 5e-5)
        for n, p_ in list(('txt.' + n, p_) for n, p_ in te.named_parameters()) + list(('img.' + n, p_) for n, p_ in ie.named_parameters()):
            out['v_grad_' + n] = p_.grad.detach().clone()
            check('vse++ grad ' + n, o_grads[n], p_.grad, 2e-6)
    save('g15_train_step', **out)


# ------------------------------------------------------------------ G16 VSRN image tower + model forward_emb
def _randomise_bn(module, gen_seed):
    g = torch.Generator().manual_seed(gen_seed)
    for m in module.modules():
        if isinstance(m, torch.nn.BatchNorm1d):
            m.weight.data = torch.rand(m.weight.shape, generator=g) * 0.6 + 0.2     # Rs_GCN initialises gamma = beta = 0
            m.bias.data = torch.randn(m.bias.shape, generator=g) * 0.05             # (an identity layer): trained-like values
            m.running_mean.data = torch.randn(m.running_mean.shape, generator=g) * 0.1
            m.running_var.data = torch.rand(m.running_var.shape, generator=g) + 0.5


def g16():
    out = {}
    for data_name in ('coco_precomp', 'f30k_precomp'):
        tag = data_name.split('_')[0]
        torch.manual_seed(160)
        enc = ImgEncoder.EncoderImagePrecompAttn(48, 64, data_name, use_abs=False, no_imgnorm=False).eval()
        enc.fc.bias.data.uniform_(-0.05, 0.05)
        _randomise_bn(enc, 161)
        x = mutils.l2norm(torch.randn(5, 36, 48), dim=-1)
        feat, gcn = enc(x)
        w = sd(enc)
        of, og = O.vsrn_image(w, x, data_name)
        check('vsrn_img_%s feat' % tag, of, feat, 2e-6)
        check('vsrn_img_%s gcn' % tag, og, gcn, 2e-6)
        out.update({'%s_images' % tag: x, '%s_feat' % tag: feat, '%s_gcn' % tag: gcn})
        for k, v in w.items():
            out['%s_w_%s' % (tag, k)] = v
    # the model wrapper: forward_emb of `get_model` VSRN (Models.py:229-324) on a ragged batch, eval mode
    rng = np.random.RandomState(16)
    cfg = dict(name='VSRN', data_name='coco_precomp', img_dim=48, embed_size=64, use_abs=False, no_imgnorm=False, vocab_size=70,
               word_dim=24, num_layers=1, no_txtnorm=False, dim_vid=64, dim_hidden=32, bidirectional=False, input_dropout_p=0.2,
               rnn_type='gru', rnn_dropout_p=0.5, max_len=20, dim_word=16, margin=0.2, measure='cosine', max_violation=True,
               finetune=False, learning_rate=2e-4, grad_clip=2.0)
    torch.manual_seed(162)
    model = Models.VSRN(cfg)
    _randomise_bn(model.img_enc, 163)
    model.val_start()
    B = 7
    lengths = ragged_lengths(rng, B, 2, 9)
    ids = torch.zeros(B, max(lengths), dtype=torch.long)
    for b, l in enumerate(lengths):
        ids[b, :l] = torch.from_numpy(rng.randint(4, 70, size=l))
    x = mutils.l2norm(torch.randn(B, 36, 48), dim=-1)
    img_emb, cap_emb, gcn = model.forward_emb(x, ids, lengths)
    loss = model.criterion(img_emb, cap_emb)
    wi, wt = sd(model.img_enc), sd(model.txt_enc)
    of, og = O.vsrn_image(wi, x, 'coco_precomp')
    oc, _ = O.encoder_text(ids, lengths, wt, False, False, False, 'VSRN')
    check('vsrn model img', of, img_emb, 2e-6)
    check('vsrn model cap', oc, cap_emb, 2e-6)
    check('vsrn model loss', O.hinge_loss(O.cosine_sim(of, oc), 0.2, True), loss, 1e-5)
    out.update(m_images=x, m_ids=ids, m_lengths=np.array(lengths), m_img_emb=img_emb, m_cap_emb=cap_emb, m_gcn=gcn, m_loss=float(loss))
    for k, v in wi.items():
        out['m_img_' + k] = v
    for k, v in wt.items():
        out['m_txt_' + k] = v
    save('g16_vsrn', **out)


# ------------------------------------------------------------------ G17 measure='order': order_sim, SAEM pdist, hinge on top
def g17():
    torch.manual_seed(17)
    im = mutils.l2norm(torch.randn(9, 40), dim=-1).abs()          # order embeddings are used with use_abs
    s = mutils.l2norm(torch.randn(13, 40), dim=-1).abs()
    S = Objectives.order_sim(im, s)
    P = Objectives.pdist(im, s)
    check('order_sim', O.order_sim(im, s), S, 1e-6)
    check('pdist', O.pdist(im, s), P, 1e-6)
    out = dict(im=im, s=s, order=S, pdist=P)
    # ContrastiveLoss(measure='order') value and gradients on a square batch (VSE++ dispatch)
    with torch.enable_grad():
        a = im[:8].clone().requires_grad_(True)
        b = s[:8].clone().requires_grad_(True)
        for mv in (False, True):
            crit = Objectives.ContrastiveLoss(config={'name': 'VSE++'}, margin=0.2, measure='order', max_violation=mv)
            a.grad = b.grad = None
            loss = crit(a, b)
            loss.backward()
            tag = 'maxviol' if mv else 'sum'
            out.update({'loss_' + tag: float(loss), 'd_im_' + tag: a.grad.clone(), 'd_s_' + tag: b.grad.clone()})
            check('order loss ' + tag, O.hinge_loss(O.order_sim(a.detach(), b.detach()), 0.2, mv), float(loss), 1e-5)
    crit_saem = Objectives.ContrastiveLoss(config={'name': 'SAEM'}, margin=0.2, measure='order', max_violation=True)
    out['saem_order_loss'] = float(crit_saem(im[:8], s[:8]))
    save('g17_order', **out)


# ------------------------------------------------------------------ G18 SAEM.train_emb (a14), dropout probabilities 0
def g18():
    """The reference's own SAEM.train_emb (Models.py:444-464) run twice on CPU with every dropout probability set to 0 (the
    reference draws its masks from torch's generator; the build uses a counter-based hash, so only p = 0 is comparable
    value by value): losses, gradients of the trainable parameters and all parameters after each Adam step."""
    import json
    import tempfile
    from itr.modalmodule import bert as rbert
    rng = np.random.RandomState(18)
    out = {}
    tmp = tempfile.mkdtemp()
    cfg_d = dict(vocab_size=100, hidden_size=32, num_hidden_layers=2, num_attention_heads=2, intermediate_size=64,
                 max_position_embeddings=40, type_vocab_size=2, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    trans_d = dict(vocab_size=10, hidden_size=32, num_hidden_layers=1, num_attention_heads=2, intermediate_size=64,
                   max_position_embeddings=40, type_vocab_size=2, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    torch.manual_seed(180)
    bm = rbert.BertModel(rbert.BertConfig.from_dict(cfg_d))
    for name, p_ in bm.named_parameters():
        p_.data.normal_(0, 0.05)
        if name.endswith('gamma'):
            p_.data.uniform_(0.8, 1.2)
    json.dump(cfg_d, open(os.path.join(tmp, 'bert_config.json'), 'w'))
    json.dump(trans_d, open(os.path.join(tmp, 'trans_cfg.json'), 'w'))
    torch.save(bm.state_dict(), os.path.join(tmp, 'pytorch_model.bin'))
    out.update(bert_cfg=json.dumps(cfg_d), trans_cfg=json.dumps(trans_d))
    for k, v in sd(bm).items():
        out['wbert_' + k] = v
    B, L, F_ = 6, 12, 40
    for stru in ('cnn', 'trans', 'pooling'):
        cfg = dict(name='SAEM', bert_config_file=os.path.join(tmp, 'bert_config.json'), init_checkpoint=os.path.join(tmp, 'pytorch_model.bin'),
                   trans_cfg=os.path.join(tmp, 'trans_cfg.json'), txt_stru=stru, final_dims=32, img_dim=F_, bi_gru=False, embed_size=32,
                   num_layers=1, margin=0.2, measure='cosine', max_violation=True, learning_rate=1e-3, grad_clip=2.0)
        with torch.enable_grad():
            torch.manual_seed(181)
            model = Models.SAEM(cfg)
            for name, p_ in list(model.txt_enc.named_parameters()) + list(model.img_enc.named_parameters()):
                if not name.startswith('bert.'):
                    p_.data.normal_(0, 0.08)
                    if name.endswith('gamma'):
                        p_.data.uniform_(0.8, 1.2)
            model.train_start()
            model.logger = evaluation.LogCollector()
            for k, v in sd(model.img_enc).items():
                out['%s_w0_img_%s' % (stru, k)] = v
            for k, v in sd(model.txt_enc).items():
                if not k.startswith('bert.'):
                    out['%s_w0_txt_%s' % (stru, k)] = v
            for step in (1, 2):
                lens = sorted([int(x) for x in rng.randint(3, L + 1, size=B)], reverse=True)
                ids = torch.from_numpy(rng.randint(1, 100, size=(B, L)))
                mask = torch.zeros(B, L, dtype=torch.long)
                for b, l in enumerate(lens):
                    mask[b, :l] = 1
                    ids[b, l:] = 0
                types = torch.zeros(B, L, dtype=torch.long)
                feats = mutils.l2norm(torch.randn(B, 36, F_), dim=-1)
                model.train_emb((feats, None, None, ids, lens, list(range(B)), mask, types), epoch=step - 1)
                out.update({'%s_s%d_feats' % (stru, step): feats, '%s_s%d_ids' % (stru, step): ids, '%s_s%d_mask' % (stru, step): mask,
                            '%s_s%d_types' % (stru, step): types, '%s_s%d_lens' % (stru, step): np.array(lens),
                            '%s_s%d_loss1' % (stru, step): float(model.logger.meters['Loss1'].val),
                            '%s_s%d_loss2' % (stru, step): float(model.logger.meters['Loss2'].val)})
                for n, p_ in list(('txt.' + n, p_) for n, p_ in model.txt_enc.named_parameters()) + \
                        list(('img.' + n, p_) for n, p_ in model.img_enc.named_parameters()):
                    if p_.grad is not None:
                        out['%s_s%d_grad_%s' % (stru, step, n)] = p_.grad.detach().clone()
                if step == 2:          # parameters after both Adam steps (the step-1 state is implied by the step-1 gradients)
                    for k, v in sd(model.img_enc).items():
                        out['%s_s%d_img_%s' % (stru, step, k)] = v
                    for k, v in sd(model.txt_enc).items():
                        if not k.startswith('bert.'):
                            out['%s_s%d_txt_%s' % (stru, step, k)] = v
            print("   %s: loss1 %.5f / %.5f  loss2 %.5f / %.5f" % (stru, out[stru + '_s1_loss1'], out[stru + '_s2_loss1'],
                                                                   out[stru + '_s1_loss2'], out[stru + '_s2_loss2']))
    save('g18_saem_train', **out)


# ------------------------------------------------------------------ G19 CAMERA.train_emb (a14), drop = 0, BERT dropout 0
def _g19_try(model_seed):
    """The reference's own CAMERA.train_emb (Models.py:613-645) run twice on CPU (drop = 0 and BERT dropout probabilities 0, see
    g18): Rank_Loss / Div_loss, the gradients of the trainable parameters after step 1 and all parameters / BatchNorm running
    statistics after step 2."""
    import json
    import tempfile
    from itr.modalmodule import bert as rbert
    rng = np.random.RandomState(19)
    out = {}
    tmp = tempfile.mkdtemp()
    cfg_d = dict(vocab_size=100, hidden_size=32, num_hidden_layers=2, num_attention_heads=2, intermediate_size=64,
                 max_position_embeddings=40, type_vocab_size=2, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    torch.manual_seed(190)
    bm = rbert.BertModel(rbert.BertConfig.from_dict(cfg_d))
    for name, p_ in bm.named_parameters():
        p_.data.normal_(0, 0.05)
        if name.endswith('gamma'):
            p_.data.uniform_(0.8, 1.2)
    json.dump(cfg_d, open(os.path.join(tmp, 'bert_config.json'), 'w'))
    torch.save(bm.state_dict(), os.path.join(tmp, 'pytorch_model.bin'))
    out['bert_cfg'] = json.dumps(cfg_d)
    for k, v in sd(bm).items():
        out['wbert_' + k] = v
    B, L, F_, E = 3, 10, 24, 32       # 3 images: half as many relu inputs as 6 -- see g19 on why that matters
    cfg = dict(name='CAMERA', bert_config_file=os.path.join(tmp, 'bert_config.json'), init_checkpoint=os.path.join(tmp, 'pytorch_model.bin'),
               img_dim=F_, embed_size=E, head=2, smry_k=12, drop=0.0, margin=0.2, max_violation=True, batch_size=B, learning_rate=1e-3,
               grad_clip=2.0, smry_lamda=0.01)
    with torch.enable_grad():
        torch.manual_seed(model_seed)
        model = Models.CAMERA(cfg)
        for name, p_ in list(model.txt_enc.named_parameters()) + list(model.img_enc.named_parameters()):
            if 'bert.' not in name:
                p_.data.normal_(0, 0.1)
                if '.bns.' in name and name.endswith('weight') or name.endswith('bn.weight'):
                    p_.data.uniform_(0.6, 1.4)
        model.train_start()
        model.logger = evaluation.LogCollector()
        for k, v in sd(model.img_enc).items():
            out['w0_img_' + k] = v
        for k, v in sd(model.txt_enc).items():
            if 'bert.' not in k:
                out['w0_txt_' + k] = v
        for step in (1, 2):
            lens = sorted([int(x) for x in rng.randint(3, L + 1, size=B)], reverse=True)
            ids = torch.from_numpy(rng.randint(1, 100, size=(B, L)))
            mask = torch.zeros(B, L, dtype=torch.long)
            for b, l in enumerate(lens):
                mask[b, :l] = 1
                ids[b, l:] = 0
            types = torch.zeros(B, L, dtype=torch.long)
            feats = mutils.l2norm(torch.randn(B, 36, F_), dim=-1)
            x1y1 = torch.rand(B, 36, 2) * 300
            boxes = torch.cat([x1y1, x1y1 + 20 + torch.rand(B, 36, 2) * 150], 2)
            wh = torch.tensor([[640., 480.]]).repeat(B, 1)
            model.train_emb((feats, boxes, wh, ids, lens, list(range(B)), mask, types))
            pre = 's%d_' % step
            out.update({pre + 'feats': feats, pre + 'boxes': boxes, pre + 'wh': wh, pre + 'ids': ids, pre + 'mask': mask, pre + 'types': types,
                        pre + 'lens': np.array(lens), pre + 'rank_loss': float(model.logger.meters['Rank_Loss'].val),
                        pre + 'div_loss': float(model.logger.meters['Div_loss'].val)})
            if step == 1:
                for n, p_ in list(('txt.' + n, p_) for n, p_ in model.txt_enc.named_parameters()) + \
                        list(('img.' + n, p_) for n, p_ in model.img_enc.named_parameters()):
                    if p_.grad is not None:
                        out[pre + 'grad_' + n] = p_.grad.detach().clone()
            else:
                for k, v in sd(model.img_enc).items():
                    out[pre + 'img_' + k] = v
                for k, v in sd(model.txt_enc).items():
                    if 'bert.' not in k:
                        out[pre + 'txt_' + k] = v
        print("   rank loss %.5f / %.5f   div %.5f / %.5f" % (out['s1_rank_loss'], out['s2_rank_loss'], out['s1_div_loss'], out['s2_div_loss']))
    return out


def g19():
    """G19 with a fixture that is NOT ill-conditioned: a pre-activation within fp32 rounding of a relu kink flips its sign
    between the CPU run and any other fp32 evaluation order, and the gradients upstream then differ by percents (round 1's
    fixture had two such values within 5e-7).  The model seed is searched until every relu input of both steps (the dilated
    summarisation convolutions camera_.py:110 with their 1 024 channels x 36 regions per image, the text head
    TextEncoder.py:187 -- 223 000 pre-activations at batch 3) is at least 1e-6 away from zero, several times the absolute
    difference two fp32 evaluation orders produce on values of this size.  (A 1e-4 margin cannot exist: with that many values
    of scale ~0.1 about a hundred fall inside it for every seed; even 1e-6 takes tens of seeds to find.)"""
    import torch.nn.functional as F
    real_relu = F.relu
    for model_seed in range(191, 791):
        seen = []

        def spy(x, *a, **k):
            seen.append(float(x.detach().abs().min()))
            return real_relu(x, *a, **k)
        F.relu = spy
        try:
            out = _g19_try(model_seed)
        finally:
            F.relu = real_relu
        margin = min(seen)
        print("   model seed %d: %d relu calls, min |pre-activation| = %.2e" % (model_seed, len(seen), margin))
        if margin >= 1e-6:
            out['relu_margin'] = np.array(margin)
            out['model_seed'] = np.array(model_seed)
            save('g19_camera_train', **out)
            return
    raise RuntimeError("no well-conditioned seed found")


# ------------------------------------------------------------------ G20 SGRAF.train_emb (a14), dropout modules set to p = 0
def g20():
    """The reference's own SGRAF.train_emb (Models.py:524-546) run twice on CPU for SAF and SGR.  Its dropout probabilities are
    hard-coded (0.4 in VisualSA / TextSA / EncoderText) and drawn from torch's generator, so every nn.Dropout of the reference
    model is switched to p = 0 here (see g18); BatchNorm runs in training mode: batch statistics, running statistics updated
    (AttentionFiltration's BatchNorm1d(1) once per caption)."""
    rng = np.random.RandomState(20)
    out = {}
    V, F_, D, E, S, B = 60, 24, 32, 16, 16, 6
    for mod in ('SAF', 'SGR'):
        cfg = dict(name='SGRAF', img_dim=F_, embed_size=D, no_imgnorm=False, vocab_size=V, word_dim=E, num_layers=1, bi_gru=True,
                   no_txtnorm=False, sim_dim=S, module_name=mod, sgr_step=3, margin=0.2, measure='cosine', max_violation=True,
                   learning_rate=2e-3, grad_clip=2.0)
        with torch.enable_grad():
            torch.manual_seed(200)
            model = Models.SGRAF(cfg)
            for m in list(model.txt_enc.modules()) + list(model.sim_enc.modules()):
                if isinstance(m, torch.nn.Dropout):
                    m.p = 0.0
            if hasattr(model.txt_enc, 'dropout') and not isinstance(model.txt_enc.dropout, torch.nn.Module):
                model.txt_enc.dropout = 0.0
            model.train_start()
            model.logger = evaluation.LogCollector()
            for which, m in (('img', model.img_enc), ('txt', model.txt_enc), ('sim', model.sim_enc)):
                for k, v in sd(m).items():
                    out['%s_w0_%s_%s' % (mod, which, k)] = v
            for step in (1, 2):
                feats, ids, lens = _train_batch(rng, B, V, F_)
                model.train_emb((feats, None, None, ids, lens, list(range(B)), None, None))
                pre = '%s_s%d_' % (mod, step)
                out.update({pre + 'feats': feats, pre + 'ids': ids, pre + 'lens': np.array(lens), pre + 'loss': float(model.logger.meters['Loss'].val)})
                if step == 1:
                    for which, m in (('img', model.img_enc), ('txt', model.txt_enc), ('sim', model.sim_enc)):
                        for n, p_ in m.named_parameters():
                            if p_.grad is not None:
                                out[pre + 'grad_%s.%s' % (which, n)] = p_.grad.detach().clone()
                else:
                    for which, m in (('img', model.img_enc), ('txt', model.txt_enc), ('sim', model.sim_enc)):
                        for k, v in sd(m).items():
                            out[pre + '%s_%s' % (which, k)] = v
            print("   %s: loss %.5f / %.5f" % (mod, out[mod + '_s1_loss'], out[mod + '_s2_loss']))
    save('g20_sgraf_train', **out)


# ------------------------------------------------------------------ G21 VSRN.train_emb (f4), dropout modules set to p = 0
def g21():
    """The reference's own VSRN.train_emb (Models.py:343-365) run twice on CPU: retrieval + captioning losses, gradients of all
    parameters (towers and captioning model) after step 1, parameters after step 2.  nn.Dropout modules switched to p = 0 (see g18);
    captions in the loader's VSRN layout (max_len + 1 ids, data_loader.py:117-125)."""
    rng = np.random.RandomState(21)
    out = {}
    V, F_, D, E, B, max_len = 40, 20, 32, 12, 5, 8
    cfg = dict(name='VSRN', data_name='coco_precomp', img_dim=F_, embed_size=D, use_abs=False, no_imgnorm=False, vocab_size=V, word_dim=E,
               num_layers=1, no_txtnorm=False, dim_vid=D, dim_hidden=16, bidirectional=False, input_dropout_p=0.2, rnn_type='gru',
               rnn_dropout_p=0.0, max_len=max_len, dim_word=10, margin=0.2, measure='cosine', max_violation=True, finetune=False,
               learning_rate=2e-3, grad_clip=2.0)
    with torch.enable_grad():
        torch.manual_seed(210)
        model = Models.VSRN(cfg)
        _randomise_bn(model.img_enc, 211)
        for m in list(model.caption_model.modules()) + list(model.txt_enc.modules()):
            if isinstance(m, torch.nn.Dropout):
                m.p = 0.0
        model.train_start()
        model.caption_model.train()
        model.logger = evaluation.LogCollector()
        mods = (('img', model.img_enc), ('txt', model.txt_enc), ('cap', model.caption_model))
        for which, m in mods:
            for k, v in sd(m).items():
                out['w0_%s_%s' % (which, k)] = v
        for step in (1, 2):
            feats = mutils.l2norm(torch.randn(B, 36, F_), dim=-1)
            true_len = [int(x) for x in rng.randint(3, max_len + 3, size=B)]
            ids = torch.zeros(B, max_len + 1, dtype=torch.long)
            mask = torch.zeros(B, max_len + 1)
            for b, l in enumerate(true_len):
                toks = [1] + [int(x) for x in rng.randint(4, V, size=l - 2)] + [2]
                if len(toks) > max_len:
                    toks[max_len] = toks[-1]
                    toks = toks[:max_len]
                ids[b, :len(toks)] = torch.tensor(toks)
                mask[b, :max_len] = 1            # the reference's mask is computed after the padding (data_loader.py:123-124)
            lens = [max_len + 1] * B
            model.train_emb((feats, None, None, ids, lens, list(range(B)), mask, None))
            pre = 's%d_' % step
            out.update({pre + 'feats': feats, pre + 'ids': ids, pre + 'mask': mask, pre + 'lens': np.array(lens),
                        pre + 'loss_caption': float(model.logger.meters['Loss_caption'].val),
                        pre + 'loss_retrieval': float(model.logger.meters['Loss_retrieval'].val)})
            if step == 1:
                for which, m in mods:
                    for n, p_ in m.named_parameters():
                        if p_.grad is not None:
                            out[pre + 'grad_%s.%s' % (which, n)] = p_.grad.detach().clone()
            else:
                for which, m in mods:
                    for k, v in sd(m).items():
                        out[pre + '%s_%s' % (which, k)] = v
        print("   caption loss %.5f / %.5f   retrieval %.5f / %.5f" % (out['s1_loss_caption'], out['s2_loss_caption'], out['s1_loss_retrieval'],
                                                                       out['s2_loss_retrieval']))
    save('g21_vsrn_train', **out)


if __name__ == "__main__":
    os.makedirs(GOLD, exist_ok=True)
    which = sys.argv[1:] or ['g1', 'g2', 'g3', 'g4', 'g5', 'g6', 'g78', 'g9', 'g10', 'g11', 'g12', 'g13', 'g14', 'g15', 'g16', 'g17', 'g18', 'g19', 'g20', 'g21', 'g22']
    for name in which:
        print("== " + name)
        globals()[name]()
