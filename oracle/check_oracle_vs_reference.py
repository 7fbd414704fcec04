#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY -- live cross-check of the CPU restatement (oracle/itr_oracle.py) against the imported reference.

Runs ONLY in the build container (exits 0 with a note when /root/reference is absent: the reference never travels).  Unlike
oracle/make_goldens.py it writes nothing: it draws FRESH seeded inputs (a seed given on the command line, default 1234: not the
seeds of the committed fixtures) and compares, for the functions of SURVEY.md 8(a) the hot path is built from,

    reference (imported under oracle/ref_shim.py)   vs   oracle/itr_oracle.py

    python oracle/check_oracle_vs_reference.py [seed]

  l2norm / l1norm                      modalmodule/utils.py:4-15                  bit-exact
  EncoderImagePrecomp.forward          modalmodule/ImgEncoder.py:133-147          <= 1e-6
  EncoderText.forward (bi-GRU)         modalmodule/TextEncoder.py:38-70           <= 1e-6
  cosine_sim                           modalmodule/Objectives.py:18-21            bit-exact
  ContrastiveLoss (VSE++, max viol.)   modalmodule/Objectives.py:76-115           <= 1e-6
  xattn_score_t2i / _i2t (4 agg x 5 norm)   modalmodule/Objectives.py:329-417     <= 2e-5
  EncoderSimilarity.forward SAF / SGR  modalmodule/Fusionmodule.py:406-451        <= 2e-6
  i2t / t2i (return_ranks=True)        metricmodule/evaluation.py:156-222         identical rank vectors (tie-free matrix)
"""
import os
import sys
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shim  # noqa: E402

if not ref_shim.reference_available():
    print("check_oracle_vs_reference: %s is absent (GPU box?): nothing to check" % ref_shim.REFERENCE_ROOT)
    sys.exit(0)
Objectives, ImgEncoder, TextEncoder, Fusionmodule, Models, evaluation, mutils = ref_shim.import_reference()
import itr_oracle as O  # noqa: E402

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1234
rng = np.random.RandomState(seed)
torch.manual_seed(seed)
worst = {}


def check(name, got, want, tol):
    d = float((torch.as_tensor(got).double() - torch.as_tensor(want).double()).abs().max())
    worst[name] = d
    status = "ok" if d <= tol else "FAIL"
    print("  %-44s max|d| = %.3e   (tol %.0e)  %s" % (name, d, tol, status))
    if d > tol:
        raise SystemExit("oracle differs from the reference: %s" % name)


def sd(module):
    return {k: v.detach().clone() for k, v in module.state_dict().items() if "num_batches_tracked" not in k}


with torch.no_grad():
    # ---- norms
    x = torch.randn(7, 36, 48)
    x[2, 5] = 0
    check("l2norm", O.l2norm(x, -1), mutils.l2norm(x, dim=-1), 0.0)
    check("l1norm", O.l1norm(x, -1), mutils.l1norm(x, dim=-1), 0.0)

    # ---- image tower
    enc = ImgEncoder.EncoderImagePrecomp(96, 64)
    feats = mutils.l2norm(torch.randn(6, 36, 96), dim=-1)
    w = sd(enc)
    check("EncoderImagePrecomp", O.encoder_image_precomp(feats, w["fc.weight"], w["fc.bias"]), enc(feats), 1e-6)

    # ---- text tower (bi-GRU, ragged, sorted descending like collate_fn)
    lens = sorted([int(v) for v in rng.randint(2, 12, size=9)], reverse=True)
    ids = torch.from_numpy(rng.randint(4, 50, size=(9, max(lens))))
    txt = TextEncoder.EncoderText(50, 20, 64, 1, use_bi_gru=True, no_txtnorm=True)
    txt.eval()
    ref_cap, ref_len = txt(ids, lens)
    got_cap, _ = O.encoder_text(ids, lens, sd(txt), True, True, False, None)
    check("EncoderText bi-GRU", got_cap, ref_cap, 1e-6)

    # ---- cosine + hinge
    im, s = torch.randn(16, 64), torch.randn(16, 64)
    check("cosine_sim", O.cosine_sim(im, s), Objectives.cosine_sim(im, s), 0.0)

# the reference's criterion keeps autograd on; value only
crit = Objectives.ContrastiveLoss({"name": "VSE++"}, margin=0.2, measure="cosine", max_violation=True)
im = torch.nn.functional.normalize(torch.randn(24, 32), dim=-1)
s = torch.nn.functional.normalize(torch.randn(24, 32), dim=-1)
check("ContrastiveLoss max_violation", O.hinge_loss(O.cosine_sim(im, s), 0.2, True), crit(im, s).detach(), 1e-6)

with torch.no_grad():
    # ---- SCAN cross attention
    Ni, Nc, R, D = 6, 10, 36, 48
    lens = [int(v) for v in rng.randint(2, 11, size=Nc)]
    L = max(lens)
    img = mutils.l2norm(torch.randn(Ni, R, D), dim=-1)
    cap = torch.randn(Nc, L, D) * 0.7
    for c, l in enumerate(lens):
        cap[c, l:] = 0
    for xa, fn in (("t2i", Objectives.xattn_score_t2i), ("i2t", Objectives.xattn_score_i2t)):
        for agg in ("LogSumExp", "Mean", "Max", "Sum"):
            for norm in ("clipped_l2norm", "l2norm", "softmax", "no_norm", "clipped"):
                cfg = dict(name="SCAN", lambda_lse=6.0, lambda_softmax=9.0, cross_attn=xa, agg_func=agg, raw_feature_norm=norm)
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    ref = fn(img, cap, lens, cfg)
                check("xattn_score_%s %s %s" % (xa, agg, norm), O.xattn_score(img, cap, lens, xa, norm, agg, 6.0, 9.0), ref, 2e-5)

    # ---- SGRAF similarity
    capn = mutils.l2norm(cap, dim=-1)
    for c, l in enumerate(lens):
        capn[c, l:] = 0
    for mod in ("SAF", "SGR"):
        torch.manual_seed(seed + 1)
        enc = Fusionmodule.EncoderSimilarity(D, 32, mod, 3)
        for m in enc.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.running_mean.normal_(0, 0.1)
                m.running_var.uniform_(0.5, 1.5)
        enc.eval()
        check("EncoderSimilarity " + mod, O.sgraf_similarity(sd(enc), img, capn, lens, mod, 3), enc(img, capn, lens), 2e-6)

# ---- ranker (tie-free random float64 matrix)
sims = rng.randn(40, 200)
(ri, (ranks_i, top_i)) = evaluation.i2t(sims, True)
(rt, (ranks_t, top_t)) = evaluation.t2i(sims, True)
c = O.rank_counts(sims)
assert (c[0] == ranks_i).all() and (c[1] == top_i).all(), "i2t rank vectors differ"
assert (c[2] == ranks_t).all() and (c[3] == top_t).all(), "t2i rank vectors differ"
print("  %-44s identical (i2t, t2i: ranks and top-1 of a 40 x 200 matrix)" % "i2t / t2i(return_ranks=True)")
print("check_oracle_vs_reference: ok (seed %d; worst %s = %.3e)" % (seed, max(worst, key=worst.get), max(worst.values())))
