"""GPU: VSRN retrieval side -- the region-relationship GCN + region GRU image tower, the last-state
GRU text tower, cosine scoring and ranking -- against G16 (captured from the reference's EncoderImagePrecompAttn and
Models.VSRN) and against the oracle at larger random shapes."""
import numpy as np
import pytest
import torch

import itr_oracle as O
from itr_amd import config as C, ops
from itr_amd.modalmodule import get_model, ImgEncoder
from itr_amd.metricmodule import evaluation

pytestmark = pytest.mark.gpu


def T(a):
    return torch.from_numpy(np.asarray(a))


def _randomise_bn(module, seed):
    g = torch.Generator().manual_seed(seed)
    for m in module.modules():
        if isinstance(m, torch.nn.BatchNorm1d):
            m.weight.data = (torch.rand(m.weight.shape, generator=g) * 0.6 + 0.2).to(m.weight.device)
            m.bias.data = (torch.randn(m.bias.shape, generator=g) * 0.05).to(m.weight.device)
            m.running_mean.data = (torch.randn(m.running_mean.shape, generator=g) * 0.1).to(m.weight.device)
            m.running_var.data = (torch.rand(m.running_var.shape, generator=g) + 0.5).to(m.weight.device)


def test_gcn_relation_kernel(dev):
    """(theta phi^T / N) g per image, ragged N and a D that is not a multiple of the 64-column LDS chunk."""
    for n_img, N, Cc in ((3, 36, 64), (2, 36, 200), (5, 7, 36), (1, 1, 4)):
        torch.manual_seed(N)
        tpg = torch.randn(n_img * N, 3 * Cc)
        y = ops.gcn_relation(tpg.to(dev), n_img, N, Cc).cpu().view(n_img, N, Cc)
        th, ph, g = (tpg.view(n_img, N, 3, Cc)[:, :, i].double() for i in range(3))
        want = (th @ ph.transpose(1, 2) / N) @ g
        assert (y.double() - want).abs().max().item() <= 2e-5 * max(1.0, want.abs().max().item())
    with pytest.raises(ValueError):
        ops.gcn_relation(torch.randn(10, 12).to(dev), 2, 5, 5)
    with pytest.raises(Exception):
        ops.gcn_relation(torch.randn(2 * 40, 12).to(dev), 2, 40, 4)       # more than 36 regions


@pytest.mark.parametrize("tag,data_name", [("coco", "coco_precomp"), ("f30k", "f30k_precomp")])
def test_vsrn_image_tower_golden(golden, dev, tag, data_name):
    g = golden("g16_vsrn")
    enc = ImgEncoder.EncoderImagePrecompAttn(48, 64, data_name).eval()
    enc.load_state_dict({k[len(tag) + 3:]: T(g[k]) for k in g.files if k.startswith(tag + "_w_")})
    enc.cuda()
    feat, gcn = enc(T(g[tag + "_images"]).to(dev))
    assert np.abs(feat.cpu().numpy() - g[tag + "_feat"]).max() <= 2e-6
    assert np.abs(gcn.cpu().numpy() - g[tag + "_gcn"]).max() <= 2e-6
    # one module, two modes: in training mode the same call runs on the autograd tape (GCN BatchNorms on batch statistics,
    # tests/test_vsrn_train_gpu.py pins its numbers against the reference's train_emb)
    enc.train()
    feat_t, gcn_t = enc(T(g[tag + "_images"]).to(dev))
    assert feat_t.requires_grad and gcn_t.requires_grad and feat_t.shape == feat.shape and not feat.requires_grad
    feat_t.square().sum().backward()
    assert enc.fc.weight.grad is not None and bool(torch.isfinite(enc.fc.weight.grad).all())


def test_vsrn_model_golden(golden, dev):
    """get_model(VSRN).forward_emb / forward_loss == the reference's Models.VSRN in eval mode; checkpoint round trip."""
    g = golden("g16_vsrn")
    cfg = C.build_config(['with', 'VSRN', 'data_name=coco_precomp', 'max_violation=True'])
    cfg.update(img_dim=48, embed_size=64, word_dim=24, vocab_size=70)
    model = get_model(cfg)
    model.load_state_dict([{k[6:]: T(g[k]) for k in g.files if k.startswith("m_img_") and k != "m_img_emb"},
                           {k[6:]: T(g[k]) for k in g.files if k.startswith("m_txt_")}])
    model.val_start()
    lengths = [int(x) for x in g["m_lengths"]]
    img_emb, cap_emb, gcn = model.forward_emb(T(g["m_images"]), T(g["m_ids"]), lengths)
    assert np.abs(img_emb.cpu().numpy() - g["m_img_emb"]).max() <= 2e-6
    assert np.abs(cap_emb.cpu().numpy() - g["m_cap_emb"]).max() <= 2e-6
    assert np.abs(gcn.cpu().numpy() - g["m_gcn"]).max() <= 2e-6
    assert float(model.forward_loss(img_emb, cap_emb)) == pytest.approx(float(g["m_loss"]), abs=1e-5)
    with pytest.raises(NotImplementedError):
        model.forward_loss(img_emb, cap_emb, gcn, T(g["m_ids"]), None)      # the captioning term lives in train_emb (tests/test_vsrn_train_gpu.py)
    sd = model.state_dict()
    assert len(sd) == 2 and 'Rs_GCN_1.W.1.running_mean' in sd[0] and 'img_rnn.weight_ih_l0' in sd[0] and 'rnn.weight_hh_l0' in sd[1]
    m2 = get_model(cfg)
    m2.load_state_dict(sd)
    m2.val_start()
    assert torch.equal(m2.forward_emb(T(g["m_images"]), T(g["m_ids"]), lengths)[0], img_emb)


def test_vsrn_tower_vs_oracle_wide(dev):
    """D = 256 (several LDS chunks, MFMA GEMM fast path), 40 images."""
    torch.manual_seed(9)
    for data_name in ("coco_precomp", "f30k_precomp"):
        enc = ImgEncoder.EncoderImagePrecompAttn(128, 256, data_name).eval()
        _randomise_bn(enc, 10)
        x = torch.randn(40, 36, 128)
        x = x / x.norm(dim=-1, keepdim=True)
        w = {k: v.clone() for k, v in enc.state_dict().items()}
        want_f, want_g = O.vsrn_image(w, x, data_name)
        enc.cuda()
        feat, gcn = enc(x.to(dev))
        assert (feat.cpu() - want_f).abs().max().item() <= 3e-6
        assert (gcn.cpu() - want_g).abs().max().item() <= 3e-6


def test_vsrn_eval_harness(dev):
    """encode_data -> cal_sims -> i2t / t2i through the reference-shaped harness: rank vectors == oracle's on the same model."""
    from test_models_gpu import FakeLoader
    rng = np.random.RandomState(3)
    cfg = C.build_config(['with', 'VSRN', 'data_name=coco_precomp'])
    cfg.update(img_dim=32, embed_size=64, word_dim=16, vocab_size=60)
    torch.manual_seed(4)
    model = get_model(cfg)
    _randomise_bn(model.img_enc, 5)
    model.val_start()
    n_img = 12
    feats = torch.randn(n_img, 36, 32)
    feats = feats / feats.norm(dim=-1, keepdim=True)
    lengths = [int(v) for v in rng.randint(2, 10, size=5 * n_img)]
    ids = torch.zeros(5 * n_img, 10, dtype=torch.long)
    for j, l in enumerate(lengths):
        ids[j, :l] = T(rng.randint(4, 60, size=l))
    loader = FakeLoader(feats.repeat_interleave(5, 0), ids, lengths, 16)
    img_embs, cap_embs, _ = evaluation.encode_data(model, loader, islength=False)
    sims = evaluation.cal_sims(model, img_embs[::5], cap_embs, None, shard_size=7)
    wi = {k: v.cpu() for k, v in model.img_enc.state_dict().items()}
    wt = {k: v.cpu() for k, v in model.txt_enc.state_dict().items()}
    want_img, _ = O.vsrn_image(wi, feats, 'coco_precomp')
    order = sorted(range(len(lengths)), key=lambda i: -lengths[i])
    want_cap = torch.zeros(len(lengths), 64)
    oc, _ = O.encoder_text(ids[order], [lengths[i] for i in order], wt, False, False, False, 'VSRN')
    want_cap[order] = oc
    want = O.cosine_sim(want_img, want_cap).numpy()
    assert np.abs(sims - want).max() <= 3e-6
    r, (ranks, top1) = evaluation.i2t(sims, return_ranks=True)
    wr, (wranks, wtop1) = O.i2t_argsort(want, return_ranks=True)
    assert (ranks == wranks).all() and (top1 == wtop1).all()
    rt, (ranks_t, top1_t) = evaluation.t2i(sims, return_ranks=True)
    wt_, (wranks_t, wtop1_t) = O.t2i_argsort(want, return_ranks=True)
    assert (ranks_t == wranks_t).all() and (top1_t == wtop1_t).all()
