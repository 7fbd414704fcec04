"""GPU: BERT / SAEM towers on the HIP path vs golden vectors captured from the reference (g10)."""
import json
import os

import numpy as np
import pytest
import torch

from itr_amd import ops
from itr_amd.modalmodule import bert, TextEncoder, ImgEncoder

pytestmark = pytest.mark.gpu


def T(a):
    return torch.from_numpy(np.asarray(a))


def wdict(g, pre):
    return {k[len(pre):]: T(g[k]) for k in g.files if k.startswith(pre)}


def maxdiff(a, b):
    return float((a.detach().cpu().double() - torch.as_tensor(np.asarray(b)).double()).abs().max())


def test_bert_model_golden(golden, dev):
    g = golden("g10_bert_saem")
    cfg = bert.BertConfig.from_dict(json.loads(str(g["bert_cfg"])))
    model = bert.BertModel(cfg)
    model.load_state_dict(wdict(g, "wbert_"))
    model.cuda().eval()
    layers, pooled = model(T(g["bert_ids"]).to(dev), T(g["bert_types"]).to(dev), T(g["bert_mask"]).to(dev))
    assert maxdiff(layers[0], g["bert_layer0"]) <= 2e-5
    assert maxdiff(layers[1], g["bert_layer1"]) <= 2e-5
    assert maxdiff(pooled, g["bert_pooled"]) <= 2e-5


def test_bert_wide_layer_golden(golden, dev):
    g = golden("g10_bert_saem")
    cfg = bert.BertConfig.from_dict(json.loads(str(g["wide_cfg"])))
    lay = bert.BERTLayer(cfg)
    lay.load_state_dict({k[len("layer."):]: v for k, v in wdict(g, "wwide_").items()})
    lay.cuda().eval()
    m = T(g["wide_mask"]).to(dev)
    y = lay(T(g["wide_x"]).to(dev), m)
    assert maxdiff(y, g["wide_y"]) <= 2e-5
    ext = ((1.0 - m) * -10000.0)[:, None, None, :]          # the reference's extended additive mask also works
    assert maxdiff(lay(T(g["wide_x"]).to(dev), ext), g["wide_y"]) <= 2e-5


@pytest.mark.parametrize("stru", ["cnn", "pooling", "trans"])
def test_saem_text_golden(golden, dev, tmp_path, stru):
    g = golden("g10_bert_saem")
    open(tmp_path / "bert_config.json", "w").write(str(g["bert_cfg"]))
    open(tmp_path / "trans_cfg.json", "w").write(str(g["trans_cfg"]))
    torch.save(wdict(g, "wbert_"), tmp_path / "pytorch_model.bin")
    cfg = dict(bert_config_file=str(tmp_path / "bert_config.json"), init_checkpoint=str(tmp_path / "pytorch_model.bin"),
               txt_stru=stru, final_dims=64, trans_cfg=str(tmp_path / "trans_cfg.json"))
    tm = TextEncoder.BertMapping(cfg)
    sd = tm.state_dict()
    sd.update(wdict(g, "wsaem_%s_" % stru))
    tm.load_state_dict(sd)
    tm.cuda().eval()
    code = tm(T(g["bert_ids"]).to(dev), T(g["bert_mask"]).to(dev), T(g["bert_types"]).to(dev), None)
    assert maxdiff(code, g["saem_text_" + stru]) <= 2e-5


def test_saem_image_golden(golden, dev, tmp_path):
    g = golden("g10_bert_saem")
    open(tmp_path / "trans_cfg.json", "w").write(str(g["trans_cfg"]))
    im = ImgEncoder.TransformerMapping(dict(trans_cfg=str(tmp_path / "trans_cfg.json"), img_dim=96, final_dims=64))
    im.load_state_dict(wdict(g, "wsaemimg_"))
    im.cuda().eval()
    assert maxdiff(im(T(g["saem_img_x"]).to(dev)), g["saem_img_y"]) <= 2e-5


@pytest.mark.parametrize("n_layers", [2, 12])
def test_bert_base_shape_vs_oracle(dev, n_layers):
    """BERT-base geometry (768 hidden, 12 heads of 64, 3072 intermediate), 32 tokens; 12 layers = the full stack of
    BASELINE config 4 (bert.py:305-358), EVERY layer's output against the oracle."""
    import itr_oracle as O
    torch.manual_seed(0)
    cfg = bert.BertConfig(vocab_size=500, hidden_size=768, num_hidden_layers=n_layers, num_attention_heads=12,
                          intermediate_size=3072, max_position_embeddings=64, type_vocab_size=2)
    model = bert.BertModel(cfg)
    for p in model.parameters():
        p.data.normal_(0, 0.03)
    w = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model.cuda().eval()
    rng = np.random.RandomState(0)
    ids = torch.from_numpy(rng.randint(1, 500, size=(6, 32)))
    mask = torch.ones(6, 32, dtype=torch.long)
    for b, l in enumerate([32, 20, 11, 7, 3, 1]):
        mask[b, l:] = 0
    layers, pooled = model(ids.to(dev), None, mask.to(dev))
    ol, op = O.bert_model(w, ids, None, mask, n_layers, 12)
    assert len(layers) == len(ol) == n_layers
    for a, b in zip(layers, ol):
        assert maxdiff(a, b) <= 5e-5
    assert maxdiff(pooled, op) <= 5e-5


def test_camera_towers_golden(golden, dev, tmp_path):
    g = golden("g13_camera")
    open(tmp_path / "bert_config.json", "w").write(str(g["bert_cfg"]))
    wt = wdict(g, "wtxt_")
    torch.save({k[len("bert."):]: v for k, v in wt.items() if k.startswith("bert.")}, tmp_path / "pytorch_model.bin")
    ie = ImgEncoder.EncoderImagePrecompSelfAttn(96, 64, 4, 12)
    sd = ie.state_dict()
    sd.update(wdict(g, "wimg_"))
    ie.load_state_dict(sd)
    ie.cuda().eval()
    emb, smry = ie(T(g["images"]).to(dev), T(g["boxes"]).to(dev), T(g["imgs_wh"]).to(dev))
    assert maxdiff(smry, g["smry_mat"]) <= 5e-5
    assert maxdiff(emb, g["img_emb"]) <= 2e-5
    te = TextEncoder.CAMERAEncoderText(str(tmp_path / "bert_config.json"), str(tmp_path / "pytorch_model.bin"), 64, 4)
    sd = te.state_dict()
    sd.update(wt)
    te.load_state_dict(sd)
    te.cuda().eval()
    cap = te(T(g["ids"]).to(dev), T(g["mask"]).to(dev), T(g["types"]).to(dev))
    assert maxdiff(cap, g["cap_emb"]) <= 2e-5
    from itr_amd.modalmodule import Fusionmodule, Objectives
    sim = Fusionmodule.MultiViewMatching()(emb, cap)
    assert maxdiff(sim, g["sim"]) <= 2e-5
    loss = Objectives.TripletLoss(0.2, True)(sim[:, :5].contiguous())
    import itr_oracle as O
    want = O.hinge_loss(T(g["sim"])[:, :5], 0.2, True)
    assert abs(float(loss) - float(want)) <= 1e-4


def test_aux_losses_golden(golden, dev):
    """a18: training-time auxiliaries (torch-composed on the GPU)."""
    from itr_amd.modalmodule import Objectives
    g = golden("g13_camera")
    div = Objectives.DiversityRegularization(12, 5)(T(g["smry_mat"]).to(dev))
    assert abs(float(div) - float(g["div_reg"])) <= 1e-3 * max(1.0, abs(float(g["div_reg"])))
    ang = Objectives.AngularLoss()(T(g["ang_im"]).to(dev), T(g["ang_s"]).to(dev))
    assert abs(float(ang) - float(g["ang_loss"])) <= 1e-3 * max(1.0, abs(float(g["ang_loss"])))


@pytest.mark.parametrize("dk", [16, 32, 64])
@pytest.mark.parametrize("L", [1, 7, 16, 17, 32, 36, 49, 64])
@pytest.mark.parametrize("lds", [False, True])
def test_attention_kernel_shapes(dev, L, dk, lds):
    """softmax(Q K^T / sqrt(dk) + (1 - mask) * -10000) V (bert.py:185-207) for every tile count / head width of the two
    matrix-core kernels (register-only: the default; LDS-staged: outputs without 16-byte rows -- here a column block of a wider
    buffer that starts 4 bytes into the row),
    with a ragged key mask, fused-QKV strides, and an odd number of (sequence, head) pairs."""
    torch.manual_seed(L * 100 + dk)
    B, heads = 3, 3
    H = heads * dk
    qkv = torch.randn(B * L, 3 * H)
    mask = torch.ones(B, L)
    for b in range(B):
        mask[b, max(1, L - 2 * b):] = 0
    scale = 1.0 / dk ** 0.5
    q, k, v = (qkv[:, i * H:(i + 1) * H].reshape(B, L, heads, dk).permute(0, 2, 1, 3) for i in range(3))
    sc = q @ k.transpose(-1, -2) * scale + ((1.0 - mask) * -10000.0)[:, None, None, :]
    want = (torch.softmax(sc, -1) @ v).permute(0, 2, 1, 3).reshape(B * L, H)
    d = qkv.to(dev)
    wide = torch.full((B * L, H + 3), float("nan"), device=dev)
    out_of = lambda: wide[:, 1:1 + H] if lds else None          # misaligned rows: the LDS-staged kernel (csrc/transformer.hip)
    got = ops.mha_small(d[:, :H], d[:, H:2 * H], d[:, 2 * H:], mask.to(dev), B, L, heads, dk, scale, out=out_of())
    assert float((got.cpu() - want).abs().max()) <= 2e-6
    if lds:
        assert bool(torch.isnan(wide[:, 0]).all()) and bool(torch.isnan(wide[:, 1 + H:]).all())      # nothing written outside the block
    got2 = ops.mha_small(d[:, :H], d[:, H:2 * H], d[:, 2 * H:], None, B, L, heads, dk, scale, out=out_of())
    want2 = (torch.softmax(q @ k.transpose(-1, -2) * scale, -1) @ v).permute(0, 2, 1, 3).reshape(B * L, H)
    assert float((got2.cpu() - want2).abs().max()) <= 2e-6


@pytest.mark.parametrize("dk,rows", [(32, 16 * 700 + 5), (16, 16 * 300 + 11), (32, 3), (16, 16)])
def test_agsa_gate(dev, dk, rows):
    """GatedQueryAttLayer's gate (camera_.py:36-44) as one kernel (csrc/agsa_gate.hip) against float64: M = sigmoid(fc_g(fc_q(q) *
    fc_k(k))), q * M[:, :dk], k * M[:, dk:]; a ragged last 16-row tile, fewer rows than one tile, exactly one tile."""
    torch.manual_seed(dk + rows)
    q, k = torch.randn(rows, dk), torch.randn(rows, dk)
    lin = lambda o, i: (torch.randn(o, i) / i ** 0.5, torch.randn(o) * 0.1)
    fq, fk, fg = lin(dk, dk), lin(dk, dk), lin(2 * dk, dk)
    G = (q.double() @ fq[0].double().t() + fq[1].double()) * (k.double() @ fk[0].double().t() + fk[1].double())
    M = torch.sigmoid(G @ fg[0].double().t() + fg[1].double())
    qo, ko = ops.agsa_gate(q.to(dev), k.to(dev), *[(w.to(dev), b.to(dev)) for w, b in (fq, fk, fg)])
    assert float((qo.cpu().double() - q.double() * M[:, :dk]).abs().max()) <= 2e-6
    assert float((ko.cpu().double() - k.double() * M[:, dk:]).abs().max()) <= 2e-6
    with pytest.raises(NotImplementedError):
        ops.agsa_gate(torch.zeros(4, 64, device=dev), torch.zeros(4, 64, device=dev), *[(torch.zeros(o, 64, device=dev), torch.zeros(o, device=dev)) for o in (64, 64, 128)])


@pytest.mark.gpu
def test_gelu_epilogue_sweep(dev):
    """The gelu of the GEMM epilogues (csrc/itr_common.h::gelu_erf: a branch-free erf, tools/fit_erf.py) against
    x/2 (1 + erf(x / sqrt 2)) in float64 (bert.py:29-34) on a dense sweep of [-9, 9] plus the special values: at least as
    close to the float64 value as torch's own fp32 gelu, the error bound of the fit, and the IEEE corner cases of the formula."""
    n = 128 * 2048
    x = torch.linspace(-9.0, 9.0, n * 64, device=dev).reshape(n, 64).contiguous()
    eye = torch.eye(64, device=dev)
    got = ops.linear(x, eye, None, act='gelu').double()         # x @ I = x exactly, then the epilogue's activation
    xd = x.double()
    want = 0.5 * xd * (1.0 + torch.erf(xd / np.sqrt(2.0)))
    err = float((got - want).abs().max())
    err_torch = float((torch.nn.functional.gelu(x).double() - want).abs().max())
    assert err <= 6e-7 and err <= err_torch, (err, err_torch)
    sp = torch.zeros(128, 64, device=dev)                       # one special value per row, in column 0 (the other columns see value * 0)
    sp[:8, 0] = torch.tensor([0.0, -0.0, 1e-30, -1e-30, 40.0, -40.0, float("inf"), float("nan")], device=dev)
    out = ops.linear(sp, eye, None, act='gelu')[:8, 0].cpu()
    assert out[0] == 0 and out[1] == 0 and abs(float(out[2])) <= 1e-30 and abs(float(out[3])) <= 1e-30
    assert float(out[4]) == 40.0 and float(out[5]) == 0.0 and out[6] == float("inf") and bool(torch.isnan(out[7]))
