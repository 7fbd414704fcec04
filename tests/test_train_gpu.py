"""GPU: the training step (a14 `train_emb`): every differentiable HIP op against torch autograd run on the CPU
oracle, the optimizer kernels, and model.train_emb against the REFERENCE's own SCAN.train_emb / VSE++ components
(tests/golden/g15_train_step.npz: loss, clipped gradients and parameters after each Adam step)."""
import json

import numpy as np
import pytest
import torch

import itr_oracle as O
from itr_amd import autograd as ag, config as C, ops
from itr_amd.modalmodule import get_model

pytestmark = pytest.mark.gpu


def T(a):
    return torch.from_numpy(np.asarray(a))


def md(a, b):
    a = a.detach().cpu().double() if torch.is_tensor(a) else torch.as_tensor(np.asarray(a)).double()
    b = b.detach().cpu().double() if torch.is_tensor(b) else torch.as_tensor(np.asarray(b)).double()
    assert a.shape == b.shape, (a.shape, b.shape)
    return float((a - b).abs().max()) if a.numel() else 0.0


# ------------------------------------------------------------------------------------------ primitives
@pytest.mark.parametrize("rows,cols", [(1, 1), (37, 130), (300, 64), (4608, 1024)])
def test_transpose_and_colsum(dev, rows, cols):
    torch.manual_seed(rows)
    x = torch.randn(rows, cols)
    assert torch.equal(ag.transpose2d(x.to(dev)).cpu(), x.t().contiguous())
    assert md(ag.colsum(x.to(dev)), x.double().sum(0)) <= 1e-5 * max(1.0, rows ** 0.5)


def test_l2norm_fwd_bwd(dev):
    torch.manual_seed(0)
    x = torch.randn(50, 96)
    x[7] = 0
    w = torch.randn(50, 96)
    xr = x.clone().requires_grad_(True)
    with torch.enable_grad():
        z = O.l2norm(xr, -1)
        (z * w).sum().backward()
    xg = x.to(dev).requires_grad_(True)
    with torch.enable_grad():
        zg = ag.l2norm_rows(xg)
        (zg * w.to(dev)).sum().backward()
    assert md(zg, z) <= 1e-6
    keep = [i for i in range(50) if i != 7]       # the zero row: slope 1 / eps either way, value differs by rounding only
    assert md(xg.grad[keep], xr.grad[keep]) <= 2e-6
    assert torch.isfinite(xg.grad).all()


@pytest.mark.parametrize("shape", [((33, 40), 17), ((5, 36, 64), 32)])
def test_linear_backward(dev, shape):
    torch.manual_seed(1)
    (xs, N) = shape
    x, w, b = torch.randn(*xs), torch.randn(N, xs[-1]), torch.randn(N)
    g = torch.randn(*xs[:-1], N)
    ref = [t.clone().requires_grad_(True) for t in (x, w, b)]
    with torch.enable_grad():
        ((ref[0] @ ref[1].t() + ref[2]) * g).sum().backward()
    got = [t.to(dev).requires_grad_(True) for t in (x, w, b)]
    with torch.enable_grad():
        (ag.linear(*got) * g.to(dev)).sum().backward()
    for a, r in zip(got, ref):
        assert md(a.grad, r.grad) <= 2e-5


def test_cosine_backward(dev):
    torch.manual_seed(2)
    im, s, g = torch.randn(13, 48), torch.randn(29, 48), torch.randn(13, 29)
    a, b = im.clone().requires_grad_(True), s.clone().requires_grad_(True)
    with torch.enable_grad():
        ((a @ b.t()) * g).sum().backward()
    ad, bd = im.to(dev).requires_grad_(True), s.to(dev).requires_grad_(True)
    with torch.enable_grad():
        (ag.cosine_scores(ad, bd) * g.to(dev)).sum().backward()
    assert md(ad.grad, a.grad) <= 2e-5 and md(bd.grad, b.grad) <= 2e-5


def _gru_weights(V, E, D, bi, seed):
    torch.manual_seed(seed)
    w = {'embed.weight': torch.empty(V, E).uniform_(-0.1, 0.1)}
    k = 1.0 / D ** 0.5
    for suf in ([''] + (['_reverse'] if bi else [])):
        w['rnn.weight_ih_l0' + suf] = torch.empty(3 * D, E).uniform_(-k, k)
        w['rnn.weight_hh_l0' + suf] = torch.empty(3 * D, D).uniform_(-k, k)
        w['rnn.bias_ih_l0' + suf] = torch.empty(3 * D).uniform_(-k, k)
        w['rnn.bias_hh_l0' + suf] = torch.empty(3 * D).uniform_(-k, k)
    return w


@pytest.mark.parametrize("bi", [False, True])
@pytest.mark.parametrize("last", [False, True])
def test_gru_backward_vs_oracle_autograd(dev, bi, last):
    rng = np.random.RandomState(3)
    V, E, D, B = 40, 12, 32, 9
    lens = sorted([int(x) for x in rng.randint(1, 8, size=B)], reverse=True)
    ids = torch.zeros(B, max(lens), dtype=torch.long)
    for b, l in enumerate(lens):
        ids[b, :l] = T(rng.randint(0, V, size=l))
    w = _gru_weights(V, E, D, bi, 4)
    # ---- oracle under autograd
    wl = {k: v.clone().requires_grad_(True) for k, v in w.items()}
    with torch.enable_grad():
        cap, _ = O.encoder_text(ids, lens, wl, bi, False, False, 'VSE++' if last else None)
        gsel = torch.randn(cap.shape, generator=torch.Generator().manual_seed(5))
        if not last:
            for b, l in enumerate(lens):
                gsel[b, l:] = 0
        (cap * gsel).sum().backward()
    # ---- HIP
    wd = {k: v.to(dev).requires_grad_(True) for k, v in w.items()}
    from itr_amd.modalmodule.TextEncoder import pack_tokens
    toks, off, lens2, mask = pack_tokens(ids.to(dev), lens)
    with torch.enable_grad():
        seq = ag.gru_sequence(toks, off, lens2, wd['embed.weight'], {k[4:]: v for k, v in wd.items() if k.startswith('rnn.')}, bi)
        if last:
            seq = ag.gather_rows(seq, off + torch.as_tensor(lens2, device=dev) - 1)
            out = ag.l2norm_rows(seq)
            gp = gsel.to(dev)
        else:
            out = ag.l2norm_rows(seq)
            gp = gsel.to(dev)[mask[:, :max(lens)]]
        (out * gp).sum().backward()
    want = cap if last else cap[mask.cpu()[:, :max(lens)]]
    assert md(out, want) <= 2e-6
    for k in w:
        assert md(wd[k].grad, wl[k].grad) <= 5e-6, k


NORMS = ['clipped_l2norm', 'l2norm', 'no_norm', 'clipped', 'softmax', 'l1norm', 'clipped_l1norm']
AGGS = ['LogSumExp', 'Mean', 'Sum', 'Max']


@pytest.mark.parametrize("xa", ['t2i', 'i2t'])
@pytest.mark.parametrize("norm", NORMS)
@pytest.mark.parametrize("agg", AGGS)
def test_scan_backward_vs_oracle_autograd(dev, norm, agg, xa):
    rng = np.random.RandomState(6)
    torch.manual_seed(6)
    Bi, Bc, D = 5, 7, 64
    lens = [int(x) for x in rng.randint(1, 10, size=Bc)]         # any order
    L = max(lens)
    img = O.l2norm(torch.randn(Bi, 36, D), -1)
    cap = torch.randn(Bc, L, D) * 0.6
    gS = torch.randn(Bi, Bc)
    a, c = img.clone().requires_grad_(True), cap.clone().requires_grad_(True)
    with torch.enable_grad():
        S = O.xattn_score(a, c, lens, xa, norm, agg, 6.0, 9.0)
        (S * gS).sum().backward()
    off = np.concatenate([[0], np.cumsum(lens)[:-1]])
    words = torch.cat([cap[k, :lens[k]] for k in range(Bc)], 0)
    ad, wd = img.to(dev).requires_grad_(True), words.to(dev).requires_grad_(True)
    with torch.enable_grad():
        Sd = (ag.scan_t2i_scores if xa == 't2i' else ag.scan_i2t_scores)(ad, wd, off, lens, norm, agg, 6.0, 9.0)
        (Sd * gS.to(dev)).sum().backward()
    assert md(Sd, S) <= 2e-5
    want_w = torch.cat([c.grad[k, :lens[k]] for k in range(Bc)], 0)
    scale = max(1.0, float(a.grad.abs().max()), float(want_w.abs().max()))
    assert md(ad.grad, a.grad) <= 2e-5 * scale
    assert md(wd.grad, want_w) <= 2e-5 * scale


@pytest.mark.parametrize("xa", ['t2i', 'i2t'])
@pytest.mark.parametrize("R,max_len", [(30, 12), (1, 5), (35, 80), (49, 20), (100, 64), (100, 96), (37, 70)])
def test_scan_any_region_count(dev, xa, R, max_len):
    """images with a region count other than the reference's 36 (csrc/scan_train*.hip: the FIXED = false
    instantiations, 1..36 regions in the 36-region LDS block, 37..100 in the 100-region one).  Scores and gradients against the
    oracle under autograd, all first norms x two aggregations.  The t2i backward keeps one more regions x words block: more than
    36 regions with captions of more than 64 words is the one combination that does not fit a CU's LDS (NotImplementedError)."""
    rng = np.random.RandomState(R)
    torch.manual_seed(R)
    Bi, Bc, D = 3, 5, 64
    lens = [max_len] + [int(x) for x in rng.randint(1, max_len + 1, size=Bc - 1)]
    img = O.l2norm(torch.randn(Bi, R, D), -1)
    cap = torch.randn(Bc, max_len, D) * 0.6
    gS = torch.randn(Bi, Bc)
    off = np.concatenate([[0], np.cumsum(lens)[:-1]])
    words = torch.cat([cap[k, :lens[k]] for k in range(Bc)], 0)
    fn = ag.scan_t2i_scores if xa == 't2i' else ag.scan_i2t_scores
    for norm, agg in [('clipped_l2norm', 'LogSumExp'), ('softmax', 'Mean'), ('l2norm', 'Max'), ('clipped_l1norm', 'Sum'), ('no_norm', 'LogSumExp')]:
        a, c = img.clone().requires_grad_(True), cap.clone().requires_grad_(True)
        with torch.enable_grad():
            S = O.xattn_score(a, c, lens, xa, norm, agg, 6.0, 9.0)
            (S * gS).sum().backward()
        ad, wd = img.to(dev).requires_grad_(True), words.to(dev).requires_grad_(True)
        with torch.enable_grad():
            Sd = fn(ad, wd, off, lens, norm, agg, 6.0, 9.0)
        tol = 2e-5 * (max(R, max_len) if agg == 'Sum' else 1)
        assert md(Sd, S) <= tol, (norm, agg)
        if xa == 't2i' and R > 36 and max_len > 64:
            with pytest.raises(NotImplementedError):
                (Sd * gS.to(dev)).sum().backward()
            continue
        (Sd * gS.to(dev)).sum().backward()
        want_w = torch.cat([c.grad[k, :lens[k]] for k in range(Bc)], 0)
        scale = max(1.0, float(a.grad.abs().max()), float(want_w.abs().max()))
        assert md(ad.grad, a.grad) <= 2e-5 * scale, (norm, agg)
        assert md(wd.grad, want_w) <= 2e-5 * scale, (norm, agg)
    with pytest.raises(NotImplementedError):
        fn(torch.zeros(2, 101, D, device=dev), wd.detach(), off, lens)


@pytest.mark.parametrize("xa", ['t2i', 'i2t'])
def test_scan_train_long_captions(dev, xa):
    """Training batches may hold captions of up to 96 tokens (the longest Flickr30k caption has 82)."""
    torch.manual_seed(11)
    Bi, D = 3, 32
    lens = [96, 70, 3]
    img = O.l2norm(torch.randn(Bi, 36, D), -1)
    cap = torch.randn(len(lens), max(lens), D) * 0.5
    gS = torch.randn(Bi, len(lens))
    a, c = img.clone().requires_grad_(True), cap.clone().requires_grad_(True)
    with torch.enable_grad():
        S = O.xattn_score(a, c, lens, xa, 'clipped_l2norm', 'LogSumExp', 6.0, 9.0)
        (S * gS).sum().backward()
    off = np.concatenate([[0], np.cumsum(lens)[:-1]])
    words = torch.cat([cap[k, :lens[k]] for k in range(len(lens))], 0)
    ad, wd = img.to(dev).requires_grad_(True), words.to(dev).requires_grad_(True)
    fn = ag.scan_t2i_scores if xa == 't2i' else ag.scan_i2t_scores
    with torch.enable_grad():
        Sd = fn(ad, wd, off, lens, 'clipped_l2norm', 'LogSumExp', 6.0, 9.0)
        (Sd * gS.to(dev)).sum().backward()
    assert md(Sd, S) <= 2e-5
    want_w = torch.cat([c.grad[k, :lens[k]] for k in range(len(lens))], 0)
    scale = max(1.0, float(a.grad.abs().max()), float(want_w.abs().max()))
    assert md(ad.grad, a.grad) <= 2e-5 * scale and md(wd.grad, want_w) <= 2e-5 * scale
    with pytest.raises(NotImplementedError):
        fn(ad, torch.zeros(97, D, device=dev), [0], [97])


def test_scan_train_rejects(dev):
    img, words = torch.zeros(2, 36, 32, device=dev), torch.zeros(5, 32, device=dev)
    with pytest.raises(ValueError):
        ag.scan_t2i_scores(img, words, [0, 2], [2, 3], 'bogus')
    with pytest.raises(RuntimeError):
        ag.linear(torch.zeros(2, 3), torch.zeros(4, 3))
    with pytest.raises(ValueError):
        ag.linear(torch.zeros(2, 3, device=dev), torch.zeros(4, 5, device=dev))


# ------------------------------------------------------------------------------------------ optimizer
def test_adam_and_clip_kernels(dev):
    torch.manual_seed(7)
    ps = [torch.randn(300, 17), torch.randn(5), torch.randn(64, 64)]
    ref = [p.clone().requires_grad_(True) for p in ps]
    got = [p.to(dev).requires_grad_(True) for p in ps]
    opt_r = torch.optim.Adam(ref, lr=3e-3)
    opt_g = ag.Adam(got, lr=3e-3)
    for it in range(3):
        gs = [torch.randn_like(p) * (5.0 if it == 0 else 0.05) for p in ps]      # first step is clipped, later ones are not
        for r, g_, gg in zip(ref, got, gs):
            r.grad, g_.grad = gg.clone(), gg.to(dev)
        total = torch.nn.utils.clip_grad_norm_(ref, 2.0)
        opt_r.step()
        opt_g.step(max_norm=2.0)
        assert abs(float(opt_g.last_grad_norm[0]) - float(total)) <= 1e-4 * float(total)
        for r, g_ in zip(ref, got):
            assert md(g_, r) <= 2e-6


# ------------------------------------------------------------------------------------------ model.train_emb
def _load(model, g, pre_img, pre_txt):
    model.img_enc.load_state_dict({k[len(pre_img):]: T(g[k]) for k in g.files if k.startswith(pre_img)})
    model.txt_enc.load_state_dict({k[len(pre_txt):]: T(g[k]) for k in g.files if k.startswith(pre_txt)}, strict=True)


def _check_step(model, g, tag_grad, tag_img, tag_txt, lr):
    coef = min(1.0, 2.0 / (float(model.optimizer.last_grad_norm[0]) + 1e-6))
    named = [('txt.' + n, p) for n, p in model.txt_enc.named_parameters()] + [('img.' + n, p) for n, p in model.img_enc.named_parameters()]
    for n, p in named:
        want = T(g[tag_grad + n])
        assert md(p.grad * coef, want) <= 5e-6 * max(1.0, float(want.abs().max())), n
    # parameters: Adam divides by sqrt(v) + 1e-8, so entries whose gradient is ~1e-7 amplify rounding noise up to ~lr
    for k, v in model.img_enc.state_dict().items():
        d = (v.cpu() - T(g[tag_img + k])).abs()
        assert float(d.max()) <= lr and float(d.mean()) <= 2e-6, k
    for k, v in model.txt_enc.state_dict().items():
        d = (v.cpu() - T(g[tag_txt + k])).abs()
        assert float(d.max()) <= lr and float(d.mean()) <= 2e-6, k


def test_scan_train_emb_matches_reference(golden, dev):
    g = golden("g15_train_step")
    cfg_ref = json.loads(bytes(g["cfg_json"]).decode())
    cfg = C.build_config(['with', 'SCAN', 'data_name=f30k_precomp'])
    cfg.update({k: v for k, v in cfg_ref.items() if k != "name"})
    model = get_model(cfg)
    _load(model, g, 'w0_img_', 'w0_txt_')
    model.train_start()
    from itr_amd.metricmodule.evaluation import LogCollector
    model.logger = LogCollector()
    for step in (1, 2):
        lens = [int(x) for x in g["s%d_lens" % step]]
        batch = (T(g["s%d_feats" % step]), None, None, T(g["s%d_ids" % step]), lens, list(range(len(lens))), None, None)
        model.train_emb(batch)
        assert model.Eiters == step
        assert abs(float(model.logger.meters['Loss'].val) - float(g["s%d_loss" % step])) <= 1e-4     # north_star: loss within 1e-4
        _check_step(model, g, 's%d_grad_' % step, 's%d_img_' % step, 's%d_txt_' % step, cfg['learning_rate'])


def test_scan_trains_through_the_module_seams(golden, dev):
    """One module, two modes: the step composed the way the reference's train_emb composes it (Models.py:182-225) --
    model.img_enc(images), model.txt_enc(captions, lengths) (padded output), model.criterion(img, cap, lens), backward,
    clip + Adam -- reproduces G15 (the reference's own two steps) exactly like model.train_emb does."""
    g = golden("g15_train_step")
    cfg_ref = json.loads(bytes(g["cfg_json"]).decode())
    cfg = C.build_config(['with', 'SCAN', 'data_name=f30k_precomp'])
    cfg.update({k: v for k, v in cfg_ref.items() if k != "name"})
    model = get_model(cfg)
    _load(model, g, 'w0_img_', 'w0_txt_')
    model.train_start()
    for step in (1, 2):
        lens = [int(x) for x in g["s%d_lens" % step]]
        images, captions = T(g["s%d_feats" % step]).cuda(), T(g["s%d_ids" % step]).cuda()
        model.optimizer.zero_grad()
        img_emb = model.img_enc(images)
        cap_emb, cap_lens = model.txt_enc(captions, lens)
        assert img_emb.requires_grad and cap_emb.requires_grad and cap_emb.shape == (len(lens), max(lens), cfg['embed_size'])
        loss = model.criterion(img_emb, cap_emb, cap_lens)
        assert abs(float(loss) - float(g["s%d_loss" % step])) <= 1e-4
        loss.backward()
        model.optimizer.step(max_norm=model.grad_clip)
        _check_step(model, g, 's%d_grad_' % step, 's%d_img_' % step, 's%d_txt_' % step, cfg['learning_rate'])
    # the two modes of one module agree: training mode = tape (differentiable), evaluation mode = fused kernels
    t_img = model.img_enc(images)
    t_cap, _ = model.txt_enc(captions, lens)
    model.val_start()
    e_img = model.img_enc(images)
    e_cap, _ = model.txt_enc(captions, lens)
    assert t_img.requires_grad and not e_img.requires_grad and not e_cap.requires_grad
    assert md(t_img, e_img) <= 2e-6 and md(t_cap, e_cap) <= 5e-6


def test_scan_i2t_train_emb_vs_oracle_step(golden, dev):
    """cross_attn='i2t': one train_emb step from the G15 weights against the oracle's restated step (the oracle's step is
    pinned by the reference's own train_emb for t2i, its i2t similarity by G5)."""
    g = golden("g15_train_step")
    cfg_ref = json.loads(bytes(g["cfg_json"]).decode())
    cfg = C.build_config(['with', 'SCAN', 'data_name=f30k_precomp'])
    cfg.update({k: v for k, v in cfg_ref.items() if k != "name"})
    cfg.update(cross_attn='i2t', lambda_softmax=4.0, lambda_lse=5.0, agg_func='Mean')
    model = get_model(cfg)
    _load(model, g, 'w0_img_', 'w0_txt_')
    model.train_start()
    from itr_amd.metricmodule.evaluation import LogCollector
    model.logger = LogCollector()
    lens = [int(x) for x in g["s1_lens"]]
    feats, ids = T(g["s1_feats"]), T(g["s1_ids"])
    wi = {k[len('w0_img_'):]: T(g[k]) for k in g.files if k.startswith('w0_img_')}
    wt = {k[len('w0_txt_'):]: T(g[k]) for k in g.files if k.startswith('w0_txt_')}
    o_loss, o_grads, nwi, nwt, _ = O.gru_model_train_step('SCAN', wi, wt, feats, ids, lens, cfg, None)
    model.train_emb((feats, None, None, ids, lens, list(range(len(lens))), None, None))
    assert abs(float(model.logger.meters['Loss'].val) - float(o_loss)) <= 1e-4
    coef = min(1.0, 2.0 / (float(model.optimizer.last_grad_norm[0]) + 1e-6))
    for n, p in [('txt.' + n, p) for n, p in model.txt_enc.named_parameters()] + [('img.' + n, p) for n, p in model.img_enc.named_parameters()]:
        assert md(p.grad * coef, o_grads[n]) <= 5e-6 * max(1.0, float(o_grads[n].abs().max())), n


def test_vsepp_train_emb_matches_reference_components(golden, dev):
    g = golden("g15_train_step")
    cfg = C.build_config(['with', 'VSE_PP', 'data_name=f30k_precomp', 'max_violation=False', 'bi_gru=False', 'learning_rate=0.001'])
    cfg.update(img_dim=24, embed_size=32, word_dim=16, vocab_size=60)
    model = get_model(cfg)
    _load(model, g, 'v_w0_img_', 'v_w0_txt_')
    model.train_start()
    from itr_amd.metricmodule.evaluation import LogCollector
    model.logger = LogCollector()
    lens = [int(x) for x in g["v_lens"]]
    model.train_emb((T(g["v_feats"]), None, None, T(g["v_ids"]), lens, list(range(len(lens))), None, None))
    assert abs(float(model.logger.meters['Loss'].val) - float(g["v_loss"])) <= 1e-4
    _check_step(model, g, 'v_grad_', 'v_s1_img_', 'v_s1_txt_', cfg['learning_rate'])


def test_every_wrapper_has_an_optimizer_and_trains(dev):
    """All six model families train (G15 / G18 / G19 / G20 / G21 replay the reference's own train_emb); what is refused are the
    raw-image towers (torchvision CNNs: out of scope)."""
    cfg = C.build_config(['with', 'VSRN', 'data_name=f30k_precomp'])
    cfg.update(img_dim=16, embed_size=32, word_dim=8, vocab_size=20, dim_vid=32, dim_hidden=8, dim_word=6)
    model = get_model(cfg)
    assert model.optimizer is not None and model.optimizer.param_groups[0]['lr'] == cfg['learning_rate']
    assert len(model.state_dict()) == 2                       # the captioning model is not part of the checkpoint (Models.py:37-45)
    with pytest.raises(NotImplementedError):
        get_model(dict(cfg, data_name='f30k'))
