"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the encode -> score -> loss / rank hot path.

This file is a CPU (torch fp32 / numpy) *restatement* of the reference algorithm
(WangFei-2019/Image-text-Retrieval), written from the maths, each function citing the
reference file:line it follows (paths relative to the reference root).  It is the checker
for the HIP path; it is never the thing shipped or measured:

  * only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py`
    may import it;
  * the product package (`image-text-retrieval_amd/itr_amd`) never imports it and has no
    CPU fallback.

Parity pinning: the reference ships no tests or golden vectors, so the
oracle is pinned against outputs of the reference itself, generated in the build container by
`oracle/make_goldens.py` (which imports /root/reference under `oracle/ref_shim.py`) and
committed as `tests/golden/*.npz`.  `tests/test_oracle_golden.py` re-checks every one of
them on every CPU test run.

Third-party arithmetic: the reference's GRU is `torch.nn.GRU` (cuDNN / ATen); its published
cell equations (PyTorch docs, gate order r,z,n) are restated in `gru_direction` below.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------------------
# a1  norms  (itr/modalmodule/utils.py:4-15)
# --------------------------------------------------------------------------------------


def l2norm(x, dim=1, eps=1e-8):
    """x / (sqrt(sum x^2) + eps): eps is added AFTER the sqrt (utils.py:11-15)."""
    return x / (x.pow(2).sum(dim=dim, keepdim=True).sqrt() + eps)


def l1norm(x, dim=1, eps=1e-8):
    """x / (sum |x| + eps) (utils.py:4-8)."""
    return x / (x.abs().sum(dim=dim, keepdim=True) + eps)


# --------------------------------------------------------------------------------------
# a2  image tower  (itr/modalmodule/ImgEncoder.py:133-147)
# --------------------------------------------------------------------------------------


def encoder_image_precomp(images, fc_weight, fc_bias, no_imgnorm=False, use_abs=False):
    """l2norm(images @ W^T + b, dim=-1) [abs].  images (..., F) -> (..., D)."""
    feat = images @ fc_weight.t() + fc_bias
    if not no_imgnorm:
        feat = l2norm(feat, dim=-1)
    if use_abs:
        feat = feat.abs()
    return feat


# --------------------------------------------------------------------------------------
# f4  VSRN image tower: fc -> l2norm -> 4 x Rs_GCN -> l2norm -> GRU over the regions -> last hidden state -> [BN] -> l2norm
#     (itr/modalmodule/ImgEncoder.py:166-231, itr/modalmodule/vsrn_.py:50-71)
# --------------------------------------------------------------------------------------


def rs_gcn(w, p, v):
    """Rs_GCN.forward on v (B, N, D) (the reference works on the (B, D, N) transpose; a Conv1d with kernel size 1 is a
    Linear over the channel axis): R = theta(v) phi(v)^T / N, y = R g(v), v* = BN(W y) + v  (vsrn_.py:50-71)."""
    def conv(name, x):
        return x @ w[p + name + '.weight'][:, :, 0].t() + w[p + name + '.bias']
    g_v, theta_v, phi_v = conv('g', v), conv('theta', v), conv('phi', v)
    R = theta_v @ phi_v.transpose(1, 2)
    y = (R / R.shape[-1]) @ g_v
    wy = conv('W.0', y)
    return _bn_eval(wy, w, p + 'W.1', channel_dim=2) + v


def vsrn_image(w, images, data_name='coco_precomp', no_imgnorm=False, use_abs=False):
    """EncoderImagePrecompAttn.forward (ImgEncoder.py:199-231) -> (features (B, D), GCN_img_emd (B, N, D)).
    NB both l2norm calls on the region tensors use the reference's DEFAULT dim=1: they normalise ACROSS the regions
    (utils.py:11), not across the features -- restated as written."""
    x = images @ w['fc.weight'].t() + w['fc.bias']
    if data_name != 'f30k_precomp':
        x = l2norm(x, dim=1)
    for i in (1, 2, 3, 4):
        x = rs_gcn(w, 'Rs_GCN_%d.' % i, x)
    gcn = l2norm(x, dim=1)
    B, N, _ = gcn.shape
    seq = gru_direction(gcn, [N] * B, w['img_rnn.weight_ih_l0'], w['img_rnn.weight_hh_l0'], w['img_rnn.bias_ih_l0'],
                        w['img_rnn.bias_hh_l0'])
    feat = seq[:, N - 1]
    if data_name == 'f30k_precomp':
        feat = _bn_eval(feat, w, 'bn', channel_dim=1)
    if not no_imgnorm:
        feat = l2norm(feat, dim=1)
    if use_abs:
        feat = feat.abs()
    return feat, gcn


# --------------------------------------------------------------------------------------
# a3  text tower: Embedding -> packed (bi)GRU -> dir-average -> [last step] -> [l2norm]
#     (itr/modalmodule/TextEncoder.py:38-70)
# --------------------------------------------------------------------------------------


def gru_direction(x, lengths, w_ih, w_hh, b_ih, b_hh, reverse=False):
    """One direction of a 1-layer GRU over padded x (B, L, E) with packed-sequence semantics:
    sample b only advances for t < lengths[b]; outputs at t >= lengths[b] are 0.
    Gate order (r, z, n); n = tanh(W_in x + b_in + r * (W_hn h + b_hn));
    h' = (1 - z) * n + z * h   (torch.nn.GRU, used at TextEncoder.py:30,48)."""
    B, L, _ = x.shape
    H = w_hh.shape[1]
    out = x.new_zeros(B, L, H)
    h = x.new_zeros(B, H)
    gi_all = x @ w_ih.t() + b_ih  # (B, L, 3H)
    lens = torch.as_tensor(lengths, dtype=torch.long)
    steps = range(L - 1, -1, -1) if reverse else range(L)
    for t in steps:
        active = (lens > t)
        if not bool(active.any()):
            continue
        gi = gi_all[:, t]
        gh = h @ w_hh.t() + b_hh
        r = torch.sigmoid(gi[:, :H] + gh[:, :H])
        z = torch.sigmoid(gi[:, H:2 * H] + gh[:, H:2 * H])
        n = torch.tanh(gi[:, 2 * H:] + r * gh[:, 2 * H:])
        h_new = (1.0 - z) * n + z * h
        m = active.unsqueeze(1)
        h = torch.where(m, h_new, h)
        out[:, t] = torch.where(m, h_new, torch.zeros_like(h_new))
    return out


def encoder_text(ids, lengths, weights, use_bi_gru=False, no_txtnorm=False, use_abs=False,
                 method_name=None):
    """weights: dict with the reference's state_dict names ('embed.weight',
    'rnn.weight_ih_l0', 'rnn.weight_hh_l0', 'rnn.bias_ih_l0', 'rnn.bias_hh_l0' and the
    '_reverse' variants).  Dropout is identity in eval mode.  Returns (cap_emb, cap_len)
    with cap_emb (B, max(lengths), D), or (B, D) when method_name in {VSE++, VSRN}
    (TextEncoder.py:57-60: gather at len-1)."""
    lengths = [int(l) for l in lengths]
    L = max(lengths)
    x = weights['embed.weight'][ids[:, :L]]
    fwd = gru_direction(x, lengths, weights['rnn.weight_ih_l0'], weights['rnn.weight_hh_l0'],
                        weights['rnn.bias_ih_l0'], weights['rnn.bias_hh_l0'])
    if use_bi_gru:
        bwd = gru_direction(x, lengths, weights['rnn.weight_ih_l0_reverse'],
                            weights['rnn.weight_hh_l0_reverse'],
                            weights['rnn.bias_ih_l0_reverse'],
                            weights['rnn.bias_hh_l0_reverse'], reverse=True)
        cap = (fwd + bwd) / 2  # TextEncoder.py:54-55
    else:
        cap = fwd
    if method_name in ('VSE++', 'VSRN'):
        idx = torch.as_tensor(lengths, dtype=torch.long) - 1
        cap = cap[torch.arange(cap.shape[0]), idx]
    if not no_txtnorm:
        cap = l2norm(cap, dim=-1)
    if use_abs:
        cap = cap.abs()
    return cap, torch.as_tensor(lengths, dtype=torch.long)


# --------------------------------------------------------------------------------------
# a4 / a9 / a8  pooled similarities
# --------------------------------------------------------------------------------------


def cosine_sim(im, s):
    """im @ s^T, inputs already normalised (Objectives.py:18-21)."""
    return im @ s.t()


def order_sim(im, s):
    """-|| max(0, s - im) ||_2 for every pair (Objectives.py:24-30). -> (n_im, n_s)."""
    d = (s.unsqueeze(1) - im.unsqueeze(0)).clamp(min=0)
    return -d.pow(2).sum(2).sqrt().t()


def pdist(x1, x2):
    """SAEM euclidean distance, its "similarity" under measure='order' (Objectives.py:297-307)."""
    x1_square = torch.sum(x1 * x1, 1).view(-1, 1)
    x2_square = torch.sum(x2 * x2, 1).view(1, -1)
    return torch.sqrt(x1_square - 2 * torch.mm(x1, x2.t()) + x2_square + 1e-4)


def pdist_cos(x1, x2):
    """rows renormalised by their plain L2 norm (no eps), mm, NaN -> 0 (Objectives.py:310-323)."""
    a = x1 / x1.norm(dim=1)[:, None]
    b = x2 / x2.norm(dim=1)[:, None]
    res = a @ b.t()
    return torch.where(torch.isnan(res), torch.zeros_like(res), res)


def multi_view_matching(imgs, caps):
    """max over the k views of img[i, v] . cap[c]  (Fusionmodule.py:674-692). Both branches of
    the reference (square batched matmul / per-caption loop) compute the same thing."""
    return torch.einsum('ivd,cd->ivc', imgs, caps).max(1)[0]


# --------------------------------------------------------------------------------------
# a5 / a10  bidirectional hinge (Objectives.py:76-115, 492-517)
# --------------------------------------------------------------------------------------


def hinge_loss(scores, margin=0.2, max_violation=False):
    """sum_i red_j!=i [m + S_ij - S_ii]_+  +  sum_j red_i!=j [m + S_ij - S_jj]_+ with
    red = max (max_violation) or sum; the diagonal is zeroed before the reduction."""
    n = scores.shape[0]
    diag = scores.diag().view(n, 1)
    cost_s = (margin + scores - diag).clamp(min=0)
    cost_im = (margin + scores - diag.t()).clamp(min=0)
    eye = torch.eye(n, dtype=torch.bool)
    cost_s = cost_s.masked_fill(eye, 0)
    cost_im = cost_im.masked_fill(eye, 0)
    if max_violation:
        return cost_s.max(1)[0].sum() + cost_im.max(0)[0].sum()
    return cost_s.sum() + cost_im.sum()


def hinge_loss_and_grad(scores, margin=0.2, max_violation=False):
    with torch.enable_grad():
        s = scores.detach().clone().requires_grad_(True)
        loss = hinge_loss(s, margin, max_violation)
        loss.backward()
    return loss.detach(), s.grad.detach()


# --------------------------------------------------------------------------------------
# a6  SCAN cross attention (Objectives.py:329-476, cosine_similarity :10-15)
# --------------------------------------------------------------------------------------

_LEAKY = 0.1


def _raw_feature_norm(attn, kind):
    """attn: (batch, sourceL, queryL); every norm acts along queryL (dim 2)
    (Objectives.py:436-457).  'l1norm'/'clipped_l1norm' call an undefined `l1norm_d` in the
    reference (NameError); the evident intent l1norm(attn, 2) is implemented."""
    if kind == 'softmax':
        return torch.softmax(attn, dim=2)
    if kind == 'l2norm':
        return l2norm(attn, 2)
    if kind == 'clipped_l2norm':
        return l2norm(F.leaky_relu(attn, _LEAKY), 2)
    if kind == 'l1norm':
        return l1norm(attn, 2)
    if kind == 'clipped_l1norm':
        return l1norm(F.leaky_relu(attn, _LEAKY), 2)
    if kind == 'clipped':
        return F.leaky_relu(attn, _LEAKY)
    if kind == 'no_norm':
        return attn
    raise ValueError("unknown first norm type:", kind)


def func_attention(query, context, raw_feature_norm, smooth):
    """query (B, qL, d), context (B, sL, d) -> weighted context (B, qL, d), attn (B, qL, sL)
    (Objectives.py:421-476)."""
    attn = torch.bmm(context, query.transpose(1, 2))          # (B, sL, qL)
    attn = _raw_feature_norm(attn, raw_feature_norm)
    attn = torch.softmax(attn.transpose(1, 2) * smooth, dim=2)  # (B, qL, sL): over sL
    return torch.bmm(attn, context), attn


def _cos_rows(x1, x2, eps=1e-8):
    """w12 / clamp(|x1||x2|, min=eps) along the last dim (Objectives.py:10-15)."""
    w12 = (x1 * x2).sum(-1)
    return w12 / (x1.norm(2, -1) * x2.norm(2, -1)).clamp(min=eps)


def _aggregate(row_sim, agg_func, lambda_lse):
    """row_sim (B, n) -> (B,) (Objectives.py:355-366). LSE has no max-shift."""
    if agg_func == 'LogSumExp':
        return torch.log(torch.exp(row_sim * lambda_lse).sum(1)) / lambda_lse
    if agg_func == 'Max':
        return row_sim.max(1)[0]
    if agg_func == 'Sum':
        return row_sim.sum(1)
    if agg_func == 'Mean':
        return row_sim.mean(1)
    raise ValueError("unknown aggfunc: {}".format(agg_func))


def xattn_score(images, captions, cap_lens, cross_attn='t2i', raw_feature_norm='clipped_l2norm',
                agg_func='LogSumExp', lambda_lse=6.0, lambda_softmax=9.0):
    """images (Ni, R, d), captions (Nc, L, d) padded, cap_lens (Nc,) -> (Ni, Nc).
    t2i: words attend over regions (Objectives.py:329-372); i2t: regions attend over words
    (:376-417).  One caption at a time, like the reference."""
    if cross_attn not in ('t2i', 'i2t'):
        raise ValueError("unknown cross_attn:", cross_attn)
    n_image = images.shape[0]
    cols = []
    for c in range(captions.shape[0]):
        w = int(cap_lens[c])
        cap = captions[c, :w].unsqueeze(0).expand(n_image, w, -1)
        if cross_attn == 't2i':
            ctx, _ = func_attention(cap, images, raw_feature_norm, lambda_softmax)
            row_sim = _cos_rows(cap, ctx)            # (Ni, w)
        else:
            ctx, _ = func_attention(images, cap, raw_feature_norm, lambda_softmax)
            row_sim = _cos_rows(images, ctx)         # (Ni, R)
        cols.append(_aggregate(row_sim, agg_func, lambda_lse))
    return torch.stack(cols, 1)


# --------------------------------------------------------------------------------------
# a7  SGRAF similarity (Fusionmodule.py:373-664), eval mode (BN running stats, no dropout)
# --------------------------------------------------------------------------------------


def _linear(x, w, p):
    return x @ w[p + '.weight'].t() + w[p + '.bias']


def _bn_eval(x, w, p, channel_dim, eps=1e-5):
    shape = [1] * x.dim()
    shape[channel_dim] = -1
    mean = w[p + '.running_mean'].view(shape)
    var = w[p + '.running_var'].view(shape)
    return (x - mean) / torch.sqrt(var + eps) * w[p + '.weight'].view(shape) + w[p + '.bias'].view(shape)


def sgraf_visual_sa(w, local, raw_global):
    """VisualSA.forward (Fusionmodule.py:491-507). local (B, 36, D), raw_global (B, D)."""
    l_emb = torch.tanh(_bn_eval(_linear(local, w, 'v_global_w.embedding_local.0'), w,
                                'v_global_w.embedding_local.1', 1))
    g_emb = torch.tanh(_bn_eval(_linear(raw_global, w, 'v_global_w.embedding_global.0'), w,
                                'v_global_w.embedding_global.1', 1))
    common = l_emb * g_emb.unsqueeze(1)
    weights = torch.softmax(_linear(common, w, 'v_global_w.embedding_common.0').squeeze(2), dim=1)
    return l2norm((weights.unsqueeze(2) * local).sum(1), dim=-1)


def sgraf_text_sa(w, local, raw_global):
    """TextSA.forward (Fusionmodule.py:543-559). local (1, W, D), raw_global (1, D)."""
    l_emb = torch.tanh(_linear(local, w, 't_global_w.embedding_local.0'))
    g_emb = torch.tanh(_linear(raw_global, w, 't_global_w.embedding_global.0'))
    common = l_emb * g_emb.unsqueeze(1)
    weights = torch.softmax(_linear(common, w, 't_global_w.embedding_common.0').squeeze(2), dim=1)
    return l2norm((weights.unsqueeze(2) * local).sum(1), dim=-1)


def sgraf_scan_attention(query, context, smooth=9.0):
    """SCAN_attention (Fusionmodule.py:632-664): clipped_l2norm attention, softmax over the
    context axis, weighted context then l2-normalised."""
    attn = torch.bmm(context, query.transpose(1, 2))
    attn = l2norm(F.leaky_relu(attn, _LEAKY), 2)
    attn = torch.softmax(attn.transpose(1, 2) * smooth, dim=2)
    return l2norm(torch.bmm(attn, context), dim=-1)


def sgraf_similarity(w, img_emb, cap_emb, cap_lens, module_name='SAF', sgr_step=3):
    """EncoderSimilarity.forward (Fusionmodule.py:406-451). w: dict of the module's state_dict
    tensors.  -> (Ni, Nc) in (0, 1)."""
    n_image = img_emb.shape[0]
    img_glo = sgraf_visual_sa(w, img_emb, img_emb.mean(1))
    cols = []
    for c in range(cap_emb.shape[0]):
        nw = int(cap_lens[c])
        cap_i = cap_emb[c, :nw].unsqueeze(0)
        cap_exp = cap_i.expand(n_image, nw, -1)
        cap_glo = sgraf_text_sa(w, cap_i, cap_i.mean(1))
        ctx = sgraf_scan_attention(cap_exp, img_emb, smooth=9.0)
        sim_loc = l2norm(_linear((ctx - cap_exp).pow(2), w, 'sim_tranloc_w'), dim=-1)
        sim_glo = l2norm(_linear((img_glo - cap_glo).pow(2), w, 'sim_tranglo_w'), dim=-1)
        sim_emb = torch.cat([sim_glo.unsqueeze(1), sim_loc], 1)     # (Ni, nw+1, s)
        if module_name == 'SGR':
            for k in range(sgr_step):                                # GraphReasoning :581-587
                p = 'SGR_module.sgr%d' % k
                q = _linear(sim_emb, w, p + '.graph_query_w')
                kk = _linear(sim_emb, w, p + '.graph_key_w')
                edge = torch.softmax(torch.bmm(q, kk.transpose(1, 2)), dim=-1)
                sim_emb = torch.relu(_linear(torch.bmm(edge, sim_emb), w, p + '.sim_graph_w'))
            sim_vec = sim_emb[:, 0]
        elif module_name == 'SAF':                                   # AttentionFiltration :615-619
            a = _linear(sim_emb, w, 'SAF_module.attn_sim_w').transpose(1, 2)   # (Ni, 1, nw+1)
            a = l1norm(torch.sigmoid(_bn_eval(a, w, 'SAF_module.bn', 1)), dim=-1)
            sim_vec = l2norm(torch.bmm(a, sim_emb).squeeze(1), dim=-1)
        else:
            raise ValueError('Invalid input of config.module_name in configs.py')
        cols.append(torch.sigmoid(_linear(sim_vec, w, 'sim_eval_w')).squeeze(1))
    return torch.stack(cols, 1)


# --------------------------------------------------------------------------------------
# a7 / a14  SGRAF in TRAINING mode (SGRAF.train_emb, Models.py:518-546): EncoderSimilarity.forward with BatchNorm batch
# statistics (VisualSA: BatchNorm1d(36) over (B, D) per region and BatchNorm1d(D) over B; AttentionFiltration: BatchNorm1d(1)
# over the B x (W + 1) logits of ONE caption, because the reference calls it once per caption), dropout at p = 0 (the
# reference's 0.4 sites draw from torch's generator: the golden G20 switches them off), differentiable (torch autograd).
# Pinned by G20 = the reference's own SGRAF.train_emb: loss and every clipped gradient (tests/test_oracle_golden.py).
# --------------------------------------------------------------------------------------


def _bn_train(x, w, p, channel_dim, eps=1e-5):
    dims = [d for d in range(x.dim()) if d != channel_dim]
    mean = x.mean(dims, keepdim=True)
    var = x.var(dims, unbiased=False, keepdim=True)
    shape = [1] * x.dim()
    shape[channel_dim] = -1
    return (x - mean) / torch.sqrt(var + eps) * w[p + '.weight'].view(shape) + w[p + '.bias'].view(shape)


def sgraf_similarity_train(w, img_emb, cap_emb, cap_lens, module_name='SAF', sgr_step=3):
    """EncoderSimilarity.forward (Fusionmodule.py:406-451) in training mode -> (Ni, Nc).  Same loop over the captions as the reference."""
    n_image = img_emb.shape[0]
    l_emb = torch.tanh(_bn_train(_linear(img_emb, w, 'v_global_w.embedding_local.0'), w, 'v_global_w.embedding_local.1', 1))
    g_emb = torch.tanh(_bn_train(_linear(img_emb.mean(1), w, 'v_global_w.embedding_global.0'), w, 'v_global_w.embedding_global.1', 1))
    weights = torch.softmax(_linear(l_emb * g_emb.unsqueeze(1), w, 'v_global_w.embedding_common.0').squeeze(2), dim=1)
    img_glo = l2norm((weights.unsqueeze(2) * img_emb).sum(1), dim=-1)
    cols = []
    for c in range(cap_emb.shape[0]):
        nw = int(cap_lens[c])
        cap_i = cap_emb[c, :nw].unsqueeze(0)
        cap_exp = cap_i.expand(n_image, nw, -1)
        cap_glo = sgraf_text_sa(w, cap_i, cap_i.mean(1))                 # TextSA has no BatchNorm
        ctx = sgraf_scan_attention(cap_exp, img_emb, smooth=9.0)
        sim_loc = l2norm(_linear((ctx - cap_exp).pow(2), w, 'sim_tranloc_w'), dim=-1)
        sim_glo = l2norm(_linear((img_glo - cap_glo).pow(2), w, 'sim_tranglo_w'), dim=-1)
        sim_emb = torch.cat([sim_glo.unsqueeze(1), sim_loc], 1)
        if module_name == 'SGR':
            for k in range(sgr_step):
                p = 'SGR_module.sgr%d' % k
                q = _linear(sim_emb, w, p + '.graph_query_w')
                kk = _linear(sim_emb, w, p + '.graph_key_w')
                edge = torch.softmax(torch.bmm(q, kk.transpose(1, 2)), dim=-1)
                sim_emb = torch.relu(_linear(torch.bmm(edge, sim_emb), w, p + '.sim_graph_w'))
            sim_vec = sim_emb[:, 0]
        elif module_name == 'SAF':
            a = _linear(sim_emb, w, 'SAF_module.attn_sim_w').transpose(1, 2)          # (Ni, 1, nw + 1)
            a = l1norm(torch.sigmoid(_bn_train(a, w, 'SAF_module.bn', 1)), dim=-1)
            sim_vec = l2norm(torch.bmm(a, sim_emb).squeeze(1), dim=-1)
        else:
            raise ValueError('Invalid input of config.module_name in configs.py')
        cols.append(torch.sigmoid(_linear(sim_vec, w, 'sim_eval_w')).squeeze(1))
    return torch.stack(cols, 1)


def sgraf_model_train_grads(wi, wt, ws, images, ids, lengths, cfg):
    """One SGRAF.train_emb step up to the clipped gradients (Models.py:524-546): towers -> training-mode similarity -> hinge ->
    backward -> clip_grad_norm_.  wi: {'fc.weight', 'fc.bias'}; wt: EncoderText state_dict; ws: EncoderSimilarity state_dict
    (parameters and BatchNorm buffers).  Returns (loss, {name: clipped gradient}) with names 'img.<k>' / 'txt.<k>' / 'sim.<k>'.
    Parameter order as the reference builds it: txt_enc, img_enc, sim_enc (Models.py:498-500)."""
    is_param = lambda k: not (k.endswith('running_mean') or k.endswith('running_var') or k.endswith('num_batches_tracked'))
    names = [('txt.' + k, wt, k) for k in wt] + [('img.' + k, wi, k) for k in ('fc.weight', 'fc.bias')] + \
            [('sim.' + k, ws, k) for k in ws if is_param(k)]
    with torch.enable_grad():
        leaves = {n: d[k].detach().clone().requires_grad_(True) for n, d, k in names}
        wi_l = {k: leaves['img.' + k] for k in ('fc.weight', 'fc.bias')}
        wt_l = {k: leaves['txt.' + k] for k in wt}
        ws_l = {k: (leaves['sim.' + k] if is_param(k) else ws[k]) for k in ws}
        img = encoder_image_precomp(images, wi_l['fc.weight'], wi_l['fc.bias'], cfg.get('no_imgnorm', False))
        cap, cap_len = encoder_text(ids, lengths, wt_l, bool(cfg.get('bi_gru', False)), cfg.get('no_txtnorm', False), False, None)
        sims = sgraf_similarity_train(ws_l, img, cap, cap_len, cfg.get('module_name', 'SAF'), cfg.get('sgr_step', 3))
        loss = hinge_loss(sims, cfg.get('margin', 0.2), cfg.get('max_violation', False))
        loss.backward()
    grads = [leaves[n].grad if leaves[n].grad is not None else torch.zeros_like(leaves[n]) for n, _, _ in names]
    if cfg.get('grad_clip', 2.0) > 0:
        grads, _ = clip_grad_norm(grads, cfg.get('grad_clip', 2.0))
    return loss.detach(), {n: g for (n, _, _), g in zip(names, grads)}


# --------------------------------------------------------------------------------------
# a17  ranker (itr/metricmodule/evaluation.py:156-259)
# --------------------------------------------------------------------------------------


def _summary(ranks):
    n = len(ranks)
    r1 = 100.0 * int((ranks < 1).sum()) / n
    r5 = 100.0 * int((ranks < 5).sum()) / n
    r10 = 100.0 * int((ranks < 10).sum()) / n
    medr = np.floor(np.median(ranks)) + 1
    meanr = ranks.mean() + 1
    return (r1, r5, r10, medr, meanr)


def i2t_argsort(sims, return_ranks=False):
    """Literal restatement (argsort descending per image row, best of the 5 GT captions,
    evaluation.py:156-189).  Tie order follows numpy's argsort[::-1] like the reference."""
    sims = np.asarray(sims)
    npts = sims.shape[0]
    ranks = np.zeros(npts)
    top1 = np.zeros(npts)
    for i in range(npts):
        inds = np.argsort(sims[i])[::-1]
        pos = np.empty_like(inds)
        pos[inds] = np.arange(len(inds))
        ranks[i] = pos[5 * i:5 * i + 5].min()
        top1[i] = inds[0]
    out = _summary(ranks)
    return (out, (ranks, top1)) if return_ranks else out


def t2i_argsort(sims, return_ranks=False):
    """Literal restatement of evaluation.py:192-222 (caption j <-> image j // 5)."""
    sims = np.asarray(sims)
    npts = sims.shape[0]
    ranks = np.zeros(5 * npts)
    top1 = np.zeros(5 * npts)
    st = sims.T
    for j in range(5 * npts):
        inds = np.argsort(st[j])[::-1]
        ranks[j] = np.where(inds == j // 5)[0][0]
        top1[j] = inds[0]
    out = _summary(ranks)
    return (out, (ranks, top1)) if return_ranks else out


def rank_counts(sims, im_div=5):
    """Sort-free ranker used as the bit-exact checker of the HIP rank kernel.
    rank(query, gt) = #{k: S_k > S_gt} + #{k: S_k == S_gt and k > gt}   (with
    argsort(...)[::-1] the higher index wins a tie), i2t takes the min over the im_div GT
    captions; top1 = argmax with the highest index on ties.
    -> (i2t_rank[Ni], i2t_top1[Ni], t2i_rank[Nc], t2i_top1[Nc]) int64."""
    s = np.asarray(sims)
    ni, nc = s.shape
    i2t_rank = np.zeros(ni, np.int64)
    i2t_top1 = np.zeros(ni, np.int64)
    cidx = np.arange(nc)
    for i in range(ni):
        row = s[i]
        best = None
        for g in range(im_div * i, min(im_div * i + im_div, nc)):
            r = int((row > row[g]).sum() + ((row == row[g]) & (cidx > g)).sum())
            best = r if best is None else min(best, r)
        i2t_rank[i] = best
        i2t_top1[i] = nc - 1 - int(np.argmax(row[::-1]))
    t2i_rank = np.zeros(nc, np.int64)
    t2i_top1 = np.zeros(nc, np.int64)
    iidx = np.arange(ni)
    for j in range(nc):
        col = s[:, j]
        g = j // im_div
        t2i_rank[j] = int((col > col[g]).sum() + ((col == col[g]) & (iidx > g)).sum())
        t2i_top1[j] = ni - 1 - int(np.argmax(col[::-1]))
    return i2t_rank, i2t_top1, t2i_rank, t2i_top1


def recall_from_ranks(ranks):
    """(r1, r5, r10, medr, meanr) from 0-based ranks (evaluation.py:181-185)."""
    return _summary(np.asarray(ranks, dtype=np.float64))


def cal_recall(sims):
    """Same dict as evaluation.py:225-259 (without the prints)."""
    r, rt = i2t_argsort(sims, True)
    ri, rti = t2i_argsort(sims, True)
    ar = (r[0] + r[1] + r[2]) / 3
    ari = (ri[0] + ri[1] + ri[2]) / 3
    rsum = r[0] + r[1] + r[2] + ri[0] + ri[1] + ri[2]
    return {'rsum': rsum, 'i2t': r, 't2i': ri, 'i2t_ranks': rt[0], 'i2t_top1': rt[1],
            't2i_ranks': rti[0], 't2i_top1': rti[1],
            'result': [list(r) + list(ri) + [ar, ari, rsum]]}


# --------------------------------------------------------------------------------------
# a16  tiled similarity driver (evaluation.py:124-153), including quirk Q1
# --------------------------------------------------------------------------------------


def cal_sims(sim_fn, img_embs, cap_embs, lengths, shard_size, ref_quirk_unsliced_lengths=False):
    """sim_fn(img_block, cap_block, lens_block) -> (ni, nc).  The reference passes the FULL,
    un-sliced `lengths` to every caption shard (evaluation.py:149), so shard j > 0 is scored
    with the lengths of shard 0; set ref_quirk_unsliced_lengths=True to reproduce that."""
    n_img, n_cap = len(img_embs), len(cap_embs)
    d = np.zeros((n_img, n_cap))
    for i0 in range(0, n_img, shard_size):
        for j0 in range(0, n_cap, shard_size):
            j1 = min(j0 + shard_size, n_cap)
            lens = lengths if (ref_quirk_unsliced_lengths or lengths is None) else lengths[j0:j1]
            sim = sim_fn(torch.as_tensor(img_embs[i0:i0 + shard_size]),
                         torch.as_tensor(cap_embs[j0:j1]), lens)
            d[i0:i0 + shard_size, j0:j1] = sim.numpy()
    return d


# --------------------------------------------------------------------------------------
# a11  BERT (itr/modalmodule/bert.py:113-358) -- eval mode; weights: the BertModel state_dict
# --------------------------------------------------------------------------------------


def bert_layernorm(x, gamma, beta, eps=1e-12):
    """TF-style LayerNorm: epsilon inside the sqrt (bert.py:113-126)."""
    u = x.mean(-1, keepdim=True)
    s = (x - u).pow(2).mean(-1, keepdim=True)
    return gamma * ((x - u) / torch.sqrt(s + eps)) + beta


def bert_gelu(x):
    """erf GELU (bert.py:29-34)."""
    return x * 0.5 * (1.0 + torch.erf(x / math.sqrt(2.0)))


def bert_layer(w, p, x, mask01, heads):
    """One BERTLayer (bert.py:262-273) with parameter prefix p; mask01 (B, L) of 0/1 or None."""
    B, L, H = x.shape
    dk = H // heads

    def split(t):
        return t.view(B, L, heads, dk).permute(0, 2, 1, 3)
    q = split(_linear(x, w, p + 'attention.self.query'))
    k = split(_linear(x, w, p + 'attention.self.key'))
    v = split(_linear(x, w, p + 'attention.self.value'))
    scores = q @ k.transpose(-1, -2) / math.sqrt(dk)
    if mask01 is not None:
        scores = scores + ((1.0 - mask01.float()) * -10000.0)[:, None, None, :]
    ctx = (torch.softmax(scores, -1) @ v).permute(0, 2, 1, 3).reshape(B, L, H)
    att = bert_layernorm(_linear(ctx, w, p + 'attention.output.dense') + x, w[p + 'attention.output.LayerNorm.gamma'],
                         w[p + 'attention.output.LayerNorm.beta'])
    inter = bert_gelu(_linear(att, w, p + 'intermediate.dense'))
    return bert_layernorm(_linear(inter, w, p + 'output.dense') + att, w[p + 'output.LayerNorm.gamma'],
                          w[p + 'output.LayerNorm.beta'])


def bert_model(w, input_ids, token_type_ids, mask01, n_layers, heads, prefix=''):
    """BertModel.forward (bert.py:333-358) -> (all_encoder_layers, pooled_output)."""
    B, L = input_ids.shape
    if token_type_ids is None:
        token_type_ids = torch.zeros_like(input_ids)
    pos = torch.arange(L).unsqueeze(0).expand(B, L)
    e = (w[prefix + 'embeddings.word_embeddings.weight'][input_ids] + w[prefix + 'embeddings.position_embeddings.weight'][pos]
         + w[prefix + 'embeddings.token_type_embeddings.weight'][token_type_ids])
    x = bert_layernorm(e, w[prefix + 'embeddings.LayerNorm.gamma'], w[prefix + 'embeddings.LayerNorm.beta'])
    layers = []
    for n in range(n_layers):
        x = bert_layer(w, prefix + 'encoder.layer.%d.' % n, x, mask01, heads)
        layers.append(x)
    pooled = torch.tanh(_linear(x[:, 0], w, prefix + 'pooler.dense'))
    return layers, pooled


# a12  SAEM heads (TextEncoder.py:115-152, ImgEncoder.py:337-350)


def saem_text(w, txt_stru, input_ids, mask01, token_type_ids, n_layers, heads, trans_heads=None):
    """BertMapping.forward: w = the module's state_dict ('bert.*', 'convs1.k.*' | 'mapping_0.*' | 'layer.*', 'mapping.*')."""
    layers, _ = bert_model(w, input_ids, token_type_ids, mask01, n_layers, heads, prefix='bert.')
    last = layers[-1]
    if txt_stru == 'pooling':
        out = _linear(last, w, 'mapping_0').mean(1)
    elif txt_stru == 'cnn':
        B, L, H = last.shape
        feats = []
        for n in range(3):
            wk = w['convs1.%d.weight' % n]              # (512, 1, K, H)
            K = wk.shape[2]
            win = torch.stack([last[:, t:t + K].reshape(B, K * H) for t in range(L - K + 1)], 1)   # (B, L-K+1, K*H)
            y = torch.relu(win @ wk.reshape(wk.shape[0], K * H).t() + w['convs1.%d.bias' % n])
            feats.append(y.max(1)[0])
        out = torch.cat(feats, 1)
    elif txt_stru == 'trans':
        hid = _linear(last, w, 'mapping_0')
        out = bert_layer(w, 'layer.', hid, mask01, trans_heads).mean(1)
    else:
        raise ValueError("Unknown txt_stru: {}".format(txt_stru))
    return F.normalize(_linear(out, w, 'mapping'), p=2, dim=1)


def saem_image(w, x, heads):
    """TransformerMapping.forward: Linear -> BERTLayer (all-ones mask) -> mean over regions -> F.normalize."""
    h = bert_layer(w, 'layer.', _linear(x, w, 'mapping'), None, heads)
    return F.normalize(h.mean(1), p=2, dim=1)


# --------------------------------------------------------------------------------------
# a13  CAMERA towers (camera_.py:14-147; ImgEncoder.py:375-389; TextEncoder.py:181-192), eval mode
# --------------------------------------------------------------------------------------


def camera_gated_attention(w, p, inp, h):
    """GatedQueryAttLayer.forward (camera_.py:31-54), no mask."""
    B, L, D = inp.shape
    dk = D // h
    q, k, v = [_linear(inp, w, p + 'linears.%d' % n).view(B, L, h, dk).transpose(1, 2) for n in range(3)]
    G = _linear(q, w, p + 'fc_q') * _linear(k, w, p + 'fc_k')
    M = torch.sigmoid(_linear(G, w, p + 'fc_g'))
    q = q * M[..., :dk]
    k = k * M[..., dk:]
    att = torch.softmax(q @ k.transpose(-2, -1) / math.sqrt(dk), -1)
    return (att @ v).transpose(1, 2).reshape(B, L, D)


def camera_agsa(w, p, rgn_emb, pos_emb, h):
    """AGSA.forward with num_layers == 1 (camera_.py:70-89): rgn + bn(att(rgn * pos))."""
    B, L, D = rgn_emb.shape
    x = rgn_emb if pos_emb is None else rgn_emb * pos_emb
    x = camera_gated_attention(w, p + 'att_layers.0.', x, h)
    x = _bn_eval(x.reshape(B * L, D), w, p + 'bns.0', 1).view(B, L, D)
    return rgn_emb + x


def camera_position(w, p, boxes, imgs_wh):
    """absoluteEncode + PositionEncoder (camera_.py:118-147)."""
    x, y = boxes[:, :, 0], boxes[:, :, 1]
    bw, bh = boxes[:, :, 2] - x, boxes[:, :, 3] - y
    W, H = imgs_wh[:, 0:1], imgs_wh[:, 1:2]
    feat = torch.stack([x / W, y / H, bw / W, bh / H, bw / bh, (bw * bh) / (W * H)], dim=-1)
    return torch.sigmoid(_linear(feat, w, p + 'proj'))


def camera_summarization(w, p, rgn_emb):
    """Summarization.forward (camera_.py:108-114): 7 dilated Conv1d -> relu -> cat -> Linear."""
    ksz, dil, pad = [1, 3, 3, 3, 5, 5, 5], [1, 1, 2, 3, 1, 2, 3], [0, 1, 2, 3, 2, 4, 6]
    xt = rgn_emb.transpose(1, 2)
    ys = [torch.relu(F.conv1d(xt, w[p + 'convs_dilate.%d.weight' % i], w[p + 'convs_dilate.%d.bias' % i],
                              dilation=dil[i], padding=pad[i])) for i in range(7)]
    return _linear(torch.cat(ys, 1).transpose(1, 2), w, p + 'convs_fc')


def camera_image(w, images, boxes, imgs_wh, h):
    """EncoderImagePrecompSelfAttn.forward (ImgEncoder.py:375-389) -> (img_emb (B,k,D), smry_mat (B,R,k)).
    The two l2norm calls use the default dim=1 (the region axis), as in the reference."""
    fc = l2norm(_linear(images, w, 'fc'))
    pos = camera_position(w, 'position_enc.', boxes, imgs_wh)
    att = l2norm(camera_agsa(w, 'agsa.', fc, pos, h))
    smry = camera_summarization(w, 'mvs.', att)
    L = torch.softmax(smry, dim=1)
    return F.normalize(L.transpose(1, 2) @ att, dim=-1), smry


def camera_text(w, input_ids, mask01, token_type_ids, n_layers, bert_heads, h):
    """CAMERAEncoderText.forward (TextEncoder.py:181-192)."""
    layers, _ = bert_model(w, input_ids, token_type_ids, mask01, n_layers, bert_heads, prefix='bert.')
    x = _linear(layers[-1], w, 'mapping')
    B, L, D = x.shape
    agsa = camera_agsa(w, 'agsa.', x, None, h)
    y = _linear(torch.relu(_linear(agsa, w, 'fc1')), w, 'fc2')
    y = _bn_eval(y.reshape(B * L, D), w, 'bn', 1).view(B, L, D)
    return F.normalize((agsa + y).mean(1), p=2, dim=-1)


def diversity_regularization(smry_mat):
    """DiversityRegularization.forward (Objectives.py:532-542)."""
    s = F.normalize(smry_mat, dim=1)
    d = s.transpose(1, 2) @ s - torch.eye(s.shape[2]).unsqueeze(0)
    return (d ** 2).sum()


def angular_loss(im, s, angle_bound=1.0, max_violation=True):
    """AngularLoss.forward (Objectives.py:252-290): both directions, every other sample is a negative."""
    def one(anchors, positives, others):
        n = anchors.shape[0]
        idx = torch.tensor([[j for j in range(n) if j != i] for i in range(n)], dtype=torch.long)
        neg = others[idx]                                                   # (n, n-1, d)
        a, p = anchors.unsqueeze(1), positives.unsqueeze(1)
        x = 4.0 * angle_bound * ((a + p) @ neg.transpose(1, 2)) - 2.0 * (1.0 + angle_bound) * (a @ p.transpose(1, 2))
        if max_violation:
            return torch.log(1 + torch.exp(x.max(2)[0])).sum()
        t = x.max(2)[0]
        return (t + torch.log(torch.exp(-t) + torch.exp(x - t.unsqueeze(1)).sum(2))).mean()
    return one(im, s, s) + one(s, im, im)


# --------------------------------------------------------------------------------------
# a14  one optimisation step  model.train_emb  (itr/modalmodule/Models.py:115-145 VSE++, :198-225 SCAN):
#      forward_emb -> criterion -> backward -> clip_grad_norm_(params, grad_clip) -> Adam(lr).step()
# --------------------------------------------------------------------------------------


def clip_grad_norm(grads, max_norm):
    """torch.nn.utils.clip_grad_norm_ (norm type 2): total = ||(||g_1||, ..., ||g_n||)||;
    every gradient is multiplied by min(1, max_norm / (total + 1e-6)).  Returns (scaled grads, total)."""
    total = torch.norm(torch.stack([torch.norm(g, 2) for g in grads]), 2)
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    return [g * coef for g in grads], total


def adam_update(p, g, m, v, t, lr, beta1=0.9, beta2=0.999, eps=1e-8):
    """torch.optim.Adam, single tensor, no weight decay / amsgrad; t is the 1-based step.  Returns (p, m, v)."""
    m = beta1 * m + (1 - beta1) * g
    v = beta2 * v + (1 - beta2) * g * g
    bc1 = 1 - beta1 ** t
    bc2 = 1 - beta2 ** t
    denom = v.sqrt() / math.sqrt(bc2) + eps
    return p - (lr / bc1) * (m / denom), m, v


def gru_model_loss(kind, wi, wt, images, ids, lengths, cfg):
    """Training-mode forward of the GRU model family on one batch -> scalar loss.
    kind 'SCAN': region embeddings x word embeddings -> xattn_score -> hinge (Models.py:182-205);
    kind 'VSE++': mean-pooled regions (SURVEY Q3 build decision) x last GRU state -> cosine -> hinge."""
    bi = bool(cfg.get('bi_gru', False))
    if kind == 'SCAN':
        img = encoder_image_precomp(images, wi['fc.weight'], wi['fc.bias'], cfg.get('no_imgnorm', False))
        cap, cap_len = encoder_text(ids, lengths, wt, bi, cfg.get('no_txtnorm', False), False, None)
        scores = xattn_score(img, cap, cap_len, cfg.get('cross_attn', 't2i'), cfg.get('raw_feature_norm', 'clipped_l2norm'),
                             cfg.get('agg_func', 'LogSumExp'), cfg.get('lambda_lse', 6.0), cfg.get('lambda_softmax', 9.0))
    elif kind == 'VSE++':
        img = encoder_image_precomp(images.mean(1), wi['fc.weight'], wi['fc.bias'], cfg.get('no_imgnorm', False))
        cap, _ = encoder_text(ids, lengths, wt, bi, cfg.get('no_txtnorm', False), False, 'VSE++')
        scores = cosine_sim(img, cap)
    else:
        raise ValueError(kind)
    return hinge_loss(scores, cfg.get('margin', 0.2), cfg.get('max_violation', False))


def gru_model_train_step(kind, wi, wt, images, ids, lengths, cfg, state=None):
    """One train_emb step.  wi: {'fc.weight', 'fc.bias'}; wt: EncoderText state_dict.  state: None or
    {'t': int, 'm': {name: tensor}, 'v': {...}} with names 'txt.<k>' / 'img.<k>'.
    Returns (loss, clipped grads dict, new wi, new wt, new state).  Parameter order as the reference builds
    it: txt_enc.parameters() then img_enc.fc.parameters() (Models.py:95-97, :175-177)."""
    names = [('txt.' + k, wt, k) for k in wt] + [('img.' + k, wi, k) for k in ('fc.weight', 'fc.bias')]
    with torch.enable_grad():
        leaves = {n: d[k].detach().clone().requires_grad_(True) for n, d, k in names}
        wi_l = {k: leaves['img.' + k] for k in ('fc.weight', 'fc.bias')}
        wt_l = {k: leaves['txt.' + k] for k in wt}
        loss = gru_model_loss(kind, wi_l, wt_l, images, ids, lengths, cfg)
        loss.backward()
    grads = [leaves[n].grad if leaves[n].grad is not None else torch.zeros_like(leaves[n]) for n, _, _ in names]
    if cfg.get('grad_clip', 2.0) > 0:
        grads, _ = clip_grad_norm(grads, cfg.get('grad_clip', 2.0))
    if state is None:
        state = {'t': 0, 'm': {n: torch.zeros_like(leaves[n]) for n, _, _ in names}, 'v': {n: torch.zeros_like(leaves[n]) for n, _, _ in names}}
    t = state['t'] + 1
    new = {'t': t, 'm': {}, 'v': {}}
    out = {}
    for (n, _, _), g in zip(names, grads):
        p, m, v = adam_update(leaves[n].detach(), g, state['m'][n], state['v'][n], t, cfg['learning_rate'])
        out[n], new['m'][n], new['v'][n] = p, m, v
    new_wi = {k: out['img.' + k] for k in ('fc.weight', 'fc.bias')}
    new_wt = {k: out['txt.' + k] for k in wt}
    return loss.detach(), {n: g for (n, _, _), g in zip(names, grads)}, new_wi, new_wt, new
