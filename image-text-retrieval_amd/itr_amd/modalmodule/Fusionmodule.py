"""Similarity / fusion modules with the reference's names (itr/modalmodule/Fusionmodule.py).  The torch modules
below are PARAMETER CONTAINERS with the reference's structure (so state_dict keys and initialisation match and
reference checkpoints load); `forward` hands the parameters to the HIP path."""
import numpy as np
import torch
from torch import nn

from .. import ops


def _xavier(m):
    if isinstance(m, nn.Linear):
        r = np.sqrt(6.) / np.sqrt(m.in_features + m.out_features)
        m.weight.data.uniform_(-r, r)
        m.bias.data.fill_(0)
    elif isinstance(m, nn.BatchNorm1d):
        m.weight.data.fill_(1)
        m.bias.data.zero_()


class VisualSA(nn.Module):
    """Fusionmodule.py:454-507 (parameters only)."""

    def __init__(self, embed_dim, dropout_rate, num_region):
        super().__init__()
        self.embedding_local = nn.Sequential(nn.Linear(embed_dim, embed_dim), nn.BatchNorm1d(num_region), nn.Tanh(),
                                             nn.Dropout(dropout_rate))
        self.embedding_global = nn.Sequential(nn.Linear(embed_dim, embed_dim), nn.BatchNorm1d(embed_dim), nn.Tanh(),
                                              nn.Dropout(dropout_rate))
        self.embedding_common = nn.Sequential(nn.Linear(embed_dim, 1))
        for seq in self.children():
            for m in seq:
                _xavier(m)


class TextSA(nn.Module):
    """Fusionmodule.py:510-559 (parameters only)."""

    def __init__(self, embed_dim, dropout_rate):
        super().__init__()
        self.embedding_local = nn.Sequential(nn.Linear(embed_dim, embed_dim), nn.Tanh(), nn.Dropout(dropout_rate))
        self.embedding_global = nn.Sequential(nn.Linear(embed_dim, embed_dim), nn.Tanh(), nn.Dropout(dropout_rate))
        self.embedding_common = nn.Sequential(nn.Linear(embed_dim, 1))
        for seq in self.children():
            for m in seq:
                _xavier(m)


class GraphReasoning(nn.Module):
    """Fusionmodule.py:562-597 (parameters only)."""

    def __init__(self, sim_dim):
        super().__init__()
        self.graph_query_w = nn.Linear(sim_dim, sim_dim)
        self.graph_key_w = nn.Linear(sim_dim, sim_dim)
        self.sim_graph_w = nn.Linear(sim_dim, sim_dim)
        for m in self.children():
            _xavier(m)


class AttentionFiltration(nn.Module):
    """Fusionmodule.py:600-629 (parameters only)."""

    def __init__(self, sim_dim):
        super().__init__()
        self.attn_sim_w = nn.Linear(sim_dim, 1)
        self.bn = nn.BatchNorm1d(1)
        for m in self.children():
            _xavier(m)


class EncoderSimilarity(nn.Module):
    """Image-text similarity by SGR / SAF (Fusionmodule.py:373-451).  img_emb (n_img, 36, D), cap_emb
    (n_cap, L, D) padded, cap_lens -> (n_img, n_cap).  This forward is the evaluation mode (BatchNorm running statistics, no
    dropout) on the fused kernels; the training mode is encoder_similarity_train below."""

    def __init__(self, embed_size, sim_dim, module_name='AVE', sgr_step=3):
        super().__init__()
        self.module_name = module_name
        self.sgr_step = sgr_step
        self.v_global_w = VisualSA(embed_size, 0.4, 36)
        self.t_global_w = TextSA(embed_size, 0.4)
        self.sim_tranloc_w = nn.Linear(embed_size, sim_dim)
        self.sim_tranglo_w = nn.Linear(embed_size, sim_dim)
        self.sim_eval_w = nn.Linear(sim_dim, 1)
        if module_name == 'SGR':
            self.SGR_module = nn.Sequential()
            for i in range(sgr_step):
                self.SGR_module.add_module(f'sgr{i}', GraphReasoning(sim_dim))
        elif module_name == 'SAF':
            self.SAF_module = AttentionFiltration(sim_dim)
        else:
            raise ValueError('Invalid input of config.module_name in configs.py')
        for m in self.children():
            _xavier(m)

    def forward(self, img_emb, cap_emb, cap_lens, *args, **kwargs):
        """(B, 36, D), padded (Nc, L, D), lengths -> (B, Nc) (Fusionmodule.py:406-451).  Evaluation: the fused kernels.  Training
        mode (batch statistics, live dropout) or a differentiable call: the reference's per-caption structure on the autograd
        tape (encoder_similarity_train), fed with the captions re-packed from the padded tensor."""
        if self.training:
            from .. import autograd as ag
            Nc, L, D = cap_emb.shape
            lens = [int(x) for x in cap_lens][:Nc]
            flat = np.concatenate([b * L + np.arange(l, dtype=np.int64) for b, l in enumerate(lens)])
            words = cap_emb.reshape(Nc * L, D).index_select(0, ops.h2d(flat, cap_emb.device))
            off = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.int64)
            if not hasattr(self, '_seeds'):
                self._seeds = ag.DropoutSeeds()
            self._seeds.new_step()
            return encoder_similarity_train(self, img_emb, words, off, lens, self._seeds, self.training)
        return ops.sgraf_padded(img_emb, cap_emb, cap_lens, self.state_dict(), self.module_name, self.sgr_step)

    def forward_packed(self, img_emb, words, plan):
        return ops.sgraf_scores(img_emb, words, plan, self.state_dict(), self.module_name, self.sgr_step)


# ---------------------------------------------------------------------------------------------------------------------
# Training-mode forward of EncoderSimilarity on the autograd tape (SGRAF.train_emb).  Training mode is per-caption by construction in
# the reference (a loop over the captions of the batch, Fusionmodule.py:415-447: TextSA and AttentionFiltration see one caption at a
# time, and AttentionFiltration's BatchNorm1d(1) takes its batch statistics once per caption).  encoder_similarity_train_batched keeps
# those semantics on ragged whole-batch kernels; encoder_similarity_train_grouped is the loop itself, shared between equal lengths.
def _seq_linear(x2d, seq, idx=0):
    from .. import autograd as ag
    return ag.linear(x2d, seq[idx].weight, seq[idx].bias)


def _sa_train(mod, local, raw_global, seeds, training):
    """VisualSA / TextSA.forward (Fusionmodule.py:497-512 / :549-564) on local [B, n, D], raw_global [B, D]."""
    from .. import autograd as ag
    B, n, D = local.shape
    le = _seq_linear(local.reshape(B * n, D), mod.embedding_local)
    ge = _seq_linear(raw_global, mod.embedding_global)
    if isinstance(mod.embedding_local[1], nn.BatchNorm1d):          # VisualSA: BatchNorm1d(num_region) on (B, 36, D): channel = region
        bn_l, bn_g = mod.embedding_local[1], mod.embedding_global[1]
        if training:
            t = le.view(B, n, D).permute(0, 2, 1).reshape(B * D, n)        # rows = (image, feature), columns = regions
            le = ag.batch_norm_train(t, bn_l).view(B, D, n).permute(0, 2, 1).reshape(B * n, D)
            ge = ag.batch_norm_train(ge, bn_g)
        else:        # evaluation mode (the long-caption fallback of ops.sgraf_scores): running statistics as a folded affine
            def fold(bn):
                sc = (bn.weight / torch.sqrt(bn.running_var + bn.eps)).detach()
                return sc.contiguous(), (bn.bias - bn.running_mean * sc).detach().contiguous()
            sc, sh = fold(bn_l)
            t = le.view(B, n, D).permute(0, 2, 1).reshape(B * D, n).contiguous()
            le = ops.affine_cols(t, sc, sh).view(B, D, n).permute(0, 2, 1).reshape(B * n, D).contiguous()
            sc, sh = fold(bn_g)
            ge = ops.affine_cols(ge, sc, sh)
    p = float(mod.embedding_local[-1].p)
    le = ag.dropout(ag.act(le, 'tanh'), p, seeds, training).view(B, n, D)
    ge = ag.dropout(ag.act(ge, 'tanh'), p, seeds, training)
    common = ag.mul(le, ge.unsqueeze(1).expand(B, n, D).contiguous())
    w = _seq_linear(common.reshape(B * n, D), mod.embedding_common).view(B, n, 1)      # logits over the n rows
    new_global = ag.summarize(w, local).view(B, D)                                        # softmax over dim=1, weighted sum
    return ag.l2norm_rows(new_global, eps=1e-8)


def _scan_attention_train(caps, img_emb, smooth=9.0):
    """SCAN_attention(cap_i_expand, img_emb, smooth) (Fusionmodule.py:632-664) for a GROUP of G captions of the same length:
    caps [G, W, D], img_emb [B, 36, D] -> [G*B, W, D] (caption-major)."""
    from .. import autograd as ag
    B, R, D = img_emb.shape
    G, W, _ = caps.shape
    attn = ag.cosine_scores(img_emb.reshape(B * R, D), caps.reshape(G * W, D))       # (B*36, G*W): bmm(context, query^T), one GEMM
    attn = ag.l2norm_rows(ag.act(attn, 'leaky_relu').view(B * R * G, W), eps=1e-8)      # LeakyReLU(0.1), l2norm over the words of a caption
    smry = (attn * smooth).view(B, R, G, W).permute(2, 0, 1, 3).reshape(G * B, R, W)   # caption-major copy
    x = img_emb.unsqueeze(0).expand(G, B, R, D).reshape(G * B, R, D)
    return ag.l2norm_rows(ag.summarize(smry, x), eps=1e-8)                              # softmax over the regions, weighted sum


def _graph_step_train(gr, x):
    """GraphReasoning.forward (Fusionmodule.py:579-586) on x [B, n, S]."""
    from .. import autograd as ag
    B, n, S = x.shape
    x2 = x.reshape(B * n, S)
    q = ag.linear(x2, gr.graph_query_w.weight, gr.graph_query_w.bias).view(B, n, S)
    k = ag.linear(x2, gr.graph_key_w.weight, gr.graph_key_w.bias).view(B, n, S)
    scores_t = ag.bmm_nt(k, q)                                                   # [b, r, v] = k_r . q_v  (edge logits, transposed)
    sgr = ag.summarize(scores_t, x)                                              # softmax over r for every query v, times x
    return ag.act(ag.linear(sgr.reshape(B * n, S), gr.sim_graph_w.weight, gr.sim_graph_w.bias), 'relu').view(B, n, S)


def _saf_train(saf, x, G, training):
    """AttentionFiltration.forward (Fusionmodule.py:613-618) on x [G*B, n, S] (caption-major).  The reference calls it once per
    caption, so its BatchNorm1d(1) normalises with the statistics of ONE caption's B * n logits and updates its running statistics
    once per caption, in caption order: the G captions become G channels of one column BatchNorm with the shared gamma / beta."""
    from .. import autograd as ag
    GB, n, S = x.shape
    B = GB // G
    a = ag.linear(x.reshape(GB * n, S), saf.attn_sim_w.weight, saf.attn_sim_w.bias)       # (G*B*n, 1)
    if training:
        bn = saf.bn
        stats = []
        t = a.view(G, B * n).t().contiguous()                                             # rows = (image, node), columns = captions
        y = ag._BatchNormTrain.apply(t, bn.weight.expand(G).contiguous(), bn.bias.expand(G).contiguous(), bn.eps, stats)
        a = y.t().contiguous().view(GB * n, 1)
        mean, invstd = stats[0]
        with torch.no_grad():
            N = B * n
            var_u = (1.0 / (invstd * invstd) - bn.eps) * (N / max(N - 1, 1))
            mom = bn.momentum if bn.momentum is not None else 0.1
            for gi in range(G):                                                           # sequential momentum updates, caption order
                bn.running_mean.mul_(1 - mom).add_(mean[gi:gi + 1], alpha=mom)
                bn.running_var.mul_(1 - mom).add_(var_u[gi:gi + 1], alpha=mom)
            bn.num_batches_tracked += G
    else:
        bn = saf.bn
        sc = (bn.weight / torch.sqrt(bn.running_var + bn.eps)).detach()
        a = ops.affine_cols(a, sc.contiguous(), (bn.bias - bn.running_mean * sc).detach().contiguous())
    a = ag.act(a, 'sigmoid').view(GB, n)
    a = a / (a.abs().sum(1, keepdim=True) + 1e-8)                                # l1norm over the nodes
    sim_saf = (a.unsqueeze(2) * x).sum(1)
    return ag.l2norm_rows(sim_saf, eps=1e-8)


def encoder_similarity_train(sim_enc, img_emb, words, tok_off, lens, seeds, training=True, max_group=32, seeds_global=None, batched=None):
    """EncoderSimilarity.forward in training mode: img_emb [B, 36, D], packed word embeddings words [n_tok, D] with caption c at
    rows tok_off[c] .. tok_off[c] + lens[c] -> sims [B, n_caption].  All pairs at once on the ragged kernels of csrc/sgraf_train.hip
    (encoder_similarity_train_batched); shapes those do not take (captions of more than 96 words, D > 2048 ...) and `batched=False`
    (the cross-check of the tests) run the grouped restatement of the reference's per-caption loop below."""
    from .. import sgraf_train as sgt
    lens = [int(x) for x in lens]
    packed = bool(np.array_equal(np.asarray(tok_off, dtype=np.int64)[:len(lens)], np.concatenate([[0], np.cumsum(lens)[:-1]]))) and \
        words.shape[0] == int(np.sum(lens))
    ok = packed and training and sgt.supported(img_emb.shape[2], img_emb.shape[1], sim_enc.sim_eval_w.in_features, lens)
    if batched is None:
        batched = ok
    if batched:
        if not ok:
            raise NotImplementedError("encoder_similarity_train: this batch does not fit the batched kernels (csrc/sgraf_train.hip)")
        return encoder_similarity_train_batched(sim_enc, img_emb, words, lens, seeds, seeds_global)
    return encoder_similarity_train_grouped(sim_enc, img_emb, words, tok_off, lens, seeds, training, max_group, seeds_global)


def encoder_similarity_train_batched(sim_enc, img_emb, words, lens, seeds, seeds_global=None):
    """Training-mode EncoderSimilarity.forward (Fusionmodule.py:406-451) for ALL B x C pairs in one pass: image-major ragged matrices
    (itr_amd/sgraf_train.py), one launch per stage, the dense layers as whole-batch GEMMs.  Same arithmetic per pair as the reference's
    per-caption loop; AttentionFiltration's BatchNorm1d(1) keeps its per-caption batch statistics and running-statistic updates."""
    from .. import autograd as ag
    from .. import sgraf_train as sgt
    B, R, D = img_emb.shape
    lay = sgt.Layout(lens, img_emb.device)
    C = lay.C
    img_glo = _sa_train(sim_enc.v_global_w, img_emb, ag.mean_mid(img_emb), seeds_global or seeds, True)            # (B, D)
    # TextSA on the packed captions (Fusionmodule.py:549-564)
    t = sim_enc.t_global_w
    p = float(t.embedding_local[-1].p)
    le = ag.dropout(ag.act(_seq_linear(words, t.embedding_local), 'tanh'), p, seeds, True)                          # (T, D)
    ge = ag.dropout(ag.act(_seq_linear(sgt.seg_mean(words, lay), t.embedding_global), 'tanh'), p, seeds, True)      # (C, D)
    logit = _seq_linear(ag.mul(le, sgt.seg_spread(ge, lay)), t.embedding_common).reshape(-1)                        # (T,)
    cap_glo = ag.l2norm_rows(sgt.seg_smry(logit, words, lay), eps=1e-8)                                              # (C, D)
    # local alignments: attention of every word over the regions of every image, context, squared difference (:421-427)
    A = ag.cosine_scores(img_emb.reshape(B * R, D), words)                                                           # (B R, T)
    X = sgt.loc_ctx(sgt.loc_attn(A, lay, B, R, 9.0), img_emb, words)                                                 # (B T, D)
    sim_loc = ag.l2norm_rows(ag.linear(X, sim_enc.sim_tranloc_w.weight, sim_enc.sim_tranloc_w.bias), eps=1e-8)
    sim_glo = ag.l2norm_rows(ag.linear(sgt.pair_sqdiff(img_glo, cap_glo), sim_enc.sim_tranglo_w.weight, sim_enc.sim_tranglo_w.bias), eps=1e-8)
    nodes = sgt.assemble_nodes(sim_glo, sim_loc, lay, B)                                                             # (B (T + C), S)
    if sim_enc.module_name == 'SGR':
        steps = list(sim_enc.SGR_module)
        for gr in steps[:-1]:
            q = ag.linear(nodes, gr.graph_query_w.weight, gr.graph_query_w.bias)
            k = ag.linear(nodes, gr.graph_key_w.weight, gr.graph_key_w.bias)
            nodes = ag.act(ag.linear(sgt.graph_attn(q, k, nodes, lay, B), gr.sim_graph_w.weight, gr.sim_graph_w.bias), 'relu')
        # the last step is read at node 0 only (sim_emb[:, 0, :], Fusionmodule.py:437-438): one query row per pair, keys from all nodes
        gr = steps[-1]
        q0 = ag.linear(ag.gather_rows(nodes, lay.node0_rows(B)), gr.graph_query_w.weight, gr.graph_query_w.bias)      # (B C, S)
        k = ag.linear(nodes, gr.graph_key_w.weight, gr.graph_key_w.bias)
        sim_vec = ag.act(ag.linear(sgt.graph_attn(q0, k, nodes, lay, B, row0=True), gr.sim_graph_w.weight, gr.sim_graph_w.bias), 'relu')
    else:
        saf = sim_enc.SAF_module
        a = ag.linear(nodes, saf.attn_sim_w.weight, saf.attn_sim_w.bias).reshape(-1)
        sim_vec = ag.l2norm_rows(sgt.saf_pool(sgt.seg_bn_train(a, saf.bn, lay, B), nodes, lay, B), eps=1e-8)
    sims = ag.act(ag.linear(sim_vec, sim_enc.sim_eval_w.weight, sim_enc.sim_eval_w.bias), 'sigmoid')               # (B C, 1)
    return sims.view(B, C)


def encoder_similarity_train_grouped(sim_enc, img_emb, words, tok_off, lens, seeds, training=True, max_group=32, seeds_global=None):
    """The reference's per-caption loop on the tape: consecutive captions of the same length (collate_fn sorts by length) are
    processed as one group -- the arithmetic per caption is the reference's, the launches are shared.  Rounds 1-5 trained through
    this (85 / 224 ms per 128 x 128 step); it stays as the path for shapes the batched kernels do not take and as their cross-check."""
    from .. import autograd as ag
    B, R, D = img_emb.shape
    # (data parallel: every rank computes the global image vectors of the whole batch, with the same dropout masks -- seeds_global)
    img_glo = _sa_train(sim_enc.v_global_w, img_emb, ag.mean_mid(img_emb), seeds_global or seeds, training)
    cols = []
    c = 0
    while c < len(lens):
        n_word = lens[c]
        G = 1
        while c + G < len(lens) and lens[c + G] == n_word and G < max_group and int(tok_off[c + G]) == int(tok_off[c]) + G * n_word:
            G += 1
        o = int(tok_off[c])
        caps = words[o:o + G * n_word].view(G, n_word, D)                        # (G, W, D)
        cap_glo = _sa_train(sim_enc.t_global_w, caps, caps.mean(1), seeds, training)          # (G, D); TextSA has no BatchNorm
        ctx = _scan_attention_train(caps, img_emb)                               # (G*B, W, D)
        cap_exp = caps.unsqueeze(1).expand(G, B, n_word, D).reshape(G * B, n_word, D)
        sim_loc = (ctx - cap_exp) ** 2
        sim_loc = ag.l2norm_rows(ag.linear(sim_loc.reshape(G * B * n_word, D), sim_enc.sim_tranloc_w.weight, sim_enc.sim_tranloc_w.bias), eps=1e-8)
        glo_diff = (img_glo.unsqueeze(0) - cap_glo.unsqueeze(1)).reshape(G * B, D) ** 2
        sim_glo = ag.l2norm_rows(ag.linear(glo_diff, sim_enc.sim_tranglo_w.weight, sim_enc.sim_tranglo_w.bias), eps=1e-8)
        sim_emb = torch.cat([sim_glo.unsqueeze(1), sim_loc.view(G * B, n_word, -1)], 1)        # (G*B, W + 1, S)
        if sim_enc.module_name == 'SGR':
            for gr in sim_enc.SGR_module:
                sim_emb = _graph_step_train(gr, sim_emb)
            sim_vec = sim_emb[:, 0, :]
        else:
            sim_vec = _saf_train(sim_enc.SAF_module, sim_emb, G, training)
        sims = ag.act(ag.linear(sim_vec, sim_enc.sim_eval_w.weight, sim_enc.sim_eval_w.bias), 'sigmoid')     # (G*B, 1)
        cols.append(sims.view(G, B).t())
        c += G
    return torch.cat(cols, 1)


# ---------------------------------------------------------------------------------------------------------------------
# VSRN captioning branch (Fusionmodule.py:10-367), training only: parameter containers with the reference's names + the
# teacher-forced forward on the autograd tape.  (The reference never stores these modules in its checkpoints, Models.py:37-45.)
class Attention(nn.Module):
    """Fusionmodule.py:115-146."""

    def __init__(self, dim):
        super().__init__()
        self.dim = dim
        self.linear1 = nn.Linear(dim * 2, dim)
        self.linear2 = nn.Linear(dim, 1, bias=False)


class EncoderRNN(nn.Module):
    """Fusionmodule.py:149-206 (GRU cell, one layer, unidirectional)."""

    def __init__(self, dim_vid, dim_hidden, input_dropout_p=0.2, rnn_dropout_p=0.5, n_layers=1, bidirectional=False, rnn_cell='gru'):
        super().__init__()
        if str(rnn_cell).lower() != 'gru' or n_layers != 1 or bidirectional:
            raise NotImplementedError("VSRN captioning encoder: one unidirectional GRU layer (the reference's configuration)")
        self.dim_vid, self.dim_hidden, self.input_dropout_p = dim_vid, dim_hidden, float(input_dropout_p)
        self.vid2hid = nn.Linear(dim_vid, dim_hidden)
        nn.init.xavier_normal_(self.vid2hid.weight)
        self.rnn = nn.GRU(dim_hidden, dim_hidden, 1, batch_first=True)      # (nn.GRU's own dropout acts between layers only)


class DecoderRNN(nn.Module):
    """Fusionmodule.py:209-365 (GRU cell, one layer)."""

    def __init__(self, vocab_size, max_len, dim_hidden, dim_word, n_layers=1, rnn_cell='gru', bidirectional=False, input_dropout_p=0.1,
                 rnn_dropout_p=0.1):
        super().__init__()
        if str(rnn_cell).lower() != 'gru' or n_layers != 1 or bidirectional:
            raise NotImplementedError("VSRN captioning decoder: one unidirectional GRU layer (the reference's configuration)")
        self.dim_output, self.dim_hidden, self.dim_word, self.max_length = vocab_size, dim_hidden, dim_word, max_len
        self.input_dropout_p = float(input_dropout_p)
        self.embedding = nn.Embedding(vocab_size, dim_word)
        self.attention = Attention(dim_hidden)
        self.rnn = nn.GRU(dim_hidden + dim_word, dim_hidden, 1, batch_first=True)
        self.out = nn.Linear(dim_hidden, vocab_size)
        nn.init.xavier_normal_(self.out.weight)


class S2VTAttModel(nn.Module):
    """Fusionmodule.py:10-35."""

    def __init__(self, encoder, decoder):
        super().__init__()
        self.encoder = encoder
        self.decoder = decoder

    def caption_loss_train(self, vid_feats, labels, masks, seeds, training=True, batch_total=None):
        """LanguageModelCriterion(caption_model(vid_feats, labels, 'train'), labels[:, 1:], masks[:, 1:])  (Models.py:303-313,
        Objectives.py:138-158) on the autograd tape: encoder GRU over the regions, then max_len - 1 teacher-forced decoder steps of
        attention -> GRU cell -> vocabulary projection -> log-softmax / masked NLL; sum over steps and rows / batch size."""
        from .. import autograd as ag
        enc, dec = self.encoder, self.decoder
        B, N, Dv = vid_feats.shape
        H = enc.dim_hidden
        x = ag.dropout(ag.linear(vid_feats.reshape(B * N, Dv), enc.vid2hid.weight, enc.vid2hid.bias), enc.input_dropout_p, seeds, training)
        dev = x.device
        tokens = torch.arange(B * N, device=dev, dtype=torch.int64)
        off = torch.arange(B, device=dev, dtype=torch.int64) * N
        enc_out = ag.gru_sequence(tokens, off, [N] * B, x, dict(enc.rnn.named_parameters()), False)          # (B*N, H)
        h = ag.gather_rows(enc_out, off + (N - 1))                                                             # encoder_hidden
        enc_out3 = enc_out.view(B, N, H)
        labels = labels.to(dev)
        masks = masks.to(dev).to(torch.float32)
        steps = min(dec.max_length - 1, labels.shape[1] - 1)
        att = dec.attention
        # Attention.forward (Fusionmodule.py:136-140): linear1(cat(encoder_outputs, hidden)) = enc_out W_e^T + b + h W_h^T.  The encoder
        # half does not depend on the decoder step: it is ONE GEMM per caption batch (4 608 x 2 048 x 2 048) instead of one per step
        # (60 x 77 GFLOP forward were 2/3 of the step's flop, and as much again twice in the backward); its gradient accumulates over
        # the steps on the tape and meets W_e / enc_out once.  Same sum up to the order of the two partial products.
        W1e, W1h = att.linear1.weight[:, :H].contiguous(), att.linear1.weight[:, H:].contiguous()
        enc_part = ag.linear(enc_out, W1e, att.linear1.bias).view(B, N, H)
        hs = []
        for i in range(steps):
            cur = ag.gather_rows(dec.embedding.weight, labels[:, i].contiguous())
            e = ag.addattn_score(enc_part, ag.linear(h, W1h, None), att.linear2.weight)      # linear2(tanh(.)) in one pass, [B, N]
            context = ag.summarize(e.view(B, N, 1), enc_out3).view(B, H)                                      # softmax over the regions
            dec_in = ag.dropout(torch.cat([cur, context], 1), dec.input_dropout_p, seeds, training)
            h = ag.gru_cell(dec_in, h, dec.rnn)
            hs.append(h)
        # teacher forcing: the logits never feed back into the recurrence, so the vocabulary projection and the log-softmax / NLL of
        # ALL steps are one GEMM and one kernel (rows step-major) instead of max_len - 1 small ones -- the same sum
        logits = ag.linear(torch.cat(hs, 0), dec.out.weight, dec.out.bias)
        tgt = labels[:, 1:steps + 1].t().reshape(-1)
        msk = masks[:, 1:steps + 1].t().reshape(-1)
        # (data parallel: these rows are a shard, the divisor stays the size of the whole batch)
        return ag.nll_logsoftmax(logits, tgt, msk).sum() / (B if batch_total is None else batch_total)


class MultiViewMatching(nn.Module):
    """CAMERA: max over the k views (Fusionmodule.py:670-692)."""

    def forward(self, imgs, caps, *args, **kwargs):
        return ops.mvm_scores(imgs, caps)
