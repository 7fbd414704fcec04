"""Text towers with the reference's signatures (itr/modalmodule/TextEncoder.py)."""
import numpy as np
import torch
from torch import nn

from .. import ops
from . import bert
from .bert import BertConfig, BertModel


def pack_tokens(x, lengths):
    """(B, L) padded ids + descending lengths -> packed ids (n_tok,), tok_off (B,), host lengths, (B, L) validity mask.
    The gather indices are built on the host from `lengths` (known there), so nothing waits for the GPU: a boolean-mask
    gather would have to count its hits on the device and stall the launch queue."""
    lens = [int(l) for l in lengths]
    B, L = x.shape
    flat = np.concatenate([b * L + np.arange(l, dtype=np.int64) for b, l in enumerate(lens)]) if B else np.zeros(0, np.int64)
    idx = ops.h2d(flat, x.device)
    toks = x.reshape(-1).index_select(0, idx)
    off = ops.h2d(np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.int64), x.device)
    ar = torch.arange(L, device=x.device).unsqueeze(0)
    mask = ar < ops.h2d(np.asarray(lens, np.int64), x.device).unsqueeze(1)
    return toks.contiguous(), off, lens, mask


class EncoderText(nn.Module):
    """Embedding -> (bi)GRU -> direction average -> [last valid step] -> [l2norm] -> [abs]
    (TextEncoder.py:15-70).  `forward` returns the padded (B, L, D) tensor like the reference;
    `forward_packed` returns the packed (n_tok, D) layout the SCAN / SGRAF kernels consume."""

    def __init__(self, vocab_size, word_dim, embed_size, num_layers, use_bi_gru=False, no_txtnorm=False,
                 dropout=0., use_abs=False, method_name=None):
        super().__init__()
        if num_layers < 1:
            raise ValueError("num_layers must be >= 1")
        self.num_layers = num_layers
        self.embed_size = embed_size
        self.no_txtnorm = no_txtnorm
        self.use_abs = use_abs
        self.method_name = method_name
        self.embed = nn.Embedding(vocab_size, word_dim)
        self.dropout_p = dropout          # live in training mode (forward_packed_train), identity in evaluation
        self.use_bi_gru = use_bi_gru
        self.rnn = nn.GRU(word_dim, embed_size, num_layers, batch_first=True, bidirectional=use_bi_gru)
        self.init_weights()

    def init_weights(self):
        self.embed.weight.data.uniform_(-0.1, 0.1)

    def _weights(self):
        w = {'embed.weight': self.embed.weight.detach()}
        w.update({'rnn.' + k: v.detach() for k, v in self.rnn.named_parameters()})
        return w

    # ---- nn.GRU(num_layers > 1) (TextEncoder.py:31; no reference config uses it, the constructor argument exists): layer l > 0
    # reads the previous layer's output sequence -- for a bi-GRU the CONCATENATION [forward | backward] (2 D wide) -- so the lower
    # layers run the two directions as separate uni-directional passes of the same kernels over a dense "embedding table" (the
    # backward direction on the caption's rows in reverse order) and only the top layer uses the fused (fwd + bwd) / 2 form.
    @staticmethod
    def _reversed_rows(lens, device):
        """index of the row that is the mirror image of row r inside its caption (packed layout)."""
        idx = np.concatenate([o + np.arange(l - 1, -1, -1, dtype=np.int64) for o, l in zip(np.concatenate([[0], np.cumsum(lens)[:-1]]), lens)]) \
            if len(lens) else np.zeros(0, np.int64)
        return ops.h2d(idx, device)

    def _layer_params(self, layer, reverse, detach):
        sfx = '_l%d%s' % (layer, '_reverse' if reverse else '')
        get = (lambda n: getattr(self.rnn, n + sfx).detach()) if detach else (lambda n: getattr(self.rnn, n + sfx))
        return {'weight_ih_l0': get('weight_ih'), 'weight_hh_l0': get('weight_hh'), 'bias_ih_l0': get('bias_ih'), 'bias_hh_l0': get('bias_hh')}

    def _lower_layers(self, toks, off, lens, table, tape):
        """Layers 0 .. num_layers-2 -> (row indices, dense input table) for the top layer."""
        from .. import autograd as ag
        dev = toks.device
        for layer in range(self.num_layers - 1):
            ar = torch.arange(toks.numel(), device=dev, dtype=torch.int64)

            def run(tokens, reverse):
                prm = self._layer_params(layer, reverse, detach=not tape)
                if tape:
                    return ag.gru_sequence(tokens, off, lens, table, prm, False)
                w = {'embed.weight': table.detach()}
                w.update({'rnn.' + k: v for k, v in prm.items()})
                return ops.gru_encode(tokens, off, lens, w, False, no_txtnorm=True)
            fwd = run(toks, False)
            if self.use_bi_gru:
                rev = self._reversed_rows(lens, dev)
                bwd = run(toks.index_select(0, rev), True)            # the caption's tokens last-to-first
                bwd = ag.gather_rows(bwd, rev) if tape else bwd.index_select(0, rev)      # back to reading order
                table = torch.cat([fwd, bwd], 1)
            else:
                table = fwd
            toks = ar
        return toks, table

    def _top_weights(self, table, detach):
        top = self.num_layers - 1
        w = {'embed.weight': table.detach() if detach else table}
        for rev in ((False, True) if self.use_bi_gru else (False,)):
            for k, v in self._layer_params(top, rev, detach).items():
                w['rnn.' + k + ('_reverse' if rev else '')] = v
        return w

    def forward_packed(self, x, lengths):
        toks, off, lens, mask = pack_tokens(x, lengths)
        last = self.method_name in ('VSE++', 'VSRN')
        if self.num_layers == 1:
            w = self._weights()
        else:
            toks, table = self._lower_layers(toks, off, lens, self.embed.weight.detach(), tape=False)
            w = self._top_weights(table, detach=True)
        out = ops.gru_encode(toks, off, lens, w, self.use_bi_gru, no_txtnorm=self.no_txtnorm,
                             use_abs=self.use_abs, gather_last=last)
        return out, off, lens, mask

    def forward_packed_train(self, x, lengths, seeds=None):
        """The tower on the autograd tape, packed layout: -> (word / caption embeddings, tok_off, lens, mask).  nn.Dropout on
        the word embeddings (TextEncoder.py:42; SGRAF: p = 0.4) in training mode: the rows are gathered on the tape, dropped,
        and fed to the GRU kernels as a dense "embedding table" indexed by arange (the VSRN region-GRU trick)."""
        from .. import autograd as ag
        toks, off, lens, mask = pack_tokens(x, lengths)
        table = self.embed.weight
        if self.dropout_p > 0 and self.training:
            if seeds is None:
                if not hasattr(self, '_seeds'):
                    self._seeds = ag.DropoutSeeds()
                self._seeds.new_step()
                seeds = self._seeds
            table = ag.dropout(ag.gather_rows(self.embed.weight, toks), self.dropout_p, seeds)
            toks = torch.arange(table.shape[0], device=table.device, dtype=torch.int64)
        if self.num_layers == 1:
            seq = ag.gru_sequence(toks, off, lens, table, dict(self.rnn.named_parameters()), self.use_bi_gru)
        else:
            toks, table = self._lower_layers(toks, off, lens, table, tape=True)
            top = {k[len('rnn.'):]: v for k, v in self._top_weights(table, detach=False).items() if k.startswith('rnn.')}
            seq = ag.gru_sequence(toks, off, lens, table, top, self.use_bi_gru)
        if self.method_name in ('VSE++', 'VSRN'):
            last = off + ops.h2d(np.asarray(lens, np.int64), off.device) - 1
            seq = ag.gather_rows(seq, last)
        if not self.no_txtnorm:
            seq = ag.l2norm_rows(seq)
        if self.use_abs:                       # TextEncoder.py:66-68
            seq = seq.abs()
        return seq, off, lens, mask

    def forward(self, x, lengths):
        """One module, two modes (see EncoderImagePrecomp.forward): in training mode the tape implementation (gradients, live
        dropout), in evaluation mode the fused kernels.  Returns the reference's layout either way."""
        from .ImgEncoder import _on_tape
        train_path = self.training and (_on_tape(self, x) or self.dropout_p > 0)
        out, off, lens, mask = self.forward_packed_train(x, lengths) if train_path else self.forward_packed(x, lengths)
        cap_len = torch.as_tensor(lens, dtype=torch.int64)
        if self.method_name in ('VSE++', 'VSRN'):
            return out, cap_len
        B, L = x.shape[0], max(lens)
        cap_emb = torch.zeros(B, L, self.embed_size, device=x.device, dtype=torch.float32)
        cap_emb[mask[:, :L]] = out                    # scatter back to the padded layout (plumbing; differentiable on the tape)
        return cap_emb, cap_len


class BertMapping(nn.Module):
    """SAEM text tower: frozen BERT + conv / pooling / transformer head + Linear + F.normalize
    (TextEncoder.py:74-157).  txt_stru='rnn' cannot run in the reference either and is not built: with bi_gru it slices
    by a float (TextEncoder.py:135), and without it pack_padded_sequence (TextEncoder.py:128) rejects the batch,
    because the BERT collate leaves the mask-sum lengths unsorted (every id row has max_words entries, data_loader.py:146-157)."""

    def __init__(self, config):
        super().__init__()
        bert_config = BertConfig.from_json_file(config['bert_config_file'])
        self.bert = BertModel(bert_config)
        self.bert.load_state_dict(torch.load(config['init_checkpoint'], map_location='cpu'))
        self.freeze_layers(self.bert)
        self.txt_stru = config['txt_stru']
        if config['txt_stru'] == 'pooling':
            self.mapping_0 = nn.Linear(bert_config.hidden_size, bert_config.hidden_size)
            self.mapping = nn.Linear(bert_config.hidden_size, config['final_dims'])
        elif config['txt_stru'] == 'cnn':
            Ks = [1, 2, 3]
            self.convs1 = nn.ModuleList([nn.Conv2d(1, 512, (K, bert_config.hidden_size)) for K in Ks])
            self.mapping = nn.Linear(len(Ks) * 512, config['final_dims'])
        elif config['txt_stru'] == 'trans':
            trans_config = BertConfig.from_json_file(config['trans_cfg'])
            self.layer = bert.BERTLayer(trans_config)
            self.mapping_0 = nn.Linear(bert_config.hidden_size, trans_config.hidden_size)
            self.mapping = nn.Linear(trans_config.hidden_size, config['final_dims'])
        elif config['txt_stru'] == 'rnn':
            raise NotImplementedError("txt_stru='rnn' is broken in the reference (float slicing, TextEncoder.py:135; "
                                      "unsorted lengths into pack_padded_sequence, TextEncoder.py:128)")
        else:
            raise ValueError("Unknown txt_stru: {}".format(config['txt_stru']))

    def forward(self, input_ids, attention_mask, token_type_ids, lengths):
        all_encoder_layers, _ = self.bert(input_ids, token_type_ids=token_type_ids, attention_mask=attention_mask)
        last = all_encoder_layers[-1]
        B, L, H = last.shape
        if self.txt_stru == 'pooling':
            output = ops.mean_mid(ops.linear(last, self.mapping_0.weight.detach(), self.mapping_0.bias.detach()))
        elif self.txt_stru == 'cnn':
            flat = last.reshape(B * L, H)
            C = self.convs1[0].out_channels
            output = torch.empty(B, C * len(self.convs1), device=last.device, dtype=torch.float32)
            for n, conv in enumerate(self.convs1):
                K = conv.kernel_size[0]
                if L < K:
                    raise ValueError("sequence shorter than the conv window")
                buf = torch.empty(B * L, C, device=last.device, dtype=torch.float32)
                # Conv2d(1, C, (K, H)) == GEMM over K consecutive token rows (rows overlap: lda = H, K_dim = K*H)
                ops.linear_strided(flat, H, B * L - (K - 1), K * H, conv.weight.detach().reshape(C, K * H),
                                   conv.bias.detach(), None, out=buf)
                ops.relu_maxpool(buf.view(B, L, C), L - K + 1, out=output[:, n * C:(n + 1) * C])
        else:  # trans
            hidden = ops.linear(last, self.mapping_0.weight.detach(), self.mapping_0.bias.detach())
            hidden = self.layer(hidden, attention_mask.to(last.device).to(torch.float32))
            output = ops.mean_mid(hidden)
        code = ops.linear(output, self.mapping.weight.detach(), self.mapping.bias.detach())
        return ops.normalize(code, dim=1)

    def forward_train(self, input_ids, attention_mask, token_type_ids, lengths, seeds):
        """SAEM.train_emb: frozen BERT in training mode (no gradient, live dropout) -> head on the autograd tape ->
        nn.Dropout(hidden_dropout_prob) -> mapping -> F.normalize (TextEncoder.py:115-152)."""
        from .. import autograd as ag
        last = self.bert.forward_frozen_train(input_ids, token_type_ids, attention_mask, seeds)
        B, L, H = last.shape
        p_drop = float(self.bert.config.hidden_dropout_prob)
        if self.txt_stru == 'pooling':
            output = ag.mean_mid(ag.linear(last.reshape(B * L, H), self.mapping_0.weight, self.mapping_0.bias).view(B, L, -1))
        elif self.txt_stru == 'cnn':
            pooled = []
            for conv in self.convs1:
                K = conv.kernel_size[0]
                if L < K:
                    raise ValueError("sequence shorter than the conv window")
                npos = L - K + 1
                # Conv2d(1, C, (K, H)) on the frozen features: unfold K consecutive token rows (a copy), then one GEMM
                unf = torch.cat([last[:, k:k + npos] for k in range(K)], 2).reshape(B * npos, K * H)
                y = ag.linear(unf, conv.weight.reshape(conv.out_channels, K * H), conv.bias)
                pooled.append(ag.relu_maxpool(y.view(B, npos, conv.out_channels)))
            output = torch.cat(pooled, 1)
        else:  # trans
            hidden = ag.linear(last.reshape(B * L, H), self.mapping_0.weight, self.mapping_0.bias).view(B, L, -1)
            hidden = self.layer.forward_train(hidden, attention_mask.to(last.device).to(torch.float32), seeds, training=self.training)
            output = ag.mean_mid(hidden)
        output = ag.dropout(output, p_drop if self.txt_stru != 'trans' else float(self.layer.p_hidden), seeds, self.training)
        code = ag.linear(output, self.mapping.weight, self.mapping.bias)
        return ag.l2norm_rows(code, eps=1e-12)

    def freeze_layers(self, model):
        for child in model.children():
            for param in child.parameters():
                param.requires_grad = False


from .camera_ import AGSA, bn_affine  # noqa: E402


class CAMERAEncoderText(nn.Module):
    """CAMERA text tower (TextEncoder.py:162-197): frozen BERT -> Linear -> AGSA -> MLP + BatchNorm residual ->
    mean over ALL token positions (the reference does not mask padding) -> F.normalize."""

    def __init__(self, cfg_file, init_ckpt, embed_size, head, drop=0.0):
        super().__init__()
        bert_config = BertConfig.from_json_file(cfg_file)
        self.bert = BertModel(bert_config)
        self.bert.load_state_dict(torch.load(init_ckpt, map_location='cpu'))
        self.freeze_layers(self.bert)
        self.mapping = nn.Linear(bert_config.hidden_size, embed_size)
        self.agsa = AGSA(1, embed_size, h=head, is_share=False, drop=drop)
        self.fc1 = nn.Linear(embed_size, embed_size)
        self.fc2 = nn.Linear(embed_size, embed_size)
        self.bn = nn.BatchNorm1d(embed_size)
        self.drop = float(drop or 0.0)

    def forward(self, input_ids, attention_mask, token_type_ids, lengths=None):
        all_encoder_layers, _ = self.bert(input_ids, token_type_ids=token_type_ids, attention_mask=attention_mask)
        x = ops.linear(all_encoder_layers[-1], self.mapping.weight.detach(), self.mapping.bias.detach())
        agsa_emb = self.agsa(x)
        h = ops.linear(agsa_emb, self.fc1.weight.detach(), self.fc1.bias.detach(), act='relu')
        h = ops.linear(h, self.fc2.weight.detach(), self.fc2.bias.detach())
        sc, sh = bn_affine(self.bn)
        x = ops.affine_cols(h, sc, sh, residual=agsa_emb)
        return ops.normalize(ops.mean_mid(x), dim=-1)

    def forward_train(self, input_ids, attention_mask, token_type_ids, seeds):
        """CAMERA.train_emb: frozen BERT in training mode -> mapping -> AGSA -> MLP + BatchNorm residual -> mean -> F.normalize
        on the autograd tape (TextEncoder.py:181-192)."""
        from .. import autograd as ag
        last = self.bert.forward_frozen_train(input_ids, token_type_ids, attention_mask, seeds)
        B, L, H = last.shape
        x = ag.linear(last.reshape(B * L, H), self.mapping.weight, self.mapping.bias).view(B, L, -1)
        agsa_emb = self.agsa.forward_train(x, None, seeds, self.training)
        E = agsa_emb.shape[-1]
        h = ag.act(ag.linear(agsa_emb.reshape(B * L, E), self.fc1.weight, self.fc1.bias), 'relu')
        h = ag.linear(ag.dropout(h, self.drop, seeds, self.training), self.fc2.weight, self.fc2.bias)
        h = ag.batch_norm_train(h, self.bn).view(B, L, E)
        x = agsa_emb + ag.dropout(h, self.drop, seeds, self.training)
        return ag.l2norm_rows(ag.mean_mid(x), eps=1e-12)

    def freeze_layers(self, model):
        for child in model.children():
            for param in child.parameters():
                param.requires_grad = False
