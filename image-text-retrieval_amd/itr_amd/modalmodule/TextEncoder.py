"""Text towers with the reference's signatures (itr/modalmodule/TextEncoder.py)."""
import numpy as np
import torch
from torch import nn

from .. import ops


def pack_tokens(x, lengths):
    """(B, L) padded ids + descending lengths -> packed ids (n_tok,), tok_off (B,), host lengths."""
    lens = [int(l) for l in lengths]
    B, L = x.shape
    ar = torch.arange(L, device=x.device).unsqueeze(0)
    lt = torch.as_tensor(lens, device=x.device).unsqueeze(1)
    mask = ar < lt
    toks = x[mask]                                   # row-major: caption by caption
    off = torch.as_tensor(np.concatenate([[0], np.cumsum(lens)[:-1]]), dtype=torch.int64, device=x.device)
    return toks.contiguous(), off, lens, mask


class EncoderText(nn.Module):
    """Embedding -> (bi)GRU -> direction average -> [last valid step] -> [l2norm] -> [abs]
    (TextEncoder.py:15-70).  `forward` returns the padded (B, L, D) tensor like the reference;
    `forward_packed` returns the packed (n_tok, D) layout the SCAN / SGRAF kernels consume."""

    def __init__(self, vocab_size, word_dim, embed_size, num_layers, use_bi_gru=False, no_txtnorm=False,
                 dropout=0., use_abs=False, method_name=None):
        super().__init__()
        if num_layers != 1:
            raise NotImplementedError("only num_layers == 1 (every config of the reference) is built")
        self.embed_size = embed_size
        self.no_txtnorm = no_txtnorm
        self.use_abs = use_abs
        self.method_name = method_name
        self.embed = nn.Embedding(vocab_size, word_dim)
        self.dropout_p = dropout          # identity in eval mode; training backward is SURVEY 8(f)-3
        self.use_bi_gru = use_bi_gru
        self.rnn = nn.GRU(word_dim, embed_size, num_layers, batch_first=True, bidirectional=use_bi_gru)
        self.init_weights()

    def init_weights(self):
        self.embed.weight.data.uniform_(-0.1, 0.1)

    def _weights(self):
        w = {'embed.weight': self.embed.weight.detach()}
        w.update({'rnn.' + k: v.detach() for k, v in self.rnn.named_parameters()})
        return w

    def forward_packed(self, x, lengths):
        toks, off, lens, mask = pack_tokens(x, lengths)
        last = self.method_name in ('VSE++', 'VSRN')
        out = ops.gru_encode(toks, off, lens, self._weights(), self.use_bi_gru, no_txtnorm=self.no_txtnorm,
                             use_abs=self.use_abs, gather_last=last)
        return out, off, lens, mask

    def forward(self, x, lengths):
        if self.training and self.dropout_p > 0:
            raise NotImplementedError("training-mode dropout/backward is not built (SURVEY 8f-3); call val_start()")
        out, off, lens, mask = self.forward_packed(x, lengths)
        cap_len = torch.as_tensor(lens, dtype=torch.int64)
        if self.method_name in ('VSE++', 'VSRN'):
            return out, cap_len
        B, L = x.shape[0], max(lens)
        cap_emb = torch.zeros(B, L, self.embed_size, device=x.device, dtype=torch.float32)
        cap_emb[mask[:, :L]] = out                    # scatter back to the padded layout (plumbing)
        return cap_emb, cap_len
