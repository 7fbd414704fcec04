"""CAMERA blocks with the reference's module / parameter names (itr/modalmodule/camera_.py): gated query
attention (AGSA), dilated-convolution multi-view summarisation, box position encoder.  Parameter containers +
HIP forward (evaluation mode: BatchNorm running statistics, dropout off)."""
import copy
import math

import torch
from torch import nn

from .. import ops
from ..settings import SETTINGS


def clones(module, N):
    return nn.ModuleList([copy.deepcopy(module) for _ in range(N)])


def bn_affine(bn):
    """eval-mode BatchNorm1d as y = x * scale + shift (parameter prep)."""
    scale = (bn.weight / torch.sqrt(bn.running_var + bn.eps)).detach()
    shift = (bn.bias - bn.running_mean * scale).detach()
    return scale.contiguous(), shift.contiguous()


def _lin(x, m, act=None):
    return ops.linear(x, m.weight.detach(), m.bias.detach(), act=act)


class GatedQueryAttLayer(nn.Module):
    """camera_.py:14-54."""

    def __init__(self, embed_size, h, is_share, drop=None):
        super().__init__()
        self.is_share = is_share
        self.h = h
        self.embed_size = embed_size
        self.d_k = embed_size // h
        self.drop_p = drop
        if is_share:
            self.linear = nn.Linear(embed_size, embed_size)
            self.linears = [self.linear, self.linear, self.linear]
        else:
            self.linears = clones(nn.Linear(embed_size, embed_size), 3)
        self.fc_q = nn.Linear(self.d_k, self.d_k)
        self.fc_k = nn.Linear(self.d_k, self.d_k)
        self.fc_g = nn.Linear(self.d_k, self.d_k * 2)

    def forward(self, inp, mask=None):
        if mask is not None:
            raise NotImplementedError("AGSA is always called without a mask in the reference")
        B, L, D = inp.shape
        dk = self.d_k
        q, k, v = [_lin(inp.reshape(B * L, D), l) for l in self.linears]       # (B*L, D) each
        q2, k2 = q.view(-1, dk), k.view(-1, dk)                                # (B*L*h, dk): heads are contiguous
        if dk in (16, 32) and SETTINGS.agsa_fused:                             # one kernel for the gate (csrc/agsa_gate.hip)
            q2, k2 = ops.agsa_gate(q2, k2, (self.fc_q.weight, self.fc_q.bias), (self.fc_k.weight, self.fc_k.bias),
                                   (self.fc_g.weight, self.fc_g.bias))
        else:
            G = ops.mul_rows(_lin(q2, self.fc_q), _lin(k2, self.fc_k))         # fc_q(query) * fc_k(key)   :38
            M = _lin(G, self.fc_g, act='sigmoid')                              # (B*L*h, 2*dk)              :39
            q2 = ops.mul_rows(q2, M[:, :dk])
            k2 = ops.mul_rows(k2, M[:, dk:])
        x = ops.mha_small(q2.view(B * L, D), k2.view(B * L, D), v, None, B, L, self.h, dk, 1.0 / math.sqrt(dk))
        return x.view(B, L, D)

    def forward_train(self, inp, seeds, training=True):
        """The layer on the autograd tape (CAMERA.train_emb); dropout on the attention probabilities when drop > 0 (:49-50)."""
        from .. import autograd as ag
        B, L, D = inp.shape
        dk = self.d_k
        x2 = inp.reshape(B * L, D)
        q, k, v = [ag.linear(x2, l.weight, l.bias) for l in self.linears]
        q2, k2 = q.view(-1, dk), k.view(-1, dk)
        G = ag.mul(ag.linear(q2, self.fc_q.weight, self.fc_q.bias), ag.linear(k2, self.fc_k.weight, self.fc_k.bias))
        M = ag.act(ag.linear(G, self.fc_g.weight, self.fc_g.bias), 'sigmoid')
        qg, kg = ag.gate_apply(q2, k2, M)
        qkv = torch.cat([qg.view(B * L, D), kg.view(B * L, D), v], 1)
        p = float(self.drop_p or 0.0)
        return ag.mha(qkv, None, B, L, self.h, p if training else 0.0, seeds.next()).view(B, L, D)


class AGSA(nn.Module):
    """Adaptive Gating Self-Attention (camera_.py:57-89)."""

    def __init__(self, num_layers, embed_size, h=1, is_share=False, drop=None):
        super().__init__()
        self.num_layers = num_layers
        self.bns = clones(nn.BatchNorm1d(embed_size), num_layers)
        self.drop = float(drop or 0.0)          # nn.Dropout(drop) after every BatchNorm (camera_.py:63, :82, :88)
        self.is_share = is_share
        self.h = h
        self.embed_size = embed_size
        self.att_layers = clones(GatedQueryAttLayer(embed_size, h, is_share, drop=drop), num_layers)

    def forward(self, rgn_emb, pos_emb=None, mask=None):
        bs, num_r, emb_dim = rgn_emb.shape
        x = rgn_emb if pos_emb is None else ops.mul_rows(rgn_emb, pos_emb.reshape(bs * num_r, emb_dim))
        agsa_emb = rgn_emb
        for i in range(self.num_layers):
            x = self.att_layers[i](x if i == 0 else agsa_emb, mask)
            sc, sh = bn_affine(self.bns[i])
            agsa_emb = ops.affine_cols(x, sc, sh, residual=agsa_emb)          # rgn + bn(att)
        return agsa_emb

    def forward_train(self, rgn_emb, pos_emb, seeds, training=True):
        """AGSA on the autograd tape with BatchNorm batch statistics and the dropout sites live (camera_.py:69-89)."""
        from .. import autograd as ag
        bs, num_r, emb_dim = rgn_emb.shape
        x = rgn_emb if pos_emb is None else ag.mul(rgn_emb, pos_emb)
        agsa_emb = rgn_emb
        for i in range(self.num_layers):
            x = self.att_layers[i].forward_train(x if i == 0 else agsa_emb, seeds, training)
            x = ag.batch_norm_train(x.reshape(bs * num_r, emb_dim), self.bns[i]).view(bs, num_r, emb_dim)
            agsa_emb = agsa_emb + ag.dropout(x, self.drop, seeds, training)
        return agsa_emb


class Summarization(nn.Module):
    """Multi-View Summarization (camera_.py:93-114): 7 dilated Conv1d over the region axis -> ReLU -> concat
    (1024 channels) -> Linear(1024, k).  Each convolution tap is one accumulate-GEMM over a zero-padded copy of
    the region sequence."""
    PAD = 6

    def __init__(self, embed_size, smry_k):
        super().__init__()
        out_c = [256, 128, 128, 128, 128, 128, 128]
        k_size = [1, 3, 3, 3, 5, 5, 5]
        dila = [1, 1, 2, 3, 1, 2, 3]
        pads = [0, 1, 2, 3, 2, 4, 6]
        self.convs_dilate = nn.ModuleList([nn.Conv1d(embed_size, out_c[i], k_size[i], dilation=dila[i], padding=pads[i])
                                           for i in range(len(out_c))])
        self.convs_fc = nn.Linear(1024, smry_k)

    def forward(self, rgn_emb):
        B, R, D = rgn_emb.shape
        P = self.PAD
        Rp = R + 2 * P
        xp = torch.zeros(B, Rp, D, device=rgn_emb.device, dtype=torch.float32)
        xp[:, P:P + R] = rgn_emb                                                # zero padding (plumbing)
        xflat = xp.view(B * Rp, D)
        M = B * Rp - 2 * P
        cat = torch.zeros(B * Rp, 1024, device=rgn_emb.device, dtype=torch.float32)
        col = 0
        for conv in self.convs_dilate:
            ks, dil, oc = conv.kernel_size[0], conv.dilation[0], conv.out_channels
            dst = cat[P:, col:col + oc]
            for j in range(ks):
                off = (j - (ks - 1) // 2) * dil
                wj = conv.weight.detach()[:, :, j].contiguous()
                src = xflat[P + off:]
                last = (j == ks - 1)
                if j == 0:
                    ops.linear_strided(src, D, M, D, wj, conv.bias.detach(), 'relu' if last else None, out=dst)
                else:
                    ops.gemm_acc(src, D, M, D, wj, None, dst, act='relu' if last else None)
            col += oc
        smry = _lin(cat, self.convs_fc)                                          # (B*Rp, k)
        return smry.view(B, Rp, -1)[:, P:P + R].contiguous()

    def forward_train(self, rgn_emb):
        """On the autograd tape: every dilated convolution is an unfold copy of its taps (zero padded) + ONE GEMM + relu."""
        from .. import autograd as ag
        B, R, D = rgn_emb.shape
        P = self.PAD
        xp = torch.nn.functional.pad(rgn_emb, (0, 0, P, P))                      # zero rows before / after the regions
        outs = []
        for conv in self.convs_dilate:
            ks, dil, oc = conv.kernel_size[0], conv.dilation[0], conv.out_channels
            taps = [xp[:, P + (j - (ks - 1) // 2) * dil:P + (j - (ks - 1) // 2) * dil + R] for j in range(ks)]
            unf = torch.cat(taps, 2).reshape(B * R, ks * D)
            w2 = conv.weight.permute(0, 2, 1).reshape(oc, ks * D)                # [oc, D, ks] -> [oc, ks * D]
            outs.append(ag.act(ag.linear(unf, w2, conv.bias), 'relu'))
        cat = torch.cat(outs, 1)                                                 # (B*R, 1024)
        return ag.linear(cat, self.convs_fc.weight, self.convs_fc.bias).view(B, R, -1)


class PositionEncoder(nn.Module):
    """camera_.py:131-147."""

    def __init__(self, embed_dim, posi_dim=6):
        super().__init__()
        self.proj = nn.Linear(posi_dim, embed_dim)

    def forward(self, boxes, imgs_wh):
        posi = ops.camera_posenc(boxes, imgs_wh)                                 # (bs, num_r, 6)
        return _lin(posi, self.proj, act='sigmoid')

    def forward_train(self, boxes, imgs_wh):
        from .. import autograd as ag
        posi = ops.camera_posenc(boxes, imgs_wh)
        B, R, _ = posi.shape
        return ag.act(ag.linear(posi.reshape(B * R, -1), self.proj.weight, self.proj.bias), 'sigmoid').view(B, R, -1)
