"""Image towers with the reference's constructor / forward signatures
(itr/modalmodule/ImgEncoder.py); parameters live in torch modules (same state_dict names so
reference checkpoints load), arithmetic runs in the HIP kernels."""
from collections import OrderedDict

import numpy as np
import torch
from torch import nn

from .. import ops
from . import bert


def _on_tape(module, *inputs):
    """True when a forward call must be differentiable: autograd is recording and an input or a parameter requires grad."""
    if not torch.is_grad_enabled():
        return False
    return any(torch.is_tensor(t) and t.requires_grad for t in inputs) or any(p.requires_grad for p in module.parameters())


class EncoderImagePrecomp(nn.Module):
    """fc + l2norm over precomputed region features (ImgEncoder.py:112-159)."""

    def __init__(self, img_dim, embed_size, no_imgnorm=False, precomp_enc_type='basic', use_abs=False):
        super().__init__()
        self.use_abs = use_abs
        self.embed_size = embed_size
        self.no_imgnorm = no_imgnorm
        if precomp_enc_type == 'basic':
            self.fc = nn.Linear(img_dim, embed_size)
            self.init_weights()
        elif precomp_enc_type == 'weight_norm':
            self.fc = torch.nn.utils.weight_norm(nn.Linear(img_dim, embed_size), dim=None)
        else:
            raise ValueError("Unknown precomp_enc_type: {}".format(precomp_enc_type))

    def init_weights(self):
        """Xavier initialisation of the fully connected layer (ImgEncoder.py:126-131)."""
        r = np.sqrt(6.) / np.sqrt(self.fc.in_features + self.fc.out_features)
        self.fc.weight.data.uniform_(-r, r)
        self.fc.bias.data.fill_(0)

    def _weight(self):
        fc = self.fc
        if hasattr(fc, 'weight_g'):   # weight_norm(dim=None): w = g * v / ||v||_F  (parameter prep, not hot path)
            return (fc.weight_g * fc.weight_v / fc.weight_v.norm()).contiguous()
        return fc.weight

    def forward(self, images):
        """One module, two modes: in training mode with autograd recording (a training step composed from the module seams like
        the reference's train_emb, Models.py:182-225) the tower runs on the tape (HIP forward AND backward nodes,
        itr_amd/autograd.py); in evaluation mode the fused kernel."""
        if self.training and _on_tape(self, images):
            return self.forward_train(images)
        if self.use_abs and self.no_imgnorm:
            raise NotImplementedError("use_abs without normalisation")
        return ops.proj_l2norm(images, self._weight().detach(), self.fc.bias.detach(), self.no_imgnorm, self.use_abs)

    def forward_train(self, images):
        from .. import autograd as ag
        x = ag.linear(images, self._weight(), self.fc.bias)
        if not self.no_imgnorm:
            x = ag.l2norm_rows(x)
        if self.use_abs:                       # order embeddings (ImgEncoder.py:143-145); elementwise glue on the tape
            x = x.abs()
        return x

    def load_state_dict(self, state_dict):
        """Accept a state_dict from the full-CNN model: keep only matching names (ImgEncoder.py:149-159)."""
        own_state = self.state_dict()
        new_state = OrderedDict((k, v) for k, v in state_dict.items() if k in own_state)
        super().load_state_dict(new_state)


class EncoderImagePooledPrecomp(EncoderImagePrecomp):
    """VSE++ on *_precomp data.  The reference is non-functional here (EncoderImagePrecomp has
    no region pooling, so 36 x 2048 input yields a 3-D embedding that `cosine_sim` cannot consume).
    Build decision: mean over the regions, then fc + l2norm (== the reference module applied to
    images.mean(1), which is what the golden vector g2 `out_2d` pins)."""

    def forward(self, images):
        if images.dim() == 3:
            images = ops.mean_mid(images)
        return super().forward(images)


from .vsrn_ import Rs_GCN  # noqa: E402


class EncoderImagePrecompAttn(nn.Module):
    """VSRN image tower (ImgEncoder.py:166-231): fc -> l2norm -> 4 x Rs_GCN -> l2norm -> GRU over the 36 regions ->
    last hidden state -> [BatchNorm1d on f30k] -> l2norm.  NB the two l2norm calls on region tensors use the reference's
    DEFAULT dim=1, i.e. they normalise across the regions (utils.py:11; :204, :216) -- reproduced.
    Returns (features (B, D), GCN_img_emd (B, N, D))."""

    def __init__(self, img_dim, embed_size, data_name, use_abs=False, no_imgnorm=False):
        super().__init__()
        self.embed_size = embed_size
        self.no_imgnorm = no_imgnorm
        self.use_abs = use_abs
        self.data_name = data_name
        self.fc = nn.Linear(img_dim, embed_size)
        self.init_weights()
        self.img_rnn = nn.GRU(embed_size, embed_size, 1, batch_first=True)
        self.Rs_GCN_1 = Rs_GCN(in_channels=embed_size, inter_channels=embed_size)
        self.Rs_GCN_2 = Rs_GCN(in_channels=embed_size, inter_channels=embed_size)
        self.Rs_GCN_3 = Rs_GCN(in_channels=embed_size, inter_channels=embed_size)
        self.Rs_GCN_4 = Rs_GCN(in_channels=embed_size, inter_channels=embed_size)
        if self.data_name == 'f30k_precomp':
            self.bn = nn.BatchNorm1d(embed_size)

    def init_weights(self):
        r = np.sqrt(6.) / np.sqrt(self.fc.in_features + self.fc.out_features)
        self.fc.weight.data.uniform_(-r, r)
        self.fc.bias.data.fill_(0)

    def forward(self, images):
        if self.training:                                # batch statistics, gradients: the tape implementation
            return self.forward_train(images)
        x = ops.linear(images, self.fc.weight.detach(), self.fc.bias.detach())
        if self.data_name != 'f30k_precomp':
            x = ops.l2norm(x, dim=1)
        for gcn in (self.Rs_GCN_1, self.Rs_GCN_2, self.Rs_GCN_3, self.Rs_GCN_4):
            x = gcn(x)
        gcn_emb = ops.l2norm(x, dim=1)
        B, N, D = gcn_emb.shape
        # the region GRU is the caption GRU kernel fed with a dense "embedding table": row r of the table IS region r,
        # every sequence is N long, the last state is what VSRN keeps (hidden_state[0], :219-222)
        w = {'embed.weight': gcn_emb.view(B * N, D)}
        for k, v in self.img_rnn.named_parameters():
            w['rnn.' + k] = v.detach()
        dev = gcn_emb.device
        tokens = torch.arange(B * N, device=dev, dtype=torch.int64)
        tok_off = torch.arange(B, device=dev, dtype=torch.int64) * N
        plain_tail = self.data_name != 'f30k_precomp'
        feat = ops.gru_encode(tokens, tok_off, [N] * B, w, False, no_txtnorm=not (plain_tail and not self.no_imgnorm),
                              use_abs=plain_tail and self.use_abs and not self.no_imgnorm, gather_last=True)
        if not plain_tail:
            bn = self.bn
            s = (bn.weight / torch.sqrt(bn.running_var + bn.eps)).detach()
            feat = ops.affine_cols(feat, s, (bn.bias - bn.running_mean * s).detach())
            if not self.no_imgnorm:
                feat = ops.l2norm(feat, dim=1)
        if self.use_abs and (not plain_tail or self.no_imgnorm):
            feat = feat.abs()
        return feat, gcn_emb

    def forward_train(self, images):
        """The tower on the autograd tape (VSRN.train_emb): GCN BatchNorms with batch statistics, region GRU through the caption
        GRU kernels with the GCN output as a dense table -> (features (B, D), GCN_img_emd (B, N, D))."""
        from .. import autograd as ag
        B, N, _ = images.shape
        x = ag.linear(images.reshape(B * N, -1), self.fc.weight, self.fc.bias).view(B, N, -1)
        if self.data_name != 'f30k_precomp':
            x = ag.l2norm_mid(x)
        for gcn in (self.Rs_GCN_1, self.Rs_GCN_2, self.Rs_GCN_3, self.Rs_GCN_4):
            x = gcn.forward_train(x)
        gcn_emb = ag.l2norm_mid(x)
        D = gcn_emb.shape[2]
        dev = gcn_emb.device
        tokens = torch.arange(B * N, device=dev, dtype=torch.int64)
        off = torch.arange(B, device=dev, dtype=torch.int64) * N
        seq = ag.gru_sequence(tokens, off, [N] * B, gcn_emb.reshape(B * N, D), dict(self.img_rnn.named_parameters()), False)
        feat = ag.gather_rows(seq, off + (N - 1))                                # hidden_state[0]: the state after the last region
        if self.data_name == 'f30k_precomp':
            feat = ag.batch_norm_train(feat, self.bn)
        if not self.no_imgnorm:
            feat = ag.l2norm_rows(feat, eps=1e-8)
        if self.use_abs:
            feat = feat.abs()
        return feat, gcn_emb


class TransformerMapping(nn.Module):
    """SAEM image tower: Linear(img_dim -> final_dims) -> one BERTLayer -> mean over regions -> F.normalize
    (ImgEncoder.py:324-350).  `trans_cfg` is not shipped with the reference: hidden_size must equal
    final_dims, the head size must be 16 / 32 / 64."""

    def __init__(self, config):
        super().__init__()
        self.config = config
        bert_config = bert.BertConfig.from_json_file(config['trans_cfg'])
        self.layer = bert.BERTLayer(bert_config)
        self.mapping = nn.Linear(config['img_dim'], config['final_dims'])

    def forward(self, x):
        x = ops.linear(x, self.mapping.weight.detach(), self.mapping.bias.detach())
        hidden_states = self.layer(x, None)                 # all-ones mask == no mask
        embed = ops.mean_mid(hidden_states)
        return ops.normalize(embed, dim=1)

    def forward_train(self, x, seeds):
        """The same tower on the autograd tape with the layer's dropout sites live (SAEM.train_emb)."""
        from .. import autograd as ag
        B, R, _ = x.shape
        h = ag.linear(x.reshape(B * R, -1), self.mapping.weight, self.mapping.bias).view(B, R, -1)
        h = self.layer.forward_train(h, None, seeds, training=self.training)
        return ag.l2norm_rows(ag.mean_mid(h), eps=1e-12)          # F.normalize: x / max(||x||, 1e-12)


from .camera_ import AGSA, Summarization, PositionEncoder  # noqa: E402


class EncoderImagePrecompSelfAttn(nn.Module):
    """CAMERA image tower (ImgEncoder.py:355-401).  NB the reference's l2norm calls use the DEFAULT dim=1, i.e.
    they normalise across the 36 regions, not across the features (:378, :384) -- reproduced."""

    def __init__(self, img_dim, embed_size, head, smry_k, drop=0.0):
        super().__init__()
        self.embed_size = embed_size
        self.fc = nn.Linear(img_dim, embed_size)
        self.init_weights()
        self.position_enc = PositionEncoder(embed_size)
        self.agsa = AGSA(1, embed_size, h=head, is_share=False, drop=drop)
        self.mvs = Summarization(embed_size, smry_k)

    def init_weights(self):
        r = np.sqrt(6.) / np.sqrt(self.fc.in_features + self.fc.out_features)
        self.fc.weight.data.uniform_(-r, r)
        self.fc.bias.data.fill_(0)

    def forward(self, images, boxes, imgs_wh):
        fc_img_emd = ops.linear(images, self.fc.weight.detach(), self.fc.bias.detach())
        fc_img_emd = ops.l2norm(fc_img_emd, dim=1)
        posi_emb = self.position_enc(boxes, imgs_wh)
        self_att_emb = self.agsa(fc_img_emd, posi_emb)
        self_att_emb = ops.l2norm(self_att_emb, dim=1)
        smry_mat = self.mvs(self_att_emb)
        return ops.camera_summarize(smry_mat, self_att_emb), smry_mat

    def forward_train(self, images, boxes, imgs_wh, seeds):
        """The tower on the autograd tape (CAMERA.train_emb): BatchNorm batch statistics, dropout sites live."""
        from .. import autograd as ag
        B, R, _ = images.shape
        fc = ag.l2norm_mid(ag.linear(images.reshape(B * R, -1), self.fc.weight, self.fc.bias).view(B, R, -1))
        pos = self.position_enc.forward_train(boxes, imgs_wh)
        att = ag.l2norm_mid(self.agsa.forward_train(fc, pos, seeds, self.training))
        smry_mat = self.mvs.forward_train(att)
        views = ag.summarize(smry_mat, att)                                       # softmax(smry, dim=1)^T att  -> (B, k, D)
        return ag.l2norm_rows(views, eps=1e-12), smry_mat                         # F.normalize(dim=-1)

    def load_state_dict(self, state_dict):
        own_state = self.state_dict()
        new_state = OrderedDict((k, v) for k, v in state_dict.items() if k in own_state)
        super().load_state_dict(new_state)
