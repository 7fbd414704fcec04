"""Image towers with the reference's constructor / forward signatures
(itr/modalmodule/ImgEncoder.py); parameters live in torch modules (same state_dict names so
reference checkpoints load), arithmetic runs in the HIP kernels."""
from collections import OrderedDict

import numpy as np
import torch
from torch import nn

from .. import ops
from . import bert


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
        if self.use_abs and self.no_imgnorm:
            raise NotImplementedError("use_abs without normalisation")
        return ops.proj_l2norm(images, self._weight().detach(), self.fc.bias.detach(), self.no_imgnorm, self.use_abs)

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

    def load_state_dict(self, state_dict):
        own_state = self.state_dict()
        new_state = OrderedDict((k, v) for k, v in state_dict.items() if k in own_state)
        super().load_state_dict(new_state)
