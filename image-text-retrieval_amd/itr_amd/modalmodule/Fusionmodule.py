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
    (n_cap, L, D) padded, cap_lens -> (n_img, n_cap).  Evaluation mode only (BatchNorm running statistics, no
    dropout): the training-mode forward / backward is SURVEY.md 8(f)-3."""

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
        if self.training:
            raise NotImplementedError("EncoderSimilarity is built for evaluation mode; call val_start()")
        return ops.sgraf_padded(img_emb, cap_emb, cap_lens, self.state_dict(), self.module_name, self.sgr_step)

    def forward_packed(self, img_emb, words, plan):
        return ops.sgraf_scores(img_emb, words, plan, self.state_dict(), self.module_name, self.sgr_step)


class MultiViewMatching(nn.Module):
    """CAMERA: max over the k views (Fusionmodule.py:670-692)."""

    def forward(self, imgs, caps, *args, **kwargs):
        return ops.mvm_scores(imgs, caps)
