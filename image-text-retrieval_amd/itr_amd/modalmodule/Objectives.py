"""Similarity measures and ranking losses with the reference's names and signatures
(itr/modalmodule/Objectives.py), computed by the HIP kernels."""
import torch
from torch import nn

from .. import ops


def cosine_similarity(x1, x2, dim=1, eps=1e-8):
    """w12 / clamp(|x1||x2|, min=eps) along `dim` (Objectives.py:10-15).  Small glue op kept for API parity;
    the SCAN kernel evaluates this expression in its epilogue."""
    w12 = torch.sum(x1 * x2, dim)
    w1 = torch.norm(x1, 2, dim)
    w2 = torch.norm(x2, 2, dim)
    return (w12 / (w1 * w2).clamp(min=eps)).squeeze()


def cosine_sim(im, s, *args):
    """Cosine similarity between all image / sentence pairs (Objectives.py:18-21)."""
    return ops.cosine_scores(im, s)


def order_sim(im, s, *args):
    """Order-embedding similarity -|max(0, s - im)|_2 (Objectives.py:24-30)."""
    return ops.order_scores(im, s)


def pdist(x1, x2, *args):
    """SAEM's euclidean distance, used as the "similarity" when measure='order' (Objectives.py:54-56, :297-307)."""
    return ops.pdist(x1, x2)


def pdist_cos(x1, x2, *args):
    """SAEM cosine (Objectives.py:310-323)."""
    return ops.pdist_cos(x1, x2)


def _xattn(images, captions, cap_lens, config, cross_attn):
    return ops.scan_xattn_padded(images, captions, cap_lens, cross_attn=cross_attn,
                                 raw_feature_norm=config['raw_feature_norm'], agg_func=config['agg_func'],
                                 lambda_lse=config['lambda_lse'], lambda_softmax=config['lambda_softmax'])


def xattn_score_t2i(images, captions, cap_lens, config):
    """(n_image, n_regions, d), (n_caption, max_n_word, d), lengths -> (n_image, n_caption)
    (Objectives.py:329-372)."""
    return _xattn(images, captions, cap_lens, config, 't2i')


def xattn_score_i2t(images, captions, cap_lens, config):
    """Objectives.py:376-417."""
    return _xattn(images, captions, cap_lens, config, 'i2t')


class ContrastiveLoss(nn.Module):
    """Bidirectional hinge loss, optionally with hardest negatives (Objectives.py:34-115).
    `self.sim` is dispatched on config['name'] exactly like the reference (:45-74)."""

    def __init__(self, config, margin=0, measure=None, max_violation=False):
        super().__init__()
        self.config = config
        self.margin = margin
        self.max_violation = max_violation
        if measure == 'order':
            self.sim = order_sim
        elif measure == 'cosine':
            self.sim = cosine_sim
        else:
            raise ValueError("unknown measure:", measure)
        if self.config['name'] == 'SAEM':
            self.sim = pdist if measure == 'order' else pdist_cos
        elif self.config['name'] == 'SCAN':
            if self.config['cross_attn'] == 't2i':
                self.sim = xattn_score_t2i
            elif self.config['cross_attn'] == 'i2t':
                self.sim = xattn_score_i2t
            else:
                raise ValueError("unknown first norm type:", self.config['raw_feature_norm'])
        elif self.config['name'] == 'SGRAF':
            self.sim = lambda x, y, m, n: x

    def _sim_on_tape(self, im, s, s_l):
        """self.sim with gradients: the same measures as autograd nodes (HIP forward + backward kernels, itr_amd/autograd.py)."""
        from .. import autograd as ag
        import numpy as np
        name, cfg = self.config['name'], self.config
        if name == 'SGRAF':
            return im                                       # the similarity matrix comes from sim_enc (Objectives.py:73-74)
        if name == 'SCAN':
            Nc, L, D = s.shape
            lens = [int(x) for x in s_l][:Nc]
            flat = np.concatenate([b * L + np.arange(l, dtype=np.int64) for b, l in enumerate(lens)])
            words = s.reshape(Nc * L, D).index_select(0, ops.h2d(flat, s.device))     # padded -> packed, on the tape
            off = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.int64)
            fn = ag.scan_t2i_scores if cfg['cross_attn'] == 't2i' else ag.scan_i2t_scores
            return fn(im, words, off, lens, cfg['raw_feature_norm'], cfg['agg_func'], cfg['lambda_lse'], cfg['lambda_softmax'])
        if name == 'SAEM':
            raise NotImplementedError("SAEM's loss terms are composed in SAEM.forward_loss / train_emb")
        return ag.order_scores(im, s) if self.sim is order_sim else ag.cosine_scores(im, s)

    def forward(self, im, s=None, s_l=None):
        if torch.is_grad_enabled() and (im.requires_grad or (torch.is_tensor(s) and s.requires_grad)):
            scores = self._sim_on_tape(im, s, s_l)
        else:
            scores = self.sim(im, s, s_l, self.config)
        return ops.hinge_loss(scores, self.margin, self.max_violation)


class TripletLoss(nn.Module):
    """CAMERA's loss on a given score matrix (Objectives.py:482-517)."""

    def __init__(self, margin=0, max_violation=False):
        super().__init__()
        self.margin = margin
        self.max_violation = max_violation

    def forward(self, scores):
        return ops.hinge_loss(scores, self.margin, self.max_violation)


class DiversityRegularization(nn.Module):
    """CAMERA diversity regulariser (Objectives.py:521-542): sum over images of || Sn^T Sn - I ||_F^2 with the columns of the
    summarisation matrix l2-normalised over the regions -- one HIP kernel each way (csrc/aux_loss.hip)."""

    def __init__(self, smry_k, batch_size):
        super().__init__()
        self.smry_k = smry_k
        self.batch_size = batch_size

    def forward(self, smry_mat):
        from .. import autograd as ag
        if smry_mat.shape[-1] != self.smry_k:
            raise RuntimeError("DiversityRegularization: %d views, built for %d" % (smry_mat.shape[-1], self.smry_k))
        return ag.diversity_reg(smry_mat)


class AngularLoss(nn.Module):
    """SAEM angular loss (Objectives.py:238-290): per anchor, x_j = 4 ab (a + p).n_j - 2 (1 + ab) a.p over the other rows of the
    batch as negatives; soft-plus of the hardest (max_violation) or log(1 + sum exp).  Three GEMMs + a row kernel each way
    (csrc/aux_loss.hip)."""

    def __init__(self, l2_reg=0.02, angle_bound=1., lambda_ang=2, max_violation=True):
        super().__init__()
        self.l2_reg = l2_reg
        self.angle_bound = angle_bound
        self.lambda_ang = lambda_ang
        self.max_violation = max_violation

    def forward(self, im, s, s_l=None, ids=None):
        return self.angular_loss(im, s, s) + self.angular_loss(s, im, im)

    def angular_loss(self, anchors, positives, others):
        from .. import autograd as ag
        return ag.angular_loss(anchors, positives, others, self.angle_bound, self.max_violation)
