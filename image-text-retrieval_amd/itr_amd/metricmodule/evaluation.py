"""Evaluation harness with the reference's signatures (itr/metricmodule/evaluation.py:75-259):
encode_data, cal_sims, i2t, t2i, cal_recall -- scoring and ranking on the HIP kernels.

Differences that are deliberate (DESIGN.md "reference quirks"):
  * cal_sims slices `lengths` per caption shard.  The reference passes the un-sliced array, so every caption
    shard j > 0 is scored with the lengths of shard 0 (evaluation.py:149);
    `ref_quirk_unsliced_lengths=True` reproduces that bit of behaviour.
  * i2t / t2i count instead of sorting; ranks are identical except on exact score ties, where numpy's
    unstable argsort makes the reference implementation-defined.
"""
import time
from collections import OrderedDict

import numpy as np
import torch

from .. import ops


class AverageMeter(object):
    """Computes and stores the average and current value (evaluation.py:15-40)."""

    def __init__(self):
        self.reset()

    def reset(self):
        self.val = 0
        self.avg = 0
        self.sum = 0
        self.count = 0

    def update(self, val, n=0):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / (.0001 + self.count)

    def __str__(self):
        if self.count == 0:
            return str(self.val)
        return '%.4f (%.4f)' % (self.val, self.avg)


class LogCollector(object):
    """A collection of logging objects that can change from train to val (evaluation.py:43-72)."""

    def __init__(self):
        self.meters = OrderedDict()

    def update(self, k, v, n=0):
        if k not in self.meters:
            self.meters[k] = AverageMeter()
        self.meters[k].update(v, n)

    def __str__(self):
        s = ''
        for i, (k, v) in enumerate(self.meters.items()):
            if i > 0:
                s += '  '
            if k == 'lr':
                v = '{:.3e}'.format(v.val)
            s += k + ' ' + str(v)
        return s

    def tb_log(self, tb_logger, prefix='', step=None):
        for k, v in self.meters.items():
            tb_logger.log_value(prefix + k, v.val, step=step)


def encode_data(model, data_loader, islength=False):
    """Encode all images and captions loadable by `data_loader` -> (img_embs, cap_embs, cap_lens) numpy arrays,
    exactly the reference's layout (evaluation.py:75-121): row `ids[k]` of every array belongs to dataset
    item ids[k]; word-level caption embeddings are zero-padded to the longest caption."""
    val_logger = LogCollector()
    model.val_start()
    max_n_word = 0
    no_init = True
    if islength:
        for (_, _, _, _, lengths_, _, _, _) in data_loader:
            max_n_word = max(max_n_word, int(lengths_[0]))
    img_embs = cap_embs = cap_lens = None
    for batch_data in data_loader:
        model.logger = val_logger
        images, boxes, imgs_wh, captions, lengths, ids, captions_mask, captions_type_ids = batch_data
        with torch.no_grad():
            emd_list = model.forward_emb(images=images, boxes=boxes, imgs_wh=imgs_wh, captions=captions,
                                         lengths=lengths, ids=ids, captions_mask=captions_mask,
                                         captions_type_ids=captions_type_ids)
        img_emb, cap_emb = emd_list[0], emd_list[1]
        if no_init:
            no_init = False
            n = len(data_loader.dataset)
            ima_size = [n] + list(img_emb.size()[1:])
            cap_size = [n] + list(cap_emb.size()[1:])
            if islength:
                cap_size[1] = max_n_word
            img_embs = np.zeros(ima_size, dtype=np.float32)
            cap_embs = np.zeros(cap_size, dtype=np.float32)
            cap_lens = np.zeros(n, dtype=np.int32)
        ids = list(ids)
        img_embs[ids] = img_emb.detach().cpu().numpy()
        if cap_emb.dim() == 3:
            cap_embs[ids, :cap_emb.size(1)] = cap_emb.detach().cpu().numpy()
        else:
            cap_embs[ids] = cap_emb.detach().cpu().numpy()
        cap_lens[ids] = [int(l) for l in lengths]
    return img_embs, cap_embs, cap_lens


def _cal_fun(model):
    if model.config['name'] in ['CAMERA']:
        return model.mvm
    return model.sim_enc if model.sim_enc is not None else model.criterion.sim


def cal_sims(model, img_embs, cap_embs, lengths=None, shard_size=128, ref_quirk_unsliced_lengths=False):
    """(n_img, n_cap) float64 similarity matrix (evaluation.py:124-153).  Same tiling loop as the reference
    so that `shard_size` keeps its meaning; each tile is one kernel launch on device-resident blocks."""
    cal_fun = _cal_fun(model)
    n_img, n_cap = len(img_embs), len(cap_embs)
    t0 = time.time()
    dev = torch.device('cuda', torch.cuda.current_device())
    img_all = torch.from_numpy(np.ascontiguousarray(img_embs)).to(dev)
    cap_all = torch.from_numpy(np.ascontiguousarray(cap_embs)).to(dev)
    d = np.zeros((n_img, n_cap))
    for i0 in range(0, n_img, shard_size):
        i1 = min(i0 + shard_size, n_img)
        for j0 in range(0, n_cap, shard_size):
            j1 = min(j0 + shard_size, n_cap)
            lens = lengths
            if lengths is not None and not ref_quirk_unsliced_lengths:
                lens = lengths[j0:j1]
            with torch.no_grad():
                sim = cal_fun(img_all[i0:i1], cap_all[j0:j1], lens, model.config)
            d[i0:i1, j0:j1] = sim.detach().cpu().numpy()
    print('Calculate similarity matrix elapses: {:.3f}s'.format(time.time() - t0))
    return d


def _ranks(sims):
    S = torch.from_numpy(np.ascontiguousarray(np.asarray(sims, dtype=np.float32))).cuda()
    i_rank, i_top, t_rank, t_best, _ = ops.rank_counts(S, 5)
    return (i_rank.cpu().numpy().astype(np.float64), i_top.cpu().numpy().astype(np.float64),
            t_rank.cpu().numpy().astype(np.float64), (t_best & 0xffffffff).cpu().numpy().astype(np.float64))


def i2t(sims, return_ranks=False):
    """Images->Text (evaluation.py:156-189): sims (N, 5N)."""
    ranks, top1, _, _ = _ranks(sims)
    out = ops.recall_from_ranks(ranks)
    return (out, (ranks, top1)) if return_ranks else out


def t2i(sims, return_ranks=False):
    """Text->Images (evaluation.py:192-222)."""
    _, _, ranks, top1 = _ranks(sims)
    out = ops.recall_from_ranks(ranks)
    return (out, (ranks, top1)) if return_ranks else out


def cal_recall(sims):
    """Result dict of evaluation.py:225-259."""
    r, rt = i2t(sims, return_ranks=True)
    ri, rti = t2i(sims, return_ranks=True)
    ar = (r[0] + r[1] + r[2]) / 3
    ari = (ri[0] + ri[1] + ri[2]) / 3
    rsum = r[0] + r[1] + r[2] + ri[0] + ri[1] + ri[2]
    print("rsum: %.1f" % rsum)
    print("Average i2t Recall: %.1f" % ar)
    print("Image to text: r1 %.1f; r5 %.1f; r10 %.1f; medr %.1f; meanr %.1f" % r)
    print("Average t2i Recall: %.1f" % ari)
    print("Text to image: r1 %.1f; r5 %.1f; r10 %.1f; medr %.1f; meanr %.1f" % ri)
    res = {'result': [list(r) + list(ri) + [ar, ari, rsum]], 'rsum': rsum, 'i2t_ave_r': ar, 'i2t_r1': r[0],
           'i2t_r5': r[1], 'i2t_r10': r[2], 'i2t_medr': r[3], 'i2t_meanr': r[4], 'i2t_ranks': rt[0],
           'i2t_top1': rt[1], 't2i_ave_r': ari, 't2i_r1': ri[0], 't2i_r5': ri[1], 't2i_r10': ri[2],
           't2i_medr': ri[3], 't2i_meanr': ri[4], 't2i_ranks': rti[0], 't2i_top1': rti[1]}
    return res
