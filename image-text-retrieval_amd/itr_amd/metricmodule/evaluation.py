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
    """Rank vectors of a similarity matrix in the arithmetic the caller holds it in: float64 input (cal_sims' output,
    an ensemble average -- what the reference argsorts, evaluation.py:169, :209, :380) is counted in float64, so the
    indices equal the reference's even where two scores differ by less than an fp32 ulp; float32 input is counted in
    float32; anything else is widened to float64 (exact for every narrower type)."""
    if torch.is_tensor(sims):
        S = sims.detach()
        if S.dtype not in (torch.float32, torch.float64):
            S = S.to(torch.float64)
        S = S.contiguous().cuda()
    else:
        a = np.asarray(sims)
        if a.dtype != np.float32:
            a = a.astype(np.float64, copy=False)
        S = torch.from_numpy(np.ascontiguousarray(a)).cuda()
    if S.dtype == torch.float64:
        i_rank, i_top, t_rank, t_top = ops.rank_counts_f64(S, 5)
    else:
        i_rank, i_top, t_rank, t_best, _ = ops.rank_counts(S, 5)
        t_top = t_best & 0xffffffff
    return (i_rank.cpu().numpy().astype(np.float64), i_top.cpu().numpy().astype(np.float64),
            t_rank.cpu().numpy().astype(np.float64), t_top.cpu().numpy().astype(np.float64))


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


# ---------------------------------------------------------------------------------------------------------
# evalrank_single / evalrank_ensemble (evaluation.py:262-435): checkpoint -> loader -> encode -> score -> rank ->
# `<save_dir>/<data_name>[_5fold]_{single,ensemble}_result.yaml`.
_MEAN_KEYS = ('i2t_r1', 'i2t_r5', 'i2t_r10', 'i2t_medr', 'i2t_meanr', 't2i_r1', 't2i_r5', 't2i_r10', 't2i_medr', 't2i_meanr')


def _plain(res):
    """YAML-portable copy: numpy scalars / arrays -> python floats / lists (the reference dumps numpy objects,
    which needs yaml's unsafe loader to read back)."""
    out = {}
    for k, v in res.items():
        if isinstance(v, dict):
            out[k] = _plain(v)
        elif isinstance(v, np.ndarray):
            out[k] = [float(x) for x in v.tolist()]
        elif isinstance(v, (list, tuple)):
            out[k] = [[float(x) for x in row] if isinstance(row, (list, tuple, np.ndarray)) else
                      (float(row) if isinstance(row, (float, int, np.floating, np.integer)) else row) for row in v]
        elif isinstance(v, (np.floating, np.integer)):
            out[k] = float(v)
        else:
            out[k] = v
    return out


def _mean_metrics(res_dic):
    """fold5 averaging (evaluation.py:307-333).  The reference writes the means INTO the last fold's dict (the same
    object as PART_5), so its PART_5 entry shows the averages; here PART_5 keeps its own numbers and
    `Mean_metrics` is a dict of its own."""
    mean = tuple(np.array(res_dic['sum_result']).mean(axis=0).flatten())
    print("---------------------------------------------------------")
    print("--------------------- Mean metrics: ---------------------")
    print("rsum: %.1f" % (mean[10] * 6))
    print("Average i2t Recall: %.1f" % mean[11])
    print("Image to text: r1 %.1f; r5 %.1f; r10 %.1f; medr %.1f; meanr %.1f" % mean[:5])
    print("Average t2i Recall: %.1f" % mean[12])
    print("Text to image: r1 %.1f; r5 %.1f; r10 %.1f; medr %.1f; meanr %.1f" % mean[5:10])
    out = {'rsum': mean[10] * 6, 'i2t_ave_r': mean[11], 't2i_ave_r': mean[12]}
    out.update({k: mean[i] for i, k in enumerate(_MEAN_KEYS)})
    return out


def _load_for_eval(model_path, data_path):
    from ..modalmodule import get_model
    from ..utils import load_checkpoint
    checkpoint = load_checkpoint(model_path)
    _config = checkpoint['_config']
    print('Best model: Epoch = {}, Eiters = {}, Rsum = {:.2f}, R1 = {:.2f}'.format(
        checkpoint['epoch'], checkpoint['Eiters'], checkpoint['best_rsum'], checkpoint.get('best_r1', checkpoint.get('best_rl', 0.0))))
    if data_path is not None:
        _config['data_path'] = data_path
    model = get_model(_config)
    model.load_state_dict(checkpoint['model'])
    return model, _config


def _score_blocks(models_embs, fold5):
    """models_embs: list of (model, img_embs, cap_embs, cap_lens, shard_size); the similarity matrices of the
    models are averaged (ensemble, evaluation.py:377-381)."""
    def sims_of(sl_img, sl_cap):
        acc = None
        for model, img, cap, lens, shard in models_embs:
            s = cal_sims(model, img[sl_img], cap[sl_cap], lengths=lens[sl_cap], shard_size=shard)
            acc = s if acc is None else acc + s
        return acc / len(models_embs)

    n = len(models_embs[0][1])
    if not fold5:
        return cal_recall(sims_of(slice(0, n, 5), slice(None)))
    res_dic = {'sum_result': []}
    for i in range(5):
        print(f"--------------------- The {i + 1} part ---------------------")
        part = cal_recall(sims_of(slice(i * 5000, (i + 1) * 5000, 5), slice(i * 5000, (i + 1) * 5000)))
        res_dic[f'PART_{i + 1}'] = part
        res_dic['sum_result'] += part['result']
    res_dic['Mean_metrics'] = _mean_metrics(res_dic)
    return res_dic


def _evalrank(model_paths, data_path, split, fold5, tag):
    import os
    import yaml
    from ..datamodule import data_loader as data
    loaded = [_load_for_eval(p, data_path) for p in model_paths]
    _config = loaded[0][1]
    print(f'Loading dataset : {_config["data_name"]} ......')
    data_loader, _ = data.get_test_loader(split, _config['data_name'], _config['batch_size'], _config['workers'], _config)
    print('Computing results...')
    # the reference asks for max-length sizing only for SGRAF (evaluation.py:284), which breaks SCAN whenever a later
    # batch holds a longer caption than the first; word-level models all need it
    islength = _config['name'] in ['SGRAF', 'SCAN']
    embs = []
    for model, cfg in loaded:
        img, cap, lens = encode_data(model, data_loader, islength=islength)
        embs.append((model, img, cap, lens, cfg['batch_size'] * 5))
    print('#Images: %d, #Captions: %d' % (embs[0][1].shape[0] / 5, embs[0][2].shape[0]))
    res_dic = _score_blocks(embs, fold5)
    res_dic['data_name'] = _config['data_name'] + ('_5fold' if fold5 else '')
    if len(model_paths) > 1 and fold5:
        res_dic['modal_path_1'], res_dic['modal_path_2'] = model_paths[0], model_paths[1]
    save_dir = os.path.dirname(model_paths[0])
    out_file = os.path.join(save_dir, f'{res_dic["data_name"]}_{tag}_result.yaml')
    with open(out_file, 'w') as yaml_file:
        yaml.safe_dump(_plain(res_dic), yaml_file)
    return res_dic


def _recall_dict(ranks):
    """cal_recall's dict (evaluation.py:225-259) from rank vectors."""
    i_rank, i_top, t_rank, t_top = [np.asarray(x, dtype=np.float64) for x in ranks]
    r, ri = ops.recall_from_ranks(i_rank), ops.recall_from_ranks(t_rank)
    ar, ari = (r[0] + r[1] + r[2]) / 3, (ri[0] + ri[1] + ri[2]) / 3
    rsum = r[0] + r[1] + r[2] + ri[0] + ri[1] + ri[2]
    return {'result': [list(r) + list(ri) + [ar, ari, rsum]], 'rsum': rsum, 'i2t_ave_r': ar, 'i2t_r1': r[0], 'i2t_r5': r[1],
            'i2t_r10': r[2], 'i2t_medr': r[3], 'i2t_meanr': r[4], 'i2t_ranks': i_rank, 'i2t_top1': i_top, 't2i_ave_r': ari,
            't2i_r1': ri[0], 't2i_r5': ri[1], 't2i_r10': ri[2], 't2i_medr': ri[3], 't2i_meanr': ri[4], 't2i_ranks': t_rank,
            't2i_top1': t_top}


def evalrank_fast(model_path, data_path=None, split='dev', fold5=False, comm=None):
    """Same result dict / YAML as evalrank_single through the sharded device-resident pipeline
    (itr_amd.evalpipe.evaluate_precomp): launch one process per GPU with torch.distributed.run; rank 0 writes
    `<run dir>/<data_name>[_5fold]_single_result.yaml`."""
    import os
    import yaml
    from .. import evalpipe
    from ..datamodule import data_loader as data
    model, _config = _load_for_eval(model_path, data_path)
    dset = data.PrecompDataset(os.path.join(_config['data_path'], _config['data_name']), split, _config)
    comm = comm or evalpipe.Comm()
    if not fold5:
        res_dic = _recall_dict(evalpipe.evaluate_precomp(model, dset, comm))
    else:
        res_dic = {'sum_result': []}
        for i in range(5):
            part = _recall_dict(evalpipe.evaluate_precomp(model, dset, comm, fold=(i, 5000)))
            res_dic[f'PART_{i + 1}'] = part
            res_dic['sum_result'] += part['result']
        res_dic['Mean_metrics'] = _mean_metrics(res_dic)
    res_dic['data_name'] = _config['data_name'] + ('_5fold' if fold5 else '')
    if comm.rank == 0:
        with open(os.path.join(os.path.dirname(model_path), f'{res_dic["data_name"]}_single_result.yaml'), 'w') as f:
            yaml.safe_dump(_plain(res_dic), f)
    return res_dic


def evalrank_single(model_path, data_path=None, split='dev', fold5=False):
    """evaluation.py:262-335."""
    return _evalrank([model_path], data_path, split, fold5, 'single')


def evalrank_ensemble(model_path, model_path2, data_path=None, split='dev', fold5=False):
    """evaluation.py:338-435: the two models' similarity matrices are averaged before ranking."""
    return _evalrank([model_path, model_path2], data_path, split, fold5, 'ensemble')
