"""Training / validation glue with the reference's names (itr/utils.py:17-186).  tensorboard_logger is optional."""
import logging
import os
import random
import time

import numpy
import torch

from .metricmodule import evaluation as eval
from .metricmodule import second2DHM
from .config import load_hyperparams

try:                                   # absent in this image; the reference logs scalars through it (utils.py:8)
    import tensorboard_logger as tb_logger
except ImportError:                    # pragma: no cover
    class _NoTB(object):
        def configure(self, *a, **k):
            pass

        def log_value(self, *a, **k):
            pass
    tb_logger = _NoTB()


def setup_seed(seed):
    """utils.py:17-22."""
    torch.manual_seed(seed)
    torch.cuda.manual_seed_all(seed)
    numpy.random.seed(seed)
    random.seed(seed)


def adjust_learning_rate(_config, optimizer, epoch):
    """LR decayed by 10 every `lr_update` epochs (utils.py:25-32)."""
    lr = _config['learning_rate'] * (0.1 ** (epoch // _config['lr_update']))
    for param_group in optimizer.param_groups:
        param_group['lr'] = lr


def load_checkpoint(path):
    """Checkpoints are pickled dicts {'epoch', 'model': [state_dicts], 'best_rsum', 'best_r1', '_config', 'Eiters'}
    (utils.py:135-142, Models.py:37-45)."""
    return torch.load(path, map_location='cpu', weights_only=False)


def load_resume(models, _config, reload=False):
    """utils.py:35-55."""
    if not os.path.exists(_config['resume']):
        raise FileNotFoundError("=> no checkpoint is found at '{}'".format(_config['resume']))
    print("=> loading checkpoint '{}'".format(_config['resume']))
    checkpoint = load_checkpoint(_config['resume'])
    start_epoch, best_rsum, best_r1 = checkpoint['epoch'], checkpoint['best_rsum'], checkpoint.get('best_r1', checkpoint.get('best_rl', 0))
    if reload:
        for name in load_hyperparams:
            _config[name] = checkpoint['_config'][name]
    model = models.get_model(_config).cuda()
    model.load_state_dict(checkpoint['model'])
    model.Eiters = checkpoint['Eiters']
    print("=> loaded checkpoint '{}' (epoch {}, best_rsum {}, best_rl {})".format(_config['resume'], start_epoch, best_rsum, best_r1))
    return model, start_epoch, best_rsum, best_r1


def is_main_process():
    """Rank 0 of a data-parallel run (or the only process): the one that logs and writes checkpoints."""
    import torch.distributed as dist
    return not (dist.is_available() and dist.is_initialized()) or dist.get_rank() == 0


def save_checkpoint(state, is_best, filename='checkpoint.pth.tar', prefix='', is_epo_end=False):
    """utils.py:58-62.  Data parallel: the replicas are identical, rank 0 writes."""
    if not is_main_process():
        return
    if is_epo_end:
        torch.save(state, os.path.join(prefix, 'epo' + str(state['epoch']) + '_' + filename))
    if is_best:
        torch.save(state, os.path.join(prefix, 'model_best.pth.tar'))


def print_options(config, width=120, per_line=3):
    """The option table the reference prints at start-up (utils.py:64-75): sorted `key: value` cells, three to a 120-column line, under a
    centred header rule.  Same text; composed from rows rather than accumulated cell by cell."""
    cells = ["%s: %s" % kv for kv in sorted(config.items())]
    cell_w = width // per_line
    rows = ["".join(c.center(cell_w) for c in cells[i:i + per_line]) for i in range(0, len(cells), per_line)]
    print("\n".join(["", "----- options -----".center(width, "-")] + rows + ["-" * width, ""]))


def train_step(_config, train_loader, model, epoch, val_loader, best_rsum=0, best_r1=0):
    """One epoch (utils.py:79-141)."""
    batch_time, data_time = eval.AverageMeter(), eval.AverageMeter()
    train_logger = eval.LogCollector()
    end = time.time()
    model.train_start()
    for i, train_data in enumerate(train_loader):
        data_time.update(time.time() - end, n=1)
        model.logger = train_logger
        model.train_emb(train_data)
        batch_time.update(time.time() - end, n=1)
        end = time.time()
        if model.Eiters % _config['log_step'] == 0:
            logging.info('Epoch: [{0}][{1}/{2}]\t{e_log}\tTime {batch_time.avg:.3f} ({bs})\tData {data_time.avg:.3f} ({ds})\t'
                         .format(epoch, i, len(train_loader), batch_time=batch_time, data_time=data_time, e_log=str(model.logger),
                                 bs=second2DHM(batch_time.sum), ds=second2DHM(data_time.sum)))
        tb_logger.log_value('epoch', epoch, step=model.Eiters)
        tb_logger.log_value('step', i, step=model.Eiters)
        tb_logger.log_value('batch_time', batch_time.val, step=model.Eiters)
        tb_logger.log_value('data_time', data_time.val, step=model.Eiters)
        model.logger.tb_log(tb_logger, step=model.Eiters)
        if model.Eiters % _config['val_step'] == 0:
            rsum, r1 = validate_step(_config, val_loader, model)
            is_best = rsum > best_rsum
            best_rsum, best_r1 = max(rsum, best_rsum), max(r1, best_r1)
            save_checkpoint({'epoch': epoch, 'model': model.state_dict(), 'best_rsum': best_rsum, 'best_r1': best_r1,
                             '_config': _config, 'Eiters': model.Eiters}, is_best, prefix=_config['save_dir'])
            model.train_start()
    return best_rsum, best_r1


def validate_step(_config, val_loader, model, fast=None):
    """Encode the validation split, score, rank -> (r_sum, r1) (utils.py:144-186).
    fast (default: _config.get('fast_eval', True) when the loader wraps a PrecompDataset): the device-resident pipeline
    (evalpipe.evaluate_precomp: every unique image encoded once, packed captions, fused scoring) instead of
    encode_data + cal_sims; the rank vectors are identical (tests/test_evalrank_gpu.py)."""
    from .datamodule.data_loader import PrecompDataset
    start = time.time()
    model.val_start()
    dset = getattr(val_loader, 'dataset', None)
    if fast is None:
        fast = bool(_config.get('fast_eval', True)) and isinstance(dset, PrecompDataset) and dset.im_div == 5 and len(dset) % 5 == 0
    if fast:
        from . import evalpipe
        i_rank, _, t_rank, _ = evalpipe.evaluate_precomp(model, dset)
        print("Calculate similarity time:", time.time() - start)
        (r1, r5, r10, medr, meanr) = eval.ops.recall_from_ranks(i_rank)
        (r1i, r5i, r10i, medri, meanri) = eval.ops.recall_from_ranks(t_rank)
    else:
        islength = _config['name'] in ['SGRAF', 'SCAN']
        img_embs, cap_embs, cap_lens = eval.encode_data(model, val_loader, islength=islength)
        img_embs = numpy.array([img_embs[i] for i in range(0, len(img_embs), 5)])   # 5 duplicated image rows -> 1
        sims = eval.cal_sims(model, img_embs, cap_embs, lengths=cap_lens, shard_size=100)
        print("Calculate similarity time:", time.time() - start)
        (r1, r5, r10, medr, meanr) = eval.i2t(sims)
        (r1i, r5i, r10i, medri, meanri) = eval.t2i(sims)
    logging.info("Image to text: r1 %.1f; r5 %.1f; r10 %.1f; medr %.1f; meanr %.1f" % (r1, r5, r10, medr, meanr))
    logging.info("Text to image: r1 %.1f; r5 %.1f; r10 %.1f; medr %.1f; meanr %.1f" % (r1i, r5i, r10i, medri, meanri))
    r_sum = r1 + r5 + r10 + r1i + r5i + r10i
    for k, v in (('r1_i2t', r1), ('r5_i2t', r5), ('r10_i2t', r10), ('medr_i2t', medr), ('meanr_i2t', meanr), ('r1_t2i', r1i),
                 ('r5_t2i', r5i), ('r10_t2i', r10i), ('medr_t2i', medri), ('meanr_t2i', meanri), ('r_sum', r_sum)):
        tb_logger.log_value(k, v, step=model.Eiters)
    return r_sum, r1
