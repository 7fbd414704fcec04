#!/usr/bin/env python3
"""`python train.py with <MODEL> key=value ...` -- the reference's sacred command line (train.py:20-72, itr/config.py)
on the MI355X-native package: config -> precomp loaders -> model -> epochs of train_step / validate_step ->
checkpoints in the reference's layout.  One process per GPU (LOCAL_RANK selects the device); under
torch.distributed.run with N > 1 ranks the step is data parallel (modalmodule/Models.py) and validation row-sharded
(evalpipe.py); rank 0 alone logs and writes checkpoints."""
import copy
import logging
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from itr_amd import config as cfgmod, utils                # noqa: E402
from itr_amd import modalmodule as models                  # noqa: E402
from itr_amd.datamodule import data_loader as data         # noqa: E402


def train(_config):
    _config = copy.deepcopy(_config)
    utils.setup_seed(_config['seed'])
    logging.basicConfig(format='%(asctime)s %(message)s', level=logging.INFO if utils.is_main_process() else logging.WARNING)
    os.makedirs(_config['save_dir'], exist_ok=True)
    if utils.is_main_process():
        utils.tb_logger.configure(_config['save_dir'], flush_secs=5)
        utils.print_options(_config)
    train_loader, val_loader, vocab_size = data.get_loaders(_config['data_name'], _config['batch_size'], _config['workers'], _config)
    _config['vocab_size'] = vocab_size
    if _config['resume']:
        model, start_epoch, best_rsum, best_r1 = utils.load_resume(models, _config, reload=True)
        utils.validate_step(_config, val_loader, model)
    else:
        start_epoch, best_rsum, best_r1 = 0, 0, 0
        model = models.get_model(_config).cuda()
    for epoch in range(start_epoch, _config['num_epochs']):
        utils.adjust_learning_rate(_config, model.optimizer, epoch)
        best_rsum, best_r1 = utils.train_step(_config, train_loader, model, epoch, val_loader, best_rsum, best_r1)
        r_sum, r1 = utils.validate_step(_config, val_loader, model)   # (the reference unpacks these swapped)
        is_best = r_sum > best_rsum
        best_rsum, best_r1 = max(r_sum, best_rsum), max(r1, best_r1)
        utils.save_checkpoint({'epoch': epoch, 'model': model.state_dict(), 'best_rsum': best_rsum, 'best_r1': best_r1,
                               '_config': _config, 'Eiters': model.Eiters}, is_best, filename='checkpoint.pth.tar',
                              prefix=_config['save_dir'], is_epo_end=True)


    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


def main(argv):
    # one process per GPU; several ranks may share a device only under the gloo test backend
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1))
    cfg = cfgmod.build_config(argv)
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        # data-parallel training: `python -m torch.distributed.run --nproc-per-node N train.py with ...`.  Every rank
        # builds the same loaders (same seed => same global batches) and train_emb shards each batch; rank 0 makes the
        # run directory and tells the others its name and the seed (a random one when seed=None)
        dist.init_process_group(os.environ.get("ITR_DIST_BACKEND", "nccl"))
        box = [cfgmod.config_hook(cfg, make_dirs=True) if dist.get_rank() == 0 else None]
        dist.broadcast_object_list(box, src=0)
        cfg = box[0]
    else:
        cfg = cfgmod.config_hook(cfg, make_dirs=True)      # run directory + hparams.yaml
    train(cfg)


if __name__ == "__main__":
    main(sys.argv[1:])
