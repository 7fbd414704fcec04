"""`train.py with <NAMED_CONFIG> key=value ...` configuration surface of the reference (itr/config.py) without
sacred (not installable here): same keys, same defaults (config.py:20-106), same named-config overrides
(:109-378) and the same hook semantics (:381-414), so README command lines keep working.

Only the keys are the contract -- experiment tracking (sacred observers, tensorboard) is out of scope."""
import ast
import os
import random
import time

load_hyperparams = ['img_encoder', 'crop_size', 'img_dim', 'no_imgnorm', 'use_bbox', 'finetune', 'precomp_enc_type',
                    'trans_cfg', 'head', 'text_encoder', 'bi_gru', 'word_dim', 'no_txtnorm', 'num_layers', 'max_words',
                    'txt_stru', 'embed_size', 'measure', 'use_abs', 'final_dims', 'sim_dim', 'rnn_type',
                    'bidirectional', 'dim_hidden', 'dim_vid', 'input_dropout_p', 'rnn_dropout_p', 'dim_word', 'max_len',
                    'module_name', 'sgr_step', 'max_violation', 'margin', 'cross_attn', 'raw_feature_norm', 'agg_func',
                    'lambda_lse', 'lambda_softmax', 'smry_k', 'smry_lamda', 'lr_decay_gamma', 'drop']

DEFAULTS = dict(
    name='ITR', data_path="/workspace/dataset/data", data_name="f30k_precomp", vocab_path="./itr/vocab",
    vocab_type='json', save_path="./runs", tail=None, seed=0, cuda="2", workers=8, resume=None, num_epochs=30,
    batch_size=128, learning_rate=.0002, lr_update=15, val_step=500, log_step=10, grad_clip=2., use_restval=False,
    img_encoder='vgg19', crop_size=224, img_dim=4096, no_imgnorm=False, use_bbox=False, finetune=False,
    precomp_enc_type="basic", trans_cfg='./itr/trans_cfg.json', head=64, text_encoder='gru', bi_gru=False,
    word_dim=300, no_txtnorm=False, num_layers=1, bert_path='/workspace/dataset/uncased_L-12_H-768_A-12',
    max_words=32, txt_stru='cnn', embed_size=1024, measure='cosine', use_abs=False, final_dims=256, sim_dim=256,
    rnn_type='gru', bidirectional=0, dim_hidden=512, dim_vid=2048, input_dropout_p=0.2, rnn_dropout_p=0.5,
    dim_word=300, max_len=60, module_name='SGR', sgr_step=3, max_violation=False, margin=0.2, cross_attn="t2i",
    raw_feature_norm="clipped_l2norm", agg_func="LogSumExp", lambda_lse=6, lambda_softmax=9., smry_k=12,
    smry_lamda=0.01, lr_decay_gamma=0.1, drop=0.0)

NAMED = {
    'VSE_PP': dict(name="VSE++", data_name="f30k_precomp", vocab_type='pkl', val_step=10, img_dim=4096,
                   no_txtnorm=True),
    'SCAN': dict(name="SCAN", img_dim=2048, no_txtnorm=True, bi_gru=False, cross_attn="t2i",
                 raw_feature_norm="clipped_l2norm", agg_func="LogSumExp", lambda_lse=6, lambda_softmax=9.),
    'VSRN': dict(name="VSRN", img_dim=2048, embed_size=2048, lr_update=15, bidirectional=False),
    'SAEM': dict(name="SAEM", batch_size=64, learning_rate=.0001, lr_update=10, val_step=1000, img_dim=2048,
                 text_encoder='bert', max_words=32, txt_stru='cnn', embed_size=1024, final_dims=256),
    'SGRAF': dict(name="SGRAF", module_name='SAF', sgr_step=3, num_epochs=40, lr_update=30, val_step=1000,
                  img_dim=2048, bi_gru=True, no_txtnorm=False, embed_size=1024, sim_dim=256),
    'CAMERA': dict(name="CAMERA", num_epochs=1, learning_rate=.0001, lr_update=10, img_dim=2048, use_bbox=True,
                   head=64, text_encoder='bert', max_words=32, embed_size=2048, smry_k=12, smry_lamda=0.01,
                   lr_decay_gamma=0.1, drop=0.0),
}


def _parse_value(text):
    try:
        return ast.literal_eval(text)
    except (ValueError, SyntaxError):
        return text


def build_config(argv):
    """argv as after the script name: ['with', 'SCAN', 'data_name=coco_precomp', 'bi_gru=True'].
    Layering like sacred: defaults < named configs (in order) < key=value updates, then the hook."""
    cfg = dict(DEFAULTS)
    args = list(argv)
    if args and args[0] == 'with':
        args = args[1:]
    updates = {}
    for a in args:
        if '=' in a:
            k, v = a.split('=', 1)
            if k not in DEFAULTS:
                raise KeyError("unknown config key %r" % k)
            updates[k] = _parse_value(v)
        else:
            if a not in NAMED:
                raise KeyError("unknown named config %r (known: %s)" % (a, ', '.join(NAMED)))
            cfg.update(NAMED[a])
    cfg.update(updates)
    return config_hook(cfg)


def config_hook(cfg, make_dirs=False):
    """config.py:381-414: seed, name normalisation, `<save_path>/<name>/<dataset>_<seed>_<timestamp>` run directory,
    BERT file names, `hparams.yaml` (written when the directory is made)."""
    cfg = dict(cfg)
    if cfg['seed'] is None:
        cfg['seed'] = random.randint(0, 10000)
    display = 'VSE_PP' if cfg['name'] == 'VSE++' else cfg['name']   # the reference renames the model itself; get_model here accepts both
    stamp = '_'.join([cfg['data_name'].split('_')[0], str(cfg['seed']), time.strftime('%Y-%m-%d-%H-%M-%S', time.localtime())])
    # the reference creates `<dir><tail>` but records `<dir>` (and then fails to write hparams.yaml); the tail is kept here
    cfg['save_dir'] = os.path.join(cfg['save_path'], display, stamp + (cfg['tail'] if cfg['tail'] else ''))
    if cfg['text_encoder'] == 'bert':
        cfg['vocab_file'] = os.path.join(cfg['bert_path'], 'vocab.txt')
        cfg['bert_config_file'] = os.path.join(cfg['bert_path'], 'bert_config.json')
        cfg['init_checkpoint'] = os.path.join(cfg['bert_path'], 'pytorch_model.bin')
    if make_dirs:
        import yaml
        os.makedirs(cfg['save_dir'], exist_ok=True)
        with open(os.path.join(cfg['save_dir'], 'hparams.yaml'), 'w') as yaml_file:
            yaml.safe_dump({k: v for k, v in cfg.items() if isinstance(v, (str, int, float, bool, type(None), list, dict))}, yaml_file)
    # one process per GPU: the device comes from LOCAL_RANK, `cuda` is kept for compatibility only
    return cfg
