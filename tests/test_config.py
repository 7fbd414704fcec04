"""CPU: the sacred-compatible config surface (itr/config.py) -- key names, defaults, named-config layering."""
import pytest

from itr_amd import config as C


def test_defaults_and_named_layering():
    cfg = C.build_config(['with', 'SCAN', 'data_name=coco_precomp', 'max_violation=True', 'bi_gru=True',
                          'agg_func=LogSumExp', 'cross_attn=t2i', 'lambda_lse=6', 'lambda_softmax=9'])
    assert cfg['name'] == 'SCAN' and cfg['data_name'] == 'coco_precomp'
    assert cfg['bi_gru'] is True and cfg['max_violation'] is True
    assert cfg['img_dim'] == 2048 and cfg['no_txtnorm'] is True and cfg['embed_size'] == 1024
    assert cfg['margin'] == 0.2 and cfg['word_dim'] == 300 and cfg['batch_size'] == 128
    assert cfg['lambda_lse'] == 6 and cfg['lambda_softmax'] == 9
    # <save_path>/<name>/<dataset>_<seed>_<%Y-%m-%d-%H-%M-%S> like the reference's hook (config.py:391-393)
    import re
    assert re.search(r'/SCAN/coco_0_\d{4}-\d{2}-\d{2}-\d{2}-\d{2}-\d{2}$', cfg['save_dir'])


def test_hook_writes_hparams(tmp_path):
    import yaml
    cfg = C.build_config(['with', 'SCAN', 'save_path=%s' % tmp_path, 'tail=_x', 'seed=7'])
    cfg = C.config_hook(cfg, make_dirs=True)
    assert cfg['save_dir'].endswith('_x')
    hp = yaml.safe_load(open(cfg['save_dir'] + '/hparams.yaml'))
    assert hp['name'] == 'SCAN' and hp['seed'] == 7 and hp['learning_rate'] == cfg['learning_rate']


def test_other_named_configs():
    assert C.build_config(['with', 'SGRAF', 'module_name=SGR'])['module_name'] == 'SGR'
    assert C.build_config(['with', 'SGRAF'])['bi_gru'] is True
    sa = C.build_config(['with', 'SAEM', 'bert_path=/x'])
    assert sa['batch_size'] == 64 and sa['final_dims'] == 256 and sa['bert_config_file'] == '/x/bert_config.json'
    cam = C.build_config(['with', 'CAMERA'])
    assert cam['embed_size'] == 2048 and cam['smry_k'] == 12 and cam['head'] == 64
    v = C.build_config(['with', 'VSE_PP'])
    assert v['name'] == 'VSE++' and '/VSE_PP/' in v['save_dir']


def test_unknown_keys_raise():
    with pytest.raises(KeyError):
        C.build_config(['with', 'NOPE'])
    with pytest.raises(KeyError):
        C.build_config(['with', 'SCAN', 'not_a_key=1'])


def test_every_load_hyperparam_has_a_default():
    for k in C.load_hyperparams:
        assert k in C.DEFAULTS
