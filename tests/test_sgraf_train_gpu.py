"""GPU: SGRAF.train_emb (a14) against G20 -- the reference's own SGRAF.train_emb run twice on CPU for SAF and SGR with its dropout
modules switched to p = 0: the loss of both steps, every (clipped) gradient after step 1 (towers and similarity module), parameters
and BatchNorm running statistics after the second Adam step.  Plus a live-dropout run (p = 0.4 as hard-coded in the reference)."""
import numpy as np
import pytest
import torch

from itr_amd import config as C
from itr_amd.metricmodule.evaluation import LogCollector
from itr_amd.modalmodule import get_model

pytestmark = pytest.mark.gpu


def T(a):
    return torch.from_numpy(np.asarray(a))


def _model(g, mod, zero_dropout=True):
    cfg = C.build_config(['with', 'SGRAF', 'data_name=coco_precomp', 'module_name=%s' % mod, 'max_violation=True', 'learning_rate=0.002', 'bi_gru=True'])
    cfg.update(img_dim=24, embed_size=32, word_dim=16, vocab_size=60, sim_dim=16, sgr_step=3)
    model = get_model(cfg)
    pre = mod + '_w0_'
    sds = [{k[len(pre) + 4:]: T(g[k]) for k in g.files if k.startswith(pre + which + '_')} for which in ('img', 'txt', 'sim')]
    model.load_state_dict(sds)
    if zero_dropout:
        model.txt_enc.dropout_p = 0.0
        for m in model.sim_enc.modules():
            if isinstance(m, torch.nn.Dropout):
                m.p = 0.0
    model.train_start()
    model.logger = LogCollector()
    return model


def _batch(g, mod, step):
    pre = '%s_s%d_' % (mod, step)
    lens = [int(x) for x in g[pre + 'lens']]
    return (T(g[pre + 'feats']), None, None, T(g[pre + 'ids']), lens, list(range(len(lens))), None, None)


@pytest.mark.parametrize("mod", ["SAF", "SGR"])
def test_sgraf_train_emb_matches_reference(golden, dev, mod):
    g = golden("g20_sgraf_train")
    model = _model(g, mod)
    lr = 2e-3
    for step in (1, 2):
        model.train_emb(_batch(g, mod, step))
        pre = '%s_s%d_' % (mod, step)
        assert float(model.logger.meters['Loss'].val) == pytest.approx(float(g[pre + 'loss']), abs=3e-5)
        if step == 1:
            gn = float(model.optimizer.last_grad_norm[0])
            coef = min(1.0, model.grad_clip / (gn + 1e-6))         # the reference stores the gradients after clip_grad_norm_
            n_checked, worst = 0, []
            for which, m in (('img', model.img_enc), ('txt', model.txt_enc), ('sim', model.sim_enc)):
                for n, p in m.named_parameters():
                    key = pre + 'grad_%s.%s' % (which, n)
                    assert key in g.files, key
                    want = T(g[key])
                    rel = float((p.grad.detach().cpu() * coef - want).norm() / (want.norm() + 1e-12))
                    if float(want.abs().max()) > 1e-6:
                        worst.append((rel, which + '.' + n))
                    n_checked += 1
            worst.sort(reverse=True)
            assert worst[0][0] <= 2e-4, worst[:5]
            assert n_checked >= 25
    for which, m in (('img', model.img_enc), ('txt', model.txt_enc), ('sim', model.sim_enc)):
        for k, v in m.state_dict().items():
            want = T(g['%s_s2_%s_%s' % (mod, which, k)])
            if not v.is_floating_point():
                assert int(v) == int(want), k                      # num_batches_tracked (AttentionFiltration: once per caption)
                continue
            d = (v.cpu().float() - want.float()).abs()
            if 'running_' in k:
                assert float(d.max()) <= 5e-4 * max(1.0, float(want.abs().max())), (k, float(d.max()))
            else:
                # a bias in front of a BatchNorm has a vanishing true gradient (~1e-8 in the reference run): Adam turns that noise
                # into +-lr steps in BOTH runs, so such a parameter may differ by up to 2 steps x 2 lr
                gkey = '%s_s1_grad_%s.%s' % (mod, which, k)
                if gkey in g.files and float(np.abs(g[gkey]).max()) < 1e-5:
                    assert float(d.max()) <= 4 * lr + 1e-7, (k, float(d.max()))
                else:
                    assert float(d.max()) <= 2 * lr + 1e-7 and float(d.mean()) <= 2e-4, (k, float(d.max()), float(d.mean()))


def test_sgraf_train_emb_with_dropout(golden, dev):
    g = golden("g20_sgraf_train")
    runs = []
    for rep in range(2):
        torch.manual_seed(11)
        model = _model(g, 'SAF', zero_dropout=False)
        losses = []
        for step in (1, 2):
            model.train_emb(_batch(g, 'SAF', step))
            losses.append(float(model.logger.meters['Loss'].val))
        runs.append(losses)
    assert all(np.isfinite(runs[0])) and runs[0] == runs[1]
    assert abs(runs[0][0] - float(g['SAF_s1_loss'])) > 1e-4         # the 0.4 dropout sites are live


@pytest.mark.parametrize("mod", ["SAF", "SGR"])
def test_sgraf_step_through_the_module_seams(golden, dev, mod):
    """The reference composes its SGRAF step from the module seams (Models.py:507-546): forward_emb -> sim_enc(img, cap, lens)
    -> criterion(sims) -> backward.  In training mode those seams run on the autograd tape here too (one module, two modes)
    and give G20's first-step loss and gradients."""
    g = golden("g20_sgraf_train")
    model = _model(g, mod)
    images, _, _, ids, lens, _, _, _ = _batch(g, mod, 1)
    model.optimizer.zero_grad()
    img_emb, cap_emb = model.forward_emb(images, ids, lens)
    assert img_emb.requires_grad and cap_emb.requires_grad and cap_emb.dim() == 3
    sims = model.sim_enc(img_emb, cap_emb, lens)
    loss = model.criterion(sims)
    pre = '%s_s1_' % mod
    assert float(loss) == pytest.approx(float(g[pre + 'loss']), abs=3e-5)
    loss.backward()
    model.optimizer.step(max_norm=model.grad_clip)
    coef = min(1.0, model.grad_clip / (float(model.optimizer.last_grad_norm[0]) + 1e-6))
    worst = []
    for which, m in (('img', model.img_enc), ('txt', model.txt_enc), ('sim', model.sim_enc)):
        for n, p in m.named_parameters():
            want = T(g[pre + 'grad_%s.%s' % (which, n)])
            if float(want.abs().max()) > 1e-6:
                worst.append((float((p.grad.detach().cpu() * coef - want).norm() / (want.norm() + 1e-12)), which + '.' + n))
    assert max(worst)[0] <= 2e-4, sorted(worst, reverse=True)[:5]
