"""GPU: VSRN.train_emb against G21 -- the reference's own VSRN.train_emb run twice on CPU with its dropout modules at
p = 0: captioning and retrieval losses of both steps, every (clipped) gradient after step 1 (image tower with its GCN BatchNorms, text
tower, encoder / attention decoder of the captioning model), parameters and running statistics after the second Adam step."""
import numpy as np
import pytest
import torch

from itr_amd import config as C
from itr_amd.metricmodule.evaluation import LogCollector
from itr_amd.modalmodule import get_model

pytestmark = pytest.mark.gpu


def T(a):
    return torch.from_numpy(np.asarray(a))


def test_vsrn_train_emb_matches_reference(golden, dev):
    g = golden("g21_vsrn_train")
    cfg = C.build_config(['with', 'VSRN', 'data_name=coco_precomp', 'max_violation=True', 'learning_rate=0.002'])
    cfg.update(img_dim=20, embed_size=32, word_dim=12, vocab_size=40, dim_vid=32, dim_hidden=16, dim_word=10, max_len=8, input_dropout_p=0.0,
               rnn_dropout_p=0.0)
    model = get_model(cfg)
    mods = (('img', model.img_enc), ('txt', model.txt_enc), ('cap', model.caption_model))
    for which, m in mods:
        pre = 'w0_%s_' % which
        m.load_state_dict({k[len(pre):]: T(g[k]) for k in g.files if k.startswith(pre)})
    model.cuda() if hasattr(model, 'cuda') else None
    model.caption_model.cuda()
    model.train_start()
    model.logger = LogCollector()
    lr = 2e-3
    for step in (1, 2):
        pre = 's%d_' % step
        lens = [int(x) for x in g[pre + 'lens']]
        model.train_emb((T(g[pre + 'feats']), None, None, T(g[pre + 'ids']), lens, list(range(len(lens))), T(g[pre + 'mask']), None))
        assert float(model.logger.meters['Loss_caption'].val) == pytest.approx(float(g[pre + 'loss_caption']), abs=1e-4)
        assert float(model.logger.meters['Loss_retrieval'].val) == pytest.approx(float(g[pre + 'loss_retrieval']), abs=3e-5)
        if step == 1:
            gn = float(model.optimizer.last_grad_norm[0])
            coef = min(1.0, model.grad_clip / (gn + 1e-6))
            worst, n_checked = [], 0
            for which, m in mods:
                for n, p in m.named_parameters():
                    key = pre + 'grad_%s.%s' % (which, n)
                    if key not in g.files:
                        assert p.grad is None or float(p.grad.abs().max()) == 0.0, n
                        continue
                    want = T(g[key])
                    rel = float((p.grad.detach().cpu() * coef - want).norm() / (want.norm() + 1e-12))
                    # (a convolution bias directly in front of a BatchNorm has a vanishing true gradient: what both runs hold is noise)
                    if float(want.abs().max()) > 1e-5 and not n.endswith('W.0.bias'):
                        worst.append((rel, which + '.' + n))
                    n_checked += 1
            worst.sort(reverse=True)
            assert worst[0][0] <= 3e-4, worst[:5]
            assert n_checked >= 40
    for which, m in mods:
        for k, v in m.state_dict().items():
            want = T(g['s2_%s_%s' % (which, k)])
            if not v.is_floating_point():
                assert int(v) == int(want), k
                continue
            d = (v.cpu().float() - want.float()).abs()
            if 'running_' in k:
                assert float(d.max()) <= 5e-4 * max(1.0, float(want.abs().max())), (k, float(d.max()))
                continue
            gkey = 's1_grad_%s.%s' % (which, k)
            if k.endswith('W.0.bias') or (gkey in g.files and float(np.abs(g[gkey]).max()) < 1e-5):   # vanishing true gradient: Adam random-walks both runs
                assert float(d.max()) <= 4 * lr + 1e-7, (k, float(d.max()))
            else:
                assert float(d.max()) <= 2 * lr + 1e-7 and float(d.mean()) <= 2e-4, (k, float(d.max()), float(d.mean()))


@pytest.mark.gpu
@pytest.mark.parametrize("B,N,H", [(128, 36, 512), (3, 5, 300), (2, 1, 7), (70, 36, 2048)])
def test_addattn_score_vs_float64_autograd(B, N, H):
    """itr_addattn_score(_bwd): e = linear2(tanh(enc_half + hidden_half)) (Attention.forward, Fusionmodule.py:136-141) in one pass, against
    torch float64 autograd -- the vector path (H % 4 == 0) and the scalar one, one region, the reference's 36 x 512."""
    from itr_amd import autograd as ag
    dev = torch.device("cuda:0")
    torch.manual_seed(B + H)
    x = torch.randn(B, N, H, device=dev, requires_grad=True)
    v = torch.randn(B, H, device=dev, requires_grad=True)
    w = (torch.randn(1, H, device=dev) * 0.1).requires_grad_()
    e = ag.addattn_score(x, v, w)
    g = torch.randn_like(e)
    e.backward(g)
    xd, vd, wd = (t.detach().double().requires_grad_() for t in (x, v, w))
    ed = (torch.tanh(xd + vd[:, None, :]) * wd.view(1, 1, H)).sum(-1)
    ed.backward(g.double())
    assert e.shape == (B, N)
    assert float((e.detach().double() - ed.detach()).abs().max()) <= 2e-5 * H ** 0.5
    assert float((x.grad.double() - xd.grad).abs().max()) <= 1e-5
    assert float((v.grad.double() - vd.grad).abs().max()) <= 1e-5 * N
    assert float((w.grad.double() - wd.grad).abs().max()) <= 2e-5 * (B * N) ** 0.5
