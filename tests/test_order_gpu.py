"""GPU: measure='order' -- order_sim, SAEM's pdist, the hinge on top and the training gradient -- against G17 (captured
from the reference's Objectives.order_sim / pdist / ContrastiveLoss) and the oracle at tile-crossing shapes."""
import numpy as np
import pytest
import torch

import itr_oracle as O
from itr_amd import autograd as ag, ops
from itr_amd.modalmodule import Objectives

pytestmark = pytest.mark.gpu


def T(a):
    return torch.from_numpy(np.asarray(a))


def test_order_and_pdist_golden(golden, dev):
    g = golden("g17_order")
    im, s = T(g["im"]).to(dev), T(g["s"]).to(dev)
    assert np.abs(Objectives.order_sim(im, s).cpu().numpy() - g["order"]).max() <= 1e-6
    assert np.abs(Objectives.pdist(im, s).cpu().numpy() - g["pdist"]).max() <= 1e-6
    for mv, tag in ((False, 'sum'), (True, 'maxviol')):
        crit = Objectives.ContrastiveLoss(config={'name': 'VSE++'}, margin=0.2, measure='order', max_violation=mv)
        assert float(crit(im[:8], s[:8])) == pytest.approx(float(g["loss_" + tag]), abs=1e-5)
        a = im[:8].clone().requires_grad_(True)
        b = s[:8].clone().requires_grad_(True)
        loss = ops.hinge_loss(ag.order_scores(a, b), 0.2, mv)
        loss.backward()
        assert float(loss) == pytest.approx(float(g["loss_" + tag]), abs=1e-5)
        assert np.abs(a.grad.cpu().numpy() - g["d_im_" + tag]).max() <= 2e-6
        assert np.abs(b.grad.cpu().numpy() - g["d_s_" + tag]).max() <= 2e-6
    saem = Objectives.ContrastiveLoss(config={'name': 'SAEM'}, margin=0.2, measure='order', max_violation=True)
    assert float(saem(im[:8], s[:8])) == pytest.approx(float(g["saem_order_loss"]), abs=1e-5)
    with pytest.raises(ValueError):
        Objectives.ContrastiveLoss(config={'name': 'VSE++'}, margin=0.2, measure='euclid')


@pytest.mark.parametrize("Ni,Nc,D", [(1, 1, 4), (70, 131, 100), (130, 64, 1024), (5, 300, 36)])
def test_order_scores_vs_oracle(dev, Ni, Nc, D):
    torch.manual_seed(Ni + Nc)
    im, s = torch.randn(Ni, D), torch.randn(Nc, D)
    want = O.order_sim(im.double(), s.double())
    got = ops.order_scores(im.to(dev), s.to(dev)).cpu()
    assert (got.double() - want).abs().max().item() <= 1e-5 * max(1.0, want.abs().max().item())
    wantp = O.pdist(im.double(), s.double())
    gotp = ops.pdist(im.to(dev), s.to(dev)).cpu()
    assert (gotp.double() - wantp).abs().max().item() <= 2e-5 * max(1.0, wantp.abs().max().item())


def test_order_backward_vs_autograd(dev):
    torch.manual_seed(2)
    im, s = torch.randn(33, 72).abs(), torch.randn(21, 72).abs()
    dS = torch.randn(33, 21)
    a, b = im.double().requires_grad_(True), s.double().requires_grad_(True)
    (O.order_sim(a, b) * dS.double()).sum().backward()
    ga = im.to(dev).requires_grad_(True)
    gb = s.to(dev).requires_grad_(True)
    (ag.order_scores(ga, gb) * dS.to(dev)).sum().backward()
    assert (ga.grad.cpu().double() - a.grad).abs().max().item() <= 2e-5
    assert (gb.grad.cpu().double() - b.grad).abs().max().item() <= 2e-5
    # a pair without any violated dimension (s <= im everywhere): score 0, no gradient (torch would give NaN)
    z = ag.order_scores(torch.ones(1, 8, device=dev, requires_grad=True), torch.zeros(1, 8, device=dev))
    assert float(z) == 0.0


def test_order_empty(dev):
    assert ops.order_scores(torch.zeros(0, 8, device=dev), torch.zeros(3, 8, device=dev)).shape == (0, 3)
    with pytest.raises(ValueError):
        ops.order_scores(torch.zeros(2, 8, device=dev), torch.zeros(3, 12, device=dev))
