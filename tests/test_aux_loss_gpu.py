"""GPU: the training-only auxiliary losses (csrc/aux_loss.hip) against a float64 torch restatement of the
reference's formulas (Objectives.py:238-290, :521-542): values and gradients.  The reference's own numbers pin them through G13
(values) and G18 / G19 (SAEM / CAMERA training steps)."""
import pytest
import torch

from itr_amd import autograd as ag
from itr_amd.modalmodule.Objectives import AngularLoss, DiversityRegularization

pytestmark = pytest.mark.gpu


def _angular64(anchors, positives, others, ab, max_violation):
    n = anchors.shape[0]
    idx = torch.tensor([[j for j in range(n) if j != i] for i in range(n)])
    neg = others[idx]
    a, p = anchors.unsqueeze(1), positives.unsqueeze(1)
    x = 4. * ab * torch.matmul(a + p, neg.transpose(1, 2)) - 2. * (1. + ab) * torch.matmul(a, p.transpose(1, 2))
    if max_violation:
        return torch.log(1 + torch.exp(x.max(2)[0])).sum()
    t = torch.max(x, dim=2)[0].detach()
    x = torch.exp(x - t.unsqueeze(dim=1))
    return torch.mean(t + torch.log(torch.exp(-t) + torch.sum(x, 2)))


@pytest.mark.parametrize("n,D", [(7, 24), (70, 40), (130, 16)])
@pytest.mark.parametrize("max_violation", [True, False])
def test_angular_loss(dev, n, D, max_violation):
    torch.manual_seed(n)
    im = torch.nn.functional.normalize(torch.randn(n, D), dim=1)
    s = torch.nn.functional.normalize(torch.randn(n, D), dim=1)
    I, S = im.double().requires_grad_(True), s.double().requires_grad_(True)
    want = _angular64(I, S, S, 1.0, max_violation) + _angular64(S, I, I, 1.0, max_violation)
    want.backward()
    gi, gs = im.to(dev).requires_grad_(True), s.to(dev).requires_grad_(True)
    crit = AngularLoss(max_violation=max_violation)
    got = crit(gi, gs)
    (got * 1.5).backward()
    assert float(got) == pytest.approx(float(want), rel=2e-6)
    assert float((gi.grad.cpu().double() / 1.5 - I.grad).abs().max()) <= 2e-5 * max(1.0, float(I.grad.abs().max()))
    assert float((gs.grad.cpu().double() / 1.5 - S.grad).abs().max()) <= 2e-5 * max(1.0, float(S.grad.abs().max()))


def test_angular_loss_rejects_one_row(dev):
    x = torch.randn(1, 8, device=dev)
    with pytest.raises(ValueError):
        ag.angular_loss(x, x, x)


@pytest.mark.parametrize("B,R,K", [(5, 36, 12), (3, 36, 1), (2, 50, 33)])
def test_diversity_regularization(dev, B, R, K):
    torch.manual_seed(B)
    sm = torch.randn(B, R, K)
    sm[0, :, 0] = 0.0                                     # a zero column: F.normalize's eps clamp, gradient of that column 0
    X = sm.double().requires_grad_(True)
    sn = torch.nn.functional.normalize(X, dim=1)
    want = ((torch.matmul(sn.transpose(1, 2), sn) - torch.eye(K, dtype=torch.float64).unsqueeze(0)) ** 2).sum()
    want.backward()
    gx = sm.to(dev).requires_grad_(True)
    got = DiversityRegularization(K, B)(gx)
    (got * 0.5).backward()
    assert float(got) == pytest.approx(float(want), rel=5e-6)
    assert float((gx.grad.cpu().double() * 2 - X.grad).abs().max()) <= 1e-5 * max(1.0, float(X.grad.abs().max()))
    assert float(DiversityRegularization(K, 0)(torch.zeros(0, R, K, device=dev))) == 0.0
