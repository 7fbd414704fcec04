"""GPU: training primitives of the CAMERA towers (csrc/train_camera.hip, itr_amd/autograd.py) against torch autograd on the CPU in
float64: elementwise product, activations, attention gate, training-mode BatchNorm1d (incl. running statistics), l2norm across
the regions, multi-view summarisation, multi-view matching."""
import numpy as np
import pytest
import torch

from itr_amd import autograd as ag

pytestmark = pytest.mark.gpu


def _cmp(got, want, tol):
    assert float((got.detach().cpu().double() - want.detach()).abs().max()) <= tol


def test_mul_act_gate(dev):
    torch.manual_seed(0)
    a, b, dy = torch.randn(50, 33), torch.randn(50, 33), torch.randn(50, 33)
    A, B_ = a.double().requires_grad_(True), b.double().requires_grad_(True)
    (A * B_).backward(dy.double())
    ga, gb = a.to(dev).requires_grad_(True), b.to(dev).requires_grad_(True)
    y = ag.mul(ga, gb)
    y.backward(dy.to(dev))
    _cmp(y, (A * B_), 1e-6); _cmp(ga.grad, A.grad, 1e-6); _cmp(gb.grad, B_.grad, 1e-6)
    for kind, fn in (('relu', torch.relu), ('tanh', torch.tanh), ('sigmoid', torch.sigmoid)):
        X = a.double().requires_grad_(True)
        fn(X).backward(dy.double())
        gx = a.to(dev).requires_grad_(True)
        out = ag.act(gx, kind)
        out.backward(dy.to(dev))
        _cmp(out, fn(X), 2e-6); _cmp(gx.grad, X.grad, 2e-6)
    rows, dk = 77, 32
    q, k, M = torch.randn(rows, dk), torch.randn(rows, dk), torch.rand(rows, 2 * dk)
    d1, d2 = torch.randn(rows, dk), torch.randn(rows, dk)
    Q, K, MM = (t.double().requires_grad_(True) for t in (q, k, M))
    ((Q * MM[:, :dk]) * d1.double()).sum().backward(retain_graph=True)
    ((K * MM[:, dk:]) * d2.double()).sum().backward()
    gq, gk, gm = (t.to(dev).requires_grad_(True) for t in (q, k, M))
    qo, ko = ag.gate_apply(gq, gk, gm)
    ((qo * d1.to(dev)).sum() + (ko * d2.to(dev)).sum()).backward()
    _cmp(qo, Q * MM[:, :dk], 1e-6); _cmp(ko, K * MM[:, dk:], 1e-6)
    _cmp(gq.grad, Q.grad, 1e-6); _cmp(gk.grad, K.grad, 1e-6); _cmp(gm.grad, MM.grad, 1e-6)


def test_batch_norm_train(dev):
    torch.manual_seed(1)
    for N, Cc in ((200, 70), (9, 130)):
        x, dy = torch.randn(N, Cc) * 2 + 0.5, torch.randn(N, Cc)
        ref = torch.nn.BatchNorm1d(Cc).double()
        ref.weight.data.uniform_(0.5, 1.5); ref.bias.data.normal_(0, 0.1)
        ref.running_mean.normal_(0, 0.1); ref.running_var.uniform_(0.5, 1.5)
        mine = torch.nn.BatchNorm1d(Cc)
        mine.load_state_dict({k: v.float() if v.is_floating_point() else v for k, v in ref.state_dict().items()})
        mine.to(dev)
        X = x.double().requires_grad_(True)
        ref.train()
        want = ref(X)
        want.backward(dy.double())
        gx = x.to(dev).requires_grad_(True)
        y = ag.batch_norm_train(gx, mine)
        y.backward(dy.to(dev))
        _cmp(y, want, 2e-5); _cmp(gx.grad, X.grad, 5e-5)
        _cmp(mine.weight.grad, ref.weight.grad, 2e-4); _cmp(mine.bias.grad, ref.bias.grad, 2e-4)
        _cmp(mine.running_mean, ref.running_mean, 1e-5); _cmp(mine.running_var, ref.running_var, 1e-4)
        assert int(mine.num_batches_tracked) == int(ref.num_batches_tracked) == 1


class _ReplayComm:
    """One rank of a data-parallel BatchNorm, the other ranks' contributions replayed: pass `record` collects what this rank
    adds to each all-reduce, pass `replay` returns the recorded sum over all ranks (the forward contribution depends on the local
    rows only, the backward one on the global statistics and the local dy, so three passes give the exact distributed result)."""
    on = True

    def __init__(self, log, totals=None):
        self.log, self.totals, self.i = log, totals, 0

    def all_reduce(self, t, op):
        self.log.append(t.clone())
        if self.totals is not None and self.i < len(self.totals):
            t.copy_(self.totals[self.i])
        self.i += 1
        return t


def test_batch_norm_train_over_row_shards_of_several_ranks(dev):
    """ag.bn_sync: BatchNorm on ragged row shards (7 / 4 / 9 rows) with the statistics of all rows == one BatchNorm on the whole
    batch: y, dx, the SUM of the ranks' local gamma / beta gradients, and the running statistics."""
    torch.manual_seed(5)
    N, Cc, cuts = 20, 70, (0, 7, 11, 20)
    x, dy = torch.randn(N, Cc) * 2 + 0.5, torch.randn(N, Cc)
    ref = torch.nn.BatchNorm1d(Cc).double()
    ref.weight.data.uniform_(0.5, 1.5); ref.bias.data.normal_(0, 0.1)
    X = x.double().requires_grad_(True)
    ref.train()
    want = ref(X)
    want.backward(dy.double())
    sd = {k: v.float() if v.is_floating_point() else v for k, v in ref.state_dict().items()}
    init = {k: v.clone() for k, v in torch.nn.BatchNorm1d(Cc).state_dict().items()}
    init.update(weight=sd['weight'], bias=sd['bias'])

    def run_all(totals_f, totals_b):
        out = []
        for q in range(3):
            bn = torch.nn.BatchNorm1d(Cc)
            bn.load_state_dict(init)
            bn.to(dev)
            log = []
            gx = x[cuts[q]:cuts[q + 1]].to(dev).requires_grad_(True)
            with ag.bn_sync(_ReplayComm(log, None if totals_f is None else [totals_f] + ([totals_b] if totals_b is not None else []))):
                y = ag.batch_norm_train(gx, bn)
                y.backward(dy[cuts[q]:cuts[q + 1]].to(dev))
            out.append((y, gx.grad, bn, log))
        return out

    p1 = run_all(None, None)                                             # records the forward contributions
    tf = sum(o[3][0] for o in p1)
    p2 = run_all(tf, None)                                               # right statistics; records the backward contributions
    tb = sum(o[3][1] for o in p2)
    p3 = run_all(tf, tb)
    _cmp(torch.cat([o[0] for o in p3]), want, 2e-5)
    _cmp(torch.cat([o[1] for o in p3]), X.grad, 5e-5)
    _cmp(sum(o[2].weight.grad for o in p3), ref.weight.grad, 2e-4)
    _cmp(sum(o[2].bias.grad for o in p3), ref.bias.grad, 2e-4)
    for o in p3:
        _cmp(o[2].running_mean, ref.running_mean, 1e-5); _cmp(o[2].running_var, ref.running_var, 1e-4)
        assert int(o[2].num_batches_tracked) == 1


def test_l2norm_mid_summarize_mvm(dev):
    torch.manual_seed(2)
    B, R, D, K = 4, 36, 70, 12
    x, dz = torch.randn(B, R, D), torch.randn(B, R, D)
    X = x.double().requires_grad_(True)
    want = X / (X.pow(2).sum(1, keepdim=True).sqrt() + 1e-8)
    want.backward(dz.double())
    gx = x.to(dev).requires_grad_(True)
    z = ag.l2norm_mid(gx)
    z.backward(dz.to(dev))
    _cmp(z, want, 1e-6); _cmp(gx.grad, X.grad, 2e-6)
    sm, dout = torch.randn(B, R, K), torch.randn(B, K, D)
    S_, X2 = sm.double().requires_grad_(True), x.double().requires_grad_(True)
    want = torch.softmax(S_, 1).transpose(1, 2) @ X2
    want.backward(dout.double())
    gs, gx2 = sm.to(dev).requires_grad_(True), x.to(dev).requires_grad_(True)
    out = ag.summarize(gs, gx2)
    out.backward(dout.to(dev))
    _cmp(out, want, 2e-6); _cmp(gs.grad, S_.grad, 2e-5); _cmp(gx2.grad, X2.grad, 2e-6)
    Ni, Nc = 7, 9
    img, cap, dS = torch.randn(Ni, K, D), torch.randn(Nc, D), torch.randn(Ni, Nc)
    I, Cc = img.double().requires_grad_(True), cap.double().requires_grad_(True)
    want = (I.reshape(Ni * K, D) @ Cc.t()).view(Ni, K, Nc).max(1).values
    want.backward(dS.double())
    gi, gc = img.to(dev).requires_grad_(True), cap.to(dev).requires_grad_(True)
    Sg = ag.mvm_scores(gi, gc)
    Sg.backward(dS.to(dev))
    _cmp(Sg, want, 2e-5); _cmp(gi.grad, I.grad, 2e-5); _cmp(gc.grad, Cc.grad, 2e-5)


def test_gru_cell_nll_bmm_nn(dev):
    """Single-step pieces of VSRN's captioning decoder: nn.GRU cell, log-softmax + masked NLL, small batched product."""
    torch.manual_seed(4)
    B, E, H, V = 7, 20, 24, 57
    rnn = torch.nn.GRU(E, H, 1, batch_first=True).double()
    x, h, dy = torch.randn(B, E), torch.randn(B, H), torch.randn(B, H)
    X, Hh = x.double().requires_grad_(True), h.double().requires_grad_(True)
    out, hn = rnn(X.unsqueeze(1), Hh.unsqueeze(0))
    hn[0].backward(dy.double())
    mine = torch.nn.GRU(E, H, 1, batch_first=True)
    mine.load_state_dict({k: v.float() for k, v in rnn.state_dict().items()})
    mine.to(dev)
    gx, gh = x.to(dev).requires_grad_(True), h.to(dev).requires_grad_(True)
    y = ag.gru_cell(gx, gh, mine)
    y.backward(dy.to(dev))
    _cmp(y, hn[0], 2e-6); _cmp(gx.grad, X.grad, 5e-6); _cmp(gh.grad, Hh.grad, 5e-6)
    for n, p in mine.named_parameters():
        _cmp(p.grad, dict(rnn.named_parameters())[n].grad, 2e-5)
    logits, tgt, mask, dl = torch.randn(B, V) * 3, torch.randint(0, V, (B,)), torch.tensor([1., 1, 0, 1, 1, 0, 1]), torch.randn(B)
    Lg = logits.double().requires_grad_(True)
    want = torch.nn.functional.nll_loss(torch.log_softmax(Lg, 1), tgt, reduction='none') * mask.double()
    want.backward(dl.double())
    gl = logits.to(dev).requires_grad_(True)
    loss = ag.nll_logsoftmax(gl, tgt.to(dev), mask.to(dev))
    loss.backward(dl.to(dev))
    _cmp(loss, want, 2e-6); _cmp(gl.grad, Lg.grad, 2e-6)
    a, b, dc = torch.randn(3, 9, 14), torch.randn(3, 14, 11), torch.randn(3, 9, 11)
    A, Bm = a.double().requires_grad_(True), b.double().requires_grad_(True)
    (A @ Bm).backward(dc.double())
    ga, gb = a.to(dev).requires_grad_(True), b.to(dev).requires_grad_(True)
    c = ag.bmm_nn(ga, gb)
    c.backward(dc.to(dev))
    _cmp(c, A @ Bm, 5e-6); _cmp(ga.grad, A.grad, 5e-6); _cmp(gb.grad, Bm.grad, 5e-6)
