"""GPU: the opt-in split-operand main loops of the SCAN kernel (csrc/scan_mainloop_bf16.inc, STUDY_SPLIT_PRECISION.md): "bf16x3" (bf16
planes, ~3e-6 of the fp32 kernel) and "fp16x3" (fp16 planes with scaled lo and a second accumulator set, ~3e-7: fp32 rounding
level) -- the same scores as the fp32 kernel and the oracle on the unit-norm operands of this path, every epilogue shared."""
import numpy as np
import pytest
import torch

import itr_oracle as O
from itr_amd import ops

pytestmark = pytest.mark.gpu


def _problem(n_img, n_cap, D, seed, dev, lo=1, hi=20):
    rng = np.random.RandomState(seed)
    lens = rng.randint(lo, hi + 1, size=n_cap).astype(np.int64)
    off = np.concatenate([[0], np.cumsum(lens)[:-1]])
    g = torch.Generator().manual_seed(seed)
    img = torch.randn(n_img, 36, D, generator=g)
    img = img / img.norm(dim=-1, keepdim=True)
    words = torch.randn(int(lens.sum()), D, generator=g)
    words = words / words.norm(dim=-1, keepdim=True)
    return img.to(dev), words.to(dev), lens, off


def _oracle(img, words, lens, off, xa, norm, agg, ll, ls):
    L = int(lens.max())
    cap = torch.zeros(len(lens), L, words.shape[1])
    for k in range(len(lens)):
        cap[k, :lens[k]] = words[off[k]:off[k] + lens[k]].cpu()
    return O.xattn_score(img.cpu(), cap, [int(x) for x in lens], xa, norm, agg, ll, ls)


@pytest.mark.parametrize("xa,norm,agg", [('t2i', 'clipped_l2norm', 'LogSumExp'), ('t2i', 'softmax', 'Mean'), ('t2i', 'l2norm', 'Max'),
                                         ('t2i', 'no_norm', 'Sum'), ('i2t', 'clipped_l2norm', 'LogSumExp'), ('i2t', 'l2norm', 'Mean'),
                                         ('i2t', 'clipped', 'Sum')])
@pytest.mark.parametrize("n_img,D", [(7, 64), (9, 32), (5, 256)])
@pytest.mark.parametrize("prec,tol", [('bf16x3', 1e-5), ('fp16x3', 2e-6)])
def test_scan_bf16x3_vs_oracle(dev, xa, norm, agg, n_img, D, prec, tol):
    img, words, lens, off = _problem(n_img, 23, D, 100 + n_img, dev)
    plan = ops.ScanPlan(off, lens, words.shape[0], dev)
    ll, ls = (6.0, 9.0) if xa == 't2i' else (20.0, 4.0)
    kw = dict(cross_attn=xa, raw_feature_norm=norm, agg_func=agg, lambda_lse=ll, lambda_softmax=ls)
    got = ops.scan_xattn_scores(img, words, plan, precision=prec, **kw).cpu()
    want = _oracle(img, words, lens, off, xa, norm, agg, ll, ls)
    fp32 = ops.scan_xattn_scores(img, words, plan, **kw).cpu()
    scale = max(1.0, float(want.abs().max()))
    assert float((got - want).abs().max()) <= max(tol, 4e-6) * scale    # fp32 path: 2e-5 in tests/test_scan_gpu.py
    assert float((got - fp32).abs().max()) <= tol * scale


def test_scan_bf16x3_full_size_against_fp32(dev):
    """1 000 x 5 000 at D = 1 024 (the f30k evaluation size): bf16x3 == fp32 kernel within 1e-5, bit-identical under row sharding."""
    img, words, lens, off = _problem(1000, 5000, 1024, 11, dev, lo=6)
    plan = ops.ScanPlan(off, lens, words.shape[0], dev)
    ws = ops.scan_prepare(img, words, plan, 't2i')
    S0 = ops.scan_xattn_scores(img, words, plan, workspace=ws)
    S1 = ops.scan_xattn_scores(img, words, plan, workspace=ws, precision='bf16x3')
    assert float((S1 - S0).abs().max()) <= 1e-5 and float((S1 - S0).abs().mean()) <= 5e-7
    Sb = ops.scan_xattn_scores(img[248:376].contiguous(), words, plan, precision='bf16x3')
    assert torch.equal(Sb, S1[248:376])
    # fp16 planes: fp32 rounding level, deterministic (a register-reuse race made this build produce sporadic infinities with
    # two workgroups per CU until the tail prefetches were drained inside the last chunk -- STUDY_SPLIT_PRECISION.md), three runs identical
    S2 = ops.scan_xattn_scores(img, words, plan, workspace=ws, precision='fp16x3')
    assert bool(torch.isfinite(S2).all())
    assert float((S2 - S0).abs().max()) <= 1e-6 and float((S2 - S0).abs().mean()) <= 5e-8
    for _ in range(2):
        assert torch.equal(ops.scan_xattn_scores(img, words, plan, workspace=ws, precision='fp16x3'), S2)


def test_scan_bf16x3_arguments(dev):
    img, words, lens, off = _problem(4, 6, 64, 1, dev)
    plan = ops.ScanPlan(off, lens, words.shape[0], dev)
    with pytest.raises(ValueError):
        ops.scan_xattn_scores(img, words, plan, precision='fp16')
    lens2 = np.array([70, 5], np.int64)
    off2 = np.array([0, 70], np.int64)
    w2 = torch.randn(75, 64, device=dev)
    with pytest.raises(NotImplementedError):
        ops.scan_xattn_scores(img, w2, ops.ScanPlan(off2, lens2, 75, dev), precision='bf16x3')
