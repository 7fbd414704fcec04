"""GPU: the SCAN scorer + count ranker at BASELINE.json's FULL sizes (MS-COCO 5k images x 25k captions,
f30k 1k x 5k), where the CPU oracle would need hours.  Parity is asserted through size-independent properties
of the path plus scattered oracle spot checks:

  * spot check: a random 24-image x 40-caption sub-problem picked out of the full matrix equals the oracle
    (scores of one pair depend on that pair only);
  * shard invariance: scoring a row block alone gives BIT-identical rows (the multi-GPU row sharding contract);
  * caption permutation: scoring the captions in another order (=> another tile packing) permutes the columns;
  * ranker: ranks of sampled rows / columns equal a host argsort; the rank histogram is a permutation-count
    checksum (sum over all columns of #greater equals the number of strictly-greater pairs)."""
import numpy as np
import pytest
import torch

import itr_oracle as O
from itr_amd import ops

pytestmark = pytest.mark.gpu


def _problem(n_img, seed, dev, D=1024):
    rng = np.random.RandomState(seed)
    n_cap = 5 * n_img
    lens = rng.randint(6, 21, size=n_cap).astype(np.int64)      # SURVEY 8d caption lengths
    off = np.concatenate([[0], np.cumsum(lens)[:-1]])
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    img = ops.l2norm(torch.randn(n_img, 36, D, device=dev, generator=g))
    words = torch.randn(int(lens.sum()), D, device=dev, generator=g) * 0.3
    return img, words, lens, off


@pytest.mark.parametrize("n_img,xa", [(5000, 't2i'), (1000, 't2i'), (1000, 'i2t')])
def test_scan_full_size(dev, n_img, xa):
    img, words, lens, off = _problem(n_img, 11, dev)
    n_cap = len(lens)
    plan = ops.ScanPlan(off, lens, words.shape[0], dev)
    kw = dict(cross_attn=xa, lambda_lse=6.0 if xa == 't2i' else 20.0, lambda_softmax=9.0 if xa == 't2i' else 4.0)
    S = ops.scan_xattn_scores(img, words, plan, **kw)
    assert S.shape == (n_img, n_cap) and bool(torch.isfinite(S).all())

    # -- oracle spot check on scattered rows / columns
    rng = np.random.RandomState(5)
    ri = np.sort(rng.choice(n_img, 24, replace=False))
    ci = np.sort(rng.choice(n_cap, 40, replace=False))
    L = int(lens[ci].max())
    cap = torch.zeros(len(ci), L, words.shape[1])
    for k, c in enumerate(ci):
        cap[k, :lens[c]] = words[off[c]:off[c] + lens[c]].cpu()
    want = O.xattn_score(img[ri].cpu(), cap, [int(lens[c]) for c in ci], xa, 'clipped_l2norm', 'LogSumExp',
                         kw['lambda_lse'], kw['lambda_softmax'])
    got = S[ri][:, ci].cpu()
    assert float((got - want).abs().max()) <= 2e-5        # fp32 tolerance of the small-size parity tests

    # -- shard invariance: a row block scored alone (row0 a multiple of the 4-image workgroup) is bit-identical
    r0, r1 = (n_img // 8) * 3 // 4 * 4, (n_img // 8) * 3 // 4 * 4 + n_img // 8
    Sb = ops.scan_xattn_scores(img[r0:r1].contiguous(), words, plan, **kw)
    assert torch.equal(Sb, S[r0:r1])

    # -- caption permutation: other packing order, same scores (summation order inside a caption is unchanged,
    #    the position of the caption inside its 64-column tile is not -> allow 1e-6)
    perm = rng.permutation(n_cap)
    lens_p = lens[perm]
    off_p = np.concatenate([[0], np.cumsum(lens_p)[:-1]])
    idx = torch.from_numpy(np.concatenate([np.arange(off[c], off[c] + lens[c]) for c in perm])).to(dev)
    words_p = words[idx]
    plan_p = ops.ScanPlan(off_p, lens_p, words_p.shape[0], dev)
    n_chk = min(n_img, 512)
    Sp = ops.scan_xattn_scores(img[:n_chk].contiguous(), words_p, plan_p, **kw)
    assert float((Sp - S[:n_chk][:, torch.from_numpy(perm).to(dev)]).abs().max()) <= 1e-6


def test_ranker_full_size(dev):
    """Count ranker on a 5 000 x 25 000 score matrix: sampled rows / columns against a host argsort, plus the
    global checksum  sum_j t2i_rank[j] == #{(i, j): S[i, j] > S[gt(j), j]}  (ties broken towards lower index)."""
    n_img, n_cap = 5000, 25000
    g = torch.Generator(device=dev)
    g.manual_seed(3)
    S = torch.randn(n_img, n_cap, device=dev, generator=g)
    i_rank, i_top, t_rank, t_best, s_gt = ops.rank_counts(S)
    i_rank, i_top, t_rank = i_rank.cpu().numpy(), i_top.cpu().numpy(), t_rank.cpu().numpy()
    t_top = (t_best & 0xffffffff).cpu().numpy()
    rng = np.random.RandomState(0)
    for i in rng.choice(n_img, 40, replace=False):
        row = S[i].cpu().numpy()
        order = np.argsort(row)[::-1]
        pos = np.empty(n_cap, np.int64)
        pos[order] = np.arange(n_cap)
        assert i_rank[i] == pos[5 * i:5 * i + 5].min()      # evaluation.py:172-178
        assert i_top[i] == order[0]
    for j in rng.choice(n_cap, 40, replace=False):
        col = S[:, j].cpu().numpy()
        order = np.argsort(col)[::-1]
        assert t_rank[j] == int(np.where(order == j // 5)[0][0])   # evaluation.py:208-212
        assert t_top[j] == order[0]
    gt = S[torch.arange(n_cap, device=dev) // 5, torch.arange(n_cap, device=dev)]
    assert torch.equal(gt, s_gt)
    greater = int((S > gt[None, :]).sum().item())            # continuous random scores: no exact ties
    assert int(t_rank.astype(np.int64).sum()) == greater
    assert ops.recall_from_ranks(i_rank)[:3] == tuple(100.0 * int((i_rank < k).sum()) / n_img for k in (1, 5, 10))
