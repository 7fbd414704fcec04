"""GPU: edge cases of the hot path -- empty operands, single elements, the longest supported caption, one-past-the-
limit rejections, captions that fill a 64-column tile exactly, rows that are all zero."""
import numpy as np
import pytest
import torch

import itr_oracle as O
from itr_amd import ops

pytestmark = pytest.mark.gpu


def test_empty_operands(dev):
    D = 64
    img0, img2 = torch.zeros(0, 36, D, device=dev), ops.l2norm(torch.randn(2, 36, D, device=dev))
    cap = torch.randn(3, 5, D, device=dev)
    assert ops.scan_xattn_padded(img0, cap, [5, 4, 2]).shape == (0, 3)
    assert ops.cosine_scores(torch.zeros(0, D, device=dev), torch.randn(4, D, device=dev)).shape == (0, 4)
    assert ops.cosine_scores(torch.randn(4, D, device=dev), torch.zeros(0, D, device=dev)).shape == (4, 0)
    assert ops.l2norm(torch.zeros(0, D, device=dev)).shape == (0, D)
    assert ops.linear(torch.zeros(0, D, device=dev), torch.randn(8, D, device=dev)).shape == (0, 8)
    loss = ops.hinge_loss(torch.zeros(1, 1, device=dev), 0.2, True)
    assert float(loss) == 0.0                                  # a single pair has no negatives
    assert img2.shape == (2, 36, D)


@pytest.mark.parametrize("xa", ['t2i', 'i2t'])
def test_longest_caption_and_exact_tile_fill(dev, xa):
    """One caption of 64 words (the tile width), four captions of 16 words (one full tile), and a 1-word caption."""
    torch.manual_seed(0)
    D = 32
    lens = [64, 16, 16, 16, 16, 1]
    img = O.l2norm(torch.randn(5, 36, D), -1)
    cap = torch.randn(len(lens), 64, D) * 0.5
    want = O.xattn_score(img, cap, lens, xa)
    got = ops.scan_xattn_padded(img.to(dev), cap.to(dev), lens, cross_attn=xa)
    assert float((got.cpu() - want).abs().max()) <= 2e-5
    # 65..96 words: scored by the pair kernels of the training path (no 64-column tile holds them); 97 is rejected
    lens2 = [82, 5, 96, 64, 65]
    cap2 = torch.randn(len(lens2), 96, D) * 0.5
    want2 = O.xattn_score(img, cap2, lens2, xa)
    got2 = ops.scan_xattn_padded(img.to(dev), cap2.to(dev), lens2, cross_attn=xa)
    assert float((got2.cpu() - want2).abs().max()) <= 2e-5
    with pytest.raises(NotImplementedError):
        ops.scan_xattn_padded(img.to(dev), torch.randn(1, 97, D, device=dev), [97], cross_attn=xa)
    with pytest.raises((ValueError, NotImplementedError)):
        ops.scan_xattn_padded(img.to(dev), cap.to(dev), [64, 16, 16, 16, 16, 0], cross_attn=xa)   # zero-length caption


def test_zero_rows_and_identical_scores(dev):
    """All-zero word / region rows go through the eps guards like the reference (no NaN), and a constant score matrix
    ranks by the tie rule."""
    torch.manual_seed(1)
    D = 32
    img = O.l2norm(torch.randn(3, 36, D), -1)
    img[1, 5] = 0
    cap = torch.randn(4, 6, D)
    cap[2, 1] = 0
    lens = [6, 5, 4, 3]
    for xa in ('t2i', 'i2t'):
        want = O.xattn_score(img, cap, lens, xa)
        got = ops.scan_xattn_padded(img.to(dev), cap.to(dev), lens, cross_attn=xa)
        assert torch.isfinite(got).all() and float((got.cpu() - want).abs().max()) <= 2e-5
    S = torch.full((4, 20), 0.25, device=dev)
    i_rank, i_top, t_rank, t_best, _ = ops.rank_counts(S)
    want = O.rank_counts(np.full((4, 20), 0.25))
    assert (i_rank.cpu().numpy() == want[0]).all() and (t_rank.cpu().numpy() == want[2]).all()


def test_gru_single_token_and_single_caption(dev):
    torch.manual_seed(2)
    V, E, D = 20, 8, 32
    w = {'embed.weight': torch.randn(V, E) * 0.1}
    for suf in ('', '_reverse'):
        w['rnn.weight_ih_l0' + suf] = torch.randn(3 * D, E) * 0.2
        w['rnn.weight_hh_l0' + suf] = torch.randn(3 * D, D) * 0.2
        w['rnn.bias_ih_l0' + suf] = torch.randn(3 * D) * 0.1
        w['rnn.bias_hh_l0' + suf] = torch.randn(3 * D) * 0.1
    wd = {k: v.to(dev) for k, v in w.items()}
    for lens in ([1], [3], [2, 1, 1]):
        ids = torch.zeros(len(lens), max(lens), dtype=torch.long)
        for b, l in enumerate(lens):
            ids[b, :l] = torch.randint(0, V, (l,))
        want, _ = O.encoder_text(ids, lens, w, True, False, False, None)
        toks = torch.cat([ids[b, :l] for b, l in enumerate(lens)]).to(dev)
        off = torch.as_tensor(np.concatenate([[0], np.cumsum(lens)[:-1]]), dtype=torch.int64, device=dev)
        got = ops.gru_encode(toks, off, lens, wd, True)
        flat = torch.cat([want[b, :l] for b, l in enumerate(lens)])
        assert float((got.cpu() - flat).abs().max()) <= 5e-6
    with pytest.raises(ValueError):
        ops.gru_encode(torch.zeros(3, dtype=torch.long, device=dev), torch.tensor([0, 1], device=dev), [1, 2], wd, True)   # not sorted
    with pytest.raises(IndexError):
        ops.gru_encode(torch.full((2,), V, dtype=torch.long, device=dev), torch.tensor([0], device=dev), [2], wd, True)    # id out of range
