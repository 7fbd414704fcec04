"""GPU: EncoderText with num_layers > 1 (TextEncoder.py:31 -- a constructor argument of the reference that none of its configs sets)
against the reference's own arithmetic, torch.nn.GRU on packed sequences (TextEncoder.py:45-55), forward in both modes and gradients."""
import numpy as np
import pytest
import torch
from torch.nn.utils.rnn import pack_padded_sequence, pad_packed_sequence

from itr_amd.modalmodule import TextEncoder

pytestmark = pytest.mark.gpu


def _reference(enc_cpu, ids, lens, bi, last, D):
    """EncoderText.forward as the reference writes it (TextEncoder.py:38-70) on the CPU copy of the module."""
    x = enc_cpu.embed(ids)
    out, _ = pad_packed_sequence(enc_cpu.rnn(pack_padded_sequence(x, lens, batch_first=True))[0], batch_first=True)
    if bi:
        out = (out[:, :, :D] + out[:, :, D:]) / 2
    if last:
        idx = (torch.tensor(lens) - 1).view(-1, 1, 1).expand(len(lens), 1, D)
        out = out.gather(1, idx).squeeze(1)
    return out / (out.pow(2).sum(-1, keepdim=True).sqrt() + 1e-8)


@pytest.mark.parametrize("layers,bi,method", [(2, True, None), (3, False, None), (2, True, 'VSE++'), (2, False, 'VSE++')])
def test_multilayer_gru_matches_nn_gru(dev, layers, bi, method):
    torch.manual_seed(layers * 10 + bi)
    V, E, D, B = 40, 12, 32, 7
    enc = TextEncoder.EncoderText(V, E, D, layers, use_bi_gru=bi, method_name=method)
    import copy
    ref = copy.deepcopy(enc)
    enc.cuda()
    rng = np.random.RandomState(1)
    lens = sorted([int(x) for x in rng.randint(1, 9, size=B)], reverse=True)
    ids = torch.from_numpy(rng.randint(0, V, size=(B, max(lens))))
    want = _reference(ref, ids, lens, bi, method is not None, D)
    # ---- evaluation mode: fused kernels
    enc.eval()
    with torch.no_grad():
        got, cap_len = enc(ids.cuda(), lens)
    assert list(cap_len) == lens
    if method is None:
        for b, l in enumerate(lens):
            assert float((got[b, :l].cpu() - want[b, :l]).abs().max()) <= 5e-6
            assert float(got[b, l:].abs().max() if l < got.shape[1] else 0.0) == 0.0
    else:
        assert float((got.cpu() - want).abs().max()) <= 5e-6
    # ---- training mode: the tape; gradients of a scalar against torch autograd on the CPU module
    enc.train()
    out_t, _ = enc(ids.cuda(), lens)
    assert out_t.requires_grad
    gen = torch.Generator().manual_seed(3)
    wgt = torch.randn(want.shape, generator=gen)
    if method is None:
        mask = torch.zeros_like(wgt)
        for b, l in enumerate(lens):
            mask[b, :l] = 1
        wgt = wgt * mask
    (out_t * wgt.cuda()).sum().backward()
    for p_ in ref.parameters():
        p_.grad = None
    (want * wgt).sum().backward()
    for (n, p), (_, q) in zip(enc.named_parameters(), ref.named_parameters()):
        assert p.grad is not None, n
        scale = max(1.0, float(q.grad.abs().max()))
        assert float((p.grad.cpu() - q.grad).abs().max()) <= 2e-5 * scale, n
