"""GPU: the reference-side ctypes binding printed in INTEGRATION.md (section B) is executed verbatim -- it binds the C
ABI with nothing but ctypes + torch tensors, exactly what a maintainer of the reference would paste next to
itr/modalmodule/Objectives.py -- and checked against the oracle."""
import os
import re
import types

import numpy as np
import pytest
import torch

import itr_oracle as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _doc_module():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sec = text[text.index("## B. Bind the C ABI"):text.index("## Entry point")]
    blocks = re.findall(r"```python\n(.*?)```", sec, re.S)
    assert len(blocks) == 2
    so = os.path.join(ROOT, "image-text-retrieval_amd", "itr_amd", "libitr_hip.so")
    src = (blocks[0] + "\n" + blocks[1]).replace('C.CDLL("libitr_hip.so")', 'C.CDLL(%r)' % so)
    mod = types.ModuleType("itr_hip_doc_binding")
    exec(compile(src, "INTEGRATION.md", "exec"), mod.__dict__)
    return mod


def test_integration_md_binding_runs_and_matches_oracle(dev):
    m = _doc_module()
    torch.manual_seed(0)
    rng = np.random.RandomState(0)
    im, s = torch.randn(7, 64), torch.randn(11, 64)
    want_c = O.cosine_sim(im, s)
    assert float((m.cosine_sim(im.to(dev), s.to(dev)).cpu() - want_c).abs().max()) <= 2e-6 * float(want_c.abs().max())
    Ni, Nc, D = 6, 10, 64
    lens = [int(x) for x in rng.randint(2, 15, size=Nc)]
    img = O.l2norm(torch.randn(Ni, 36, D), -1)
    cap = torch.randn(Nc, max(lens), D) * 0.5
    cfg = dict(raw_feature_norm="clipped_l2norm", agg_func="LogSumExp", lambda_softmax=9.0, lambda_lse=6.0)
    got = m.xattn_score_t2i(img.to(dev), cap.to(dev), lens, cfg)
    want = O.xattn_score(img, cap, lens, 't2i', 'clipped_l2norm', 'LogSumExp', 6.0, 9.0)
    assert float((got.cpu() - want).abs().max()) <= 2e-5
    with pytest.raises(ValueError):
        m._chk(m._lib.itr_gemm_nt(None, 1, None, 1, None, None, 1, 2, 2, 2, 0, None))    # null pointers -> ValueError + message
    sims = rng.randn(8, 40)
    sims[:, 1::2] = sims[:, 0::2] - 2.0 ** -30            # neighbours closer than an fp32 ulp (fp32 would tie them, and ties go to the higher index)
    i_rank, i_top, t_rank, t_top = m._ranks(sims)
    w = O.rank_counts(sims)                               # float64, like the reference's argsort
    assert (i_rank == w[0]).all() and (i_top == w[1]).all() and (t_rank == w[2]).all() and (t_top == w[3]).all()
    w32 = O.rank_counts(sims.astype(np.float32))
    assert any((a != b).any() for a, b in zip(w, w32))   # (an fp32 ranker gets this matrix wrong)
