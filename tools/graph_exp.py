#!/usr/bin/env python3
"""Experiment (round 4): the VSE++ text tower (bi-GRU over 5 000 captions: ~85 launches on two streams) captured in a HIP graph against
the eager launches.  Run on the GPU box."""
import os, sys, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "image-text-retrieval_amd"))
import numpy as np, torch
import bench
from itr_amd import ops, _lib
lib = _lib.load()
dev = torch.device("cuda", 0)
n_cap, vocab = 5000, 8481
wi, wt = bench.make_weights(vocab)
wt = {k: v.to(dev).contiguous() for k, v in wt.items()}
lengths, tokens = bench.make_captions(n_cap, vocab)
toks, tok_off, lens_sorted, order = bench.shard_captions(lengths, tokens, 0, n_cap, dev)
len_host = np.asarray(lens_sorted, np.int32)
len_dev = torch.from_numpy(len_host).to(dev)
B, n_tok = len(len_host), int(toks.numel())
V, E = wt['embed.weight'].shape
D = wt['rnn.weight_hh_l0'].shape[1]
wsb = lib.itr_gru_workspace_bytes(n_tok, B, E, D, 1)
ws = torch.empty(wsb, device=dev, dtype=torch.uint8)
out_last = torch.empty(B, D, device=dev)
p = lambda t: C.c_void_p(t.data_ptr())
def enc():
    _lib.check(lib.itr_gru_fwd(p(toks), p(tok_off), p(len_dev), len_host.ctypes.data_as(C.c_void_p), B, n_tok, p(wt['embed.weight']), V, E, D,
                               p(wt['rnn.weight_ih_l0']), p(wt['rnn.weight_hh_l0']), p(wt['rnn.bias_ih_l0']), p(wt['rnn.bias_hh_l0']),
                               p(wt['rnn.weight_ih_l0_reverse']), p(wt['rnn.weight_hh_l0_reverse']), p(wt['rnn.bias_ih_l0_reverse']),
                               p(wt['rnn.bias_hh_l0_reverse']), 0, 0, 1 | 2, None, p(out_last), p(ws), wsb,
                               C.c_void_p(torch.cuda.current_stream().cuda_stream)))
for _ in range(3): enc()
torch.cuda.synchronize()
ref = out_last.clone()
def timeit(f, n=30):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("eager   %.3f ms" % timeit(enc))
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g):
        enc()
    torch.cuda.synchronize()
    out_last.zero_()
    print("replay  %.3f ms" % timeit(g.replay))
    print("equal", torch.equal(out_last, ref))
except Exception as e:
    print("capture failed:", repr(e)[:300])
