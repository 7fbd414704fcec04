import sys, time, numpy as np, torch
sys.path.insert(0, "image-text-retrieval_amd"); sys.path.insert(0, "oracle")
from itr_amd import ops
import itr_oracle as O
dev = torch.device("cuda", 0)
def problem(n_img, seed, D=1024):
    rng = np.random.RandomState(seed); n_cap = 5 * n_img
    lens = rng.randint(6, 21, size=n_cap).astype(np.int64); off = np.concatenate([[0], np.cumsum(lens)[:-1]])
    g = torch.Generator(device=dev); g.manual_seed(seed)
    img = ops.l2norm(torch.randn(n_img, 36, D, device=dev, generator=g))
    words = ops.l2norm(torch.randn(int(lens.sum()), D, device=dev, generator=g))
    return img, words, lens, off
# small parity vs oracle
img, words, lens, off = problem(24, 3)
plan = ops.ScanPlan(off, lens, words.shape[0], dev)
for xa in ('t2i', 'i2t'):
    kw = dict(cross_attn=xa, lambda_lse=6.0 if xa == 't2i' else 20.0, lambda_softmax=9.0 if xa == 't2i' else 4.0)
    S0 = ops.scan_xattn_scores(img, words, plan, **kw)
    S1 = ops.scan_xattn_scores(img, words, plan, precision='bf16x3', **kw)
    S2 = ops.scan_xattn_scores(img, words, plan, precision='fp16x3', **kw)
    L = int(lens.max()); cap = torch.zeros(len(lens), L, 1024)
    for k in range(len(lens)): cap[k, :lens[k]] = words[off[k]:off[k]+lens[k]].cpu()
    want = O.xattn_score(img.cpu(), cap, [int(x) for x in lens], xa, 'clipped_l2norm', 'LogSumExp', kw['lambda_lse'], kw['lambda_softmax'])
    print(xa, "fp32 vs oracle %.2e | bf16x3 vs oracle %.2e | bf16x3 vs fp32 %.2e | fp16x3 vs oracle %.2e | fp16x3 vs fp32 %.2e" % (
        (S0.cpu()-want).abs().max(), (S1.cpu()-want).abs().max(), (S1-S0).abs().max(), (S2.cpu()-want).abs().max(), (S2-S0).abs().max()))
# timing at 1k and 5k
for n_img in (1000, 5000):
    img, words, lens, off = problem(n_img, 11)
    plan = ops.ScanPlan(off, lens, words.shape[0], dev)
    ws = ops.scan_prepare(img, words, plan, 't2i')
    for prec in ('fp32', 'bf16x3', 'fp16x3'):
        f = lambda: ops.scan_xattn_scores(img, words, plan, workspace=ws, precision=prec)
        S = f(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(2): S = f()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 2
        print(n_img, prec, "%.1f ms" % (dt * 1e3), "%.1f M pairs/s" % (n_img * 5 * n_img / dt / 1e6))
        if prec == 'fp32': Sf = S
        else: print("   max|dS| %.2e  mean %.2e" % ((S - Sf).abs().max().item(), (S - Sf).abs().mean().item()))
