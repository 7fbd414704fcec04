import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "image-text-retrieval_amd"))
import numpy as np, torch
from itr_amd import ops
import bench
dev = torch.device("cuda:0")
wi, wt = bench.make_weights(11353)
wt = {k: v.to(dev) for k, v in wt.items()}
rng = np.random.RandomState(0)
for B in (128, 1000):
    lens = sorted([int(x) for x in rng.randint(6, 21, size=B)], reverse=True)
    toks = torch.from_numpy(np.concatenate([rng.randint(4, 11353, size=l) for l in lens])).to(dev)
    off = torch.from_numpy(np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.int64)).to(dev)
    for _ in range(3): ops.gru_encode(toks, off, lens, wt, True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): ops.gru_encode(toks, off, lens, wt, True)
    torch.cuda.synchronize(); print("B=%d bi-GRU encode: %.2f ms" % (B, (time.perf_counter() - t0) * 100))
