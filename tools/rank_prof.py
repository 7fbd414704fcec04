#!/usr/bin/env python3
"""The rank stage alone (5 000 x 25 000 fp32 matrix, 10 launches) for rocprofv3:  rocprofv3 --kernel-trace --stats -- python3 tools/rank_prof.py"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "image-text-retrieval_amd"))
import torch
from itr_amd import ops
dev = torch.device("cuda", 0)
torch.manual_seed(0)
S = torch.randn(5000, 25000, device=dev)
for _ in range(10):
    out = ops.rank_counts(S)
torch.cuda.synchronize()
print("ok", int(out[0].sum()), int(out[2].sum()))
