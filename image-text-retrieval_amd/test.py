#!/usr/bin/env python3
"""Evaluate checkpoints like the reference's test.py:  python test.py MODEL_PATH [MODEL_PATH_2] [--data-path P]
[--split test|testall|dev] [--fold5]  ->  <run dir>/<data_name>[_5fold]_{single,ensemble}_result.yaml"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from itr_amd.metricmodule import evaluation     # noqa: E402

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("model_path", nargs="+")
    ap.add_argument("--data-path", default=None)
    ap.add_argument("--split", default="test")
    ap.add_argument("--fold5", action="store_true")
    ap.add_argument("--fast", action="store_true",
                    help="sharded device-resident evaluation (evalpipe.evaluate_precomp); run under torch.distributed.run for >1 GPU")
    a = ap.parse_args()
    if a.fast:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        if int(os.environ.get("WORLD_SIZE", "1")) > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group("nccl")
        evaluation.evalrank_fast(a.model_path[0], data_path=a.data_path, split=a.split, fold5=a.fold5)
        if dist.is_initialized():
            dist.destroy_process_group()
    elif len(a.model_path) == 1:
        evaluation.evalrank_single(a.model_path[0], data_path=a.data_path, split=a.split, fold5=a.fold5)
    else:
        evaluation.evalrank_ensemble(a.model_path[0], a.model_path[1], data_path=a.data_path, split=a.split, fold5=a.fold5)
