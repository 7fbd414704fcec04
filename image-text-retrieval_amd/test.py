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
    a = ap.parse_args()
    if len(a.model_path) == 1:
        evaluation.evalrank_single(a.model_path[0], data_path=a.data_path, split=a.split, fold5=a.fold5)
    else:
        evaluation.evalrank_ensemble(a.model_path[0], a.model_path[1], data_path=a.data_path, split=a.split, fold5=a.fold5)
