"""CPU: the C-ABI library loads, exports every symbol include/itr_hip.h declares, and its
host-only entry points (tile planner, recall summary, argument validation) behave.  No compute
kernel is launched here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from itr_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = open(os.path.join(ROOT, "include", "itr_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(itr_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    syms = header_symbols()
    assert len(syms) >= 15
    for s in syms:
        assert hasattr(lib, s), "libitr_hip.so does not export %s" % s
        assert s in _lib.SIGNATURES, "ctypes binding has no signature for %s" % s
    assert sorted(_lib.SIGNATURES) == syms
    assert lib.itr_abi_version() == _lib.ABI_VERSION


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libitr_hip.so")
    with pytest.raises(_lib.ItrLibraryMissing):
        _lib.load()


def test_ops_reject_cpu_tensors():
    import torch
    from itr_amd import ops
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.cosine_scores(torch.zeros(2, 4), torch.zeros(3, 4))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.l2norm(torch.zeros(2, 4))


def plan(lens):
    lib = _lib.load()
    lens = np.asarray(lens, dtype=np.int32)
    tb = np.zeros(len(lens) + 1, dtype=np.int32)
    order = np.zeros(max(len(lens), 1), dtype=np.int32)
    nt = C.c_int64(0)
    rc = lib.itr_scan_plan_tiles(lens.ctypes.data_as(C.c_void_p), len(lens), 64, tb.ctypes.data_as(C.c_void_p),
                                 order.ctypes.data_as(C.c_void_p), C.byref(nt))
    return rc, tb[:nt.value + 1], order[:len(lens)]


def test_scan_tile_planner():
    rng = np.random.RandomState(0)
    for lo, hi, n in ((1, 65, 500), (6, 21, 25000), (64, 65, 10), (1, 2, 100)):
        lens = rng.randint(lo, hi, size=n)
        rc, tb, order = plan(lens)
        assert rc == 0 and tb[0] == 0 and tb[-1] == n
        assert sorted(order.tolist()) == list(range(n))          # every caption exactly once
        for a, b in zip(tb[:-1], tb[1:]):
            assert 1 <= b - a <= 16                              # <= SC_MAXCAP captions per tile
            assert lens[order[a:b]].sum() <= 64                  # whole captions only, <= 64 words
    # the BASELINE length distribution packs to >= 97 % column occupancy
    lens = rng.randint(6, 21, size=25000)
    rc, tb, order = plan(lens)
    assert lens.sum() / (64.0 * (len(tb) - 1)) >= 0.97
    rc, tb, order = plan([])
    assert rc == 0 and len(tb) == 1
    rc, tb, order = plan([1] * 40)                               # 1-word captions: the caption cap splits them
    assert rc == 0 and list(tb) == [0, 16, 32, 40]


def test_scan_tile_planner_errors():
    lib = _lib.load()
    rc, _, _ = plan([5, 0, 3])
    assert rc == -1 and b"length 0" in lib.itr_last_error()
    with pytest.raises(ValueError):
        _lib.check(rc)
    rc, _, _ = plan([5, 65])
    assert rc == -2
    with pytest.raises(NotImplementedError):
        _lib.check(rc)


def test_recall_from_ranks_matches_reference_formula():
    from itr_amd import ops
    import itr_oracle as O
    rng = np.random.RandomState(1)
    for n in (1, 2, 7, 100, 1001):
        ranks = rng.randint(0, 50, size=n)
        got = ops.recall_from_ranks(ranks)
        want = O.recall_from_ranks(ranks)
        assert got == pytest.approx(want, abs=1e-12)


def test_badarg_codes_without_gpu():
    lib = _lib.load()
    # null pointers / bad enums are rejected before anything touches the device
    assert lib.itr_gemm_nt(None, 4, None, 4, None, None, 4, 1, 1, 4, 0, None) == -1
    assert lib.itr_l2norm_rows(None, None, 1, 4, 1e-8, 0, 0, None) == -1
    assert lib.itr_rank_counts(None, 0, 0, 0, 0, 5, None, None, None, None, None, None) == -1
    assert b"null" in lib.itr_last_error()
