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
    # Tiles are filled exactly where the lengths allow it (csrc/pack_plan.h): the BASELINE length distribution and a COCO-like one
    # (gamma-shaped, long tail) pack to >= 99.5 % column occupancy -- best fit decreasing (rounds 1-4) reached 97.9 %, and every empty
    # column is MFMA time in the kernels.  Also on a rank's eighth of the captions, and never worse than one tile per 16 captions.
    for lens in (rng.randint(6, 21, size=25000), rng.randint(6, 21, size=3125),
                 np.clip(np.round(rng.gamma(9.0, 1.3, size=25000) + 2).astype(int), 3, 60)):
        rc, tb, order = plan(lens)
        assert rc == 0 and lens.sum() / (64.0 * (len(tb) - 1)) >= 0.995, lens.sum() / (64.0 * (len(tb) - 1))
    rc, tb, order = plan([])
    assert rc == 0 and len(tb) == 1
    rc, tb, order = plan([1] * 40)                               # 1-word captions: the caption cap splits them
    assert rc == 0 and list(tb) == [0, 16, 32, 40]


def test_scan_tile_planner_is_never_worse_than_best_fit_decreasing():
    """The exact-fill planner is a greedy and loses to best fit decreasing on a few length sets (found by fuzzing: lengths drawn from
    {7, 13, 31, 32, 33, 64}: 1 % more tiles): itr_scan_plan_tiles runs both and keeps the better (csrc/pack_plan.h::pack_bins)."""
    def bfd(lens):
        open_, tiles = [[] for _ in range(65)], []
        for w in sorted(lens, reverse=True):
            r = w
            while r <= 64 and not open_[r]:
                r += 1
            if r <= 64:
                t = open_[r].pop()
            else:
                t = len(tiles)
                tiles.append(0)
                r = 64
            tiles[t] += 1
            if tiles[t] < 16 and r - w > 0:
                open_[r - w].append(t)
        return len(tiles)
    rng = np.random.RandomState(3)
    for it in range(40):
        n = int(rng.randint(1, 2500))
        lens = (rng.choice([7, 13, 64, 32, 33, 31], size=n) if it % 2 else rng.randint(1, 65, size=n)).astype(np.int64)
        rc, tb, order = plan(lens)
        assert rc == 0 and len(tb) - 1 <= bfd(lens.tolist()), (it, n)


def test_sgr_node_group_plan():
    """The plan of SGR's fused graph steps (csrc/sgr_fused.hip) is the same planner run on NODE counts (words + the global node):
    whole captions, at most 64 node rows and 16 captions per group, every caption exactly once; the BASELINE length distribution
    fills >= 97 % of the rows; and the kernel's softmax scratch (24 tiles of 16 x 16) holds every group the planner can emit:
    sum over a group's captions of ceil(nodes / 16)^2 <= 24."""
    rng = np.random.RandomState(1)
    for lo, hi, n in ((1, 64, 3000), (6, 21, 25000), (1, 3, 500), (30, 64, 200)):
        lens = rng.randint(lo, hi, size=n)
        rc, tb, order = plan(lens + 1)
        assert rc == 0 and sorted(order.tolist()) == list(range(n))
        worst = 0
        for a, b in zip(tb[:-1], tb[1:]):
            nodes = lens[order[a:b]] + 1
            assert 1 <= b - a <= 16 and nodes.sum() <= 64
            worst = max(worst, int((((nodes + 15) // 16) ** 2).sum()))
        assert worst <= 24
        if (lo, hi) == (6, 21):
            assert (lens + 1).sum() / (64.0 * (len(tb) - 1)) >= 0.995
    # the two extreme groups by construction: a 33-node graph + fifteen 2-node ones; a 49-node graph + seven 2-node ones
    for lens in ([32] + [1] * 15, [48] + [1] * 7):
        rc, tb, order = plan(np.asarray(lens) + 1)
        assert rc == 0 and len(tb) == 2
        assert int((((np.asarray(lens) + 1 + 15) // 16) ** 2).sum()) in (24, 23)


def test_gelu_erf_coefficients_are_the_fitted_ones():
    """csrc/itr_common.h::gelu_erf carries the coefficients tools/fit_erf.py computes (and the error figures its comment quotes)."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fit_erf.py")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    coefs = re.findall(r"(-?\d\.\d{9}e[+-]\d{2})f", r.stdout.splitlines()[0])
    assert len(coefs) == 8
    src = open(os.path.join(ROOT, "image-text-retrieval_amd", "csrc", "itr_common.h")).read()
    for c in coefs:
        assert c + "f" in src, "coefficient %s of tools/fit_erf.py is not in itr_common.h" % c
    errs = [float(x) for x in re.findall(r"([0-9.]+e-0[0-9])\s*$", r.stdout, flags=re.M)]
    assert len(errs) == 2 and errs[0] <= 1.3e-7 and errs[1] <= 5e-7


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
    assert lib.itr_rank_counts(None, 0, 0, 0, 0, 5, None, None, None, None, None, 0, None, 0, None) == -1
    assert b"null" in lib.itr_last_error()


EXPERIMENT_SWITCHES = ["ITR_SCAN_DEBUG", "ITR_SCAN_LDS_EXTRA", "ITR_SCAN_TPW", "ITR_SCAN_BF16_ABLATE", "ITR_SGR_TRACE", "ITR_LOC_TRACE",
                       "ITR_LOC_ONE_PER_CU", "ITR_SGR_UNFUSED", "ITR_SGR_PERSISTENT", "ITR_SGRAF_IB", "ITR_SGRAF_GLO_GEMM", "ITR_GEMM_STREAM",
                       "ITR_GRU_NO_SPLITK", "ITR_GRU_NO_OVERLAP", "ITR_GRU_PAIRED", "ITR_GRU_REDUCE_KERNEL", "ITR_MHA_LDS", "ITR_MHA_VALU"]


def test_python_layer_reads_no_environment_variable():
    """the switches that moved from the .so into the Python layer in round 5 (ITR_GEMM_BF16X3, ITR_SGRAF_IB,
    ITR_FORCE_COLLECTIVES ...) are explicit attributes of itr_amd.settings.SETTINGS now.  Nothing under itr_amd/ reads the environment;
    the two entry scripts read only what the launcher hands a rank (the reference reads CUDA_VISIBLE_DEVICES, itr/config.py:412)."""
    pkg = os.path.join(ROOT, "image-text-retrieval_amd")
    for d, _, files in os.walk(os.path.join(pkg, "itr_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(d, f)).read()
                assert re.search(r"os\.environ|getenv\s*\(|from os import", src) is None, os.path.join(d, f)
    allowed = {"RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "ITR_DIST_BACKEND"}
    for f in ("train.py", "test.py"):
        src = open(os.path.join(pkg, f)).read()
        names = set(re.findall(r"environ(?:\.get|\.setdefault)?\(\s*[\"']([A-Z_0-9]+)[\"']", src)) | set(re.findall(r"environ\[\s*[\"']([A-Z_0-9]+)[\"']", src))
        assert names <= allowed, (f, names - allowed)
    from itr_amd.settings import SETTINGS
    assert SETTINGS.sgraf_image_block is None and SETTINGS.force_collectives is False and SETTINGS.virtual_split is None
    assert SETTINGS.exchange == "all_gather" and SETTINGS.agsa_fused and SETTINGS.vsrn_residual_in_epilogue and SETTINGS.sgr_group_rows == 64
    from itr_amd import ops
    assert ops.BF16X3 is False and ops.FP16X3 is False


def test_shipped_library_reads_no_environment_variable():
    """VERDICT r4 #4 / SURVEY 8(b) "no global mutable state beyond the error string": the shipped libitr_hip.so does not import
    getenv, none of the experiment switches of earlier rounds is in the binary, and in the sources every environment read goes
    through ITR_EXP_ENV, which is getenv only under -DITR_EXPERIMENT (tools/ab_build.sh builds).  What used to be switches the
    tests relied on are explicit arguments now (itr_gemm_nt_algo, itr_gru_fwd flag bits, itr_sgraf_scores image_block / flags,
    itr_debug_scan_clock_probe)."""
    blob = open(_lib.LIB_PATH, "rb").read()
    assert b"getenv" not in blob, "libitr_hip.so imports getenv"
    for name in EXPERIMENT_SWITCHES:
        assert name.encode() + b"\0" not in blob, name
    csrc = os.path.join(ROOT, "image-text-retrieval_amd", "csrc")
    for f in sorted(os.listdir(csrc)):
        if not f.endswith((".hip", ".h", ".cpp", ".inc")):
            continue
        for n, line in enumerate(open(os.path.join(csrc, f)), 1):
            if "getenv" in line:
                assert f == "itr_common.h" and "ITR_EXP_ENV" in line, "%s:%d reads the environment directly" % (f, n)


def test_sgraf_image_block_is_sized_to_the_memory_allowed():
    """the pair stage's image block is an argument and itr_sgraf_pick_image_block (host only) sizes it to a byte
    budget: 64 when it fits, then 32 / 16 / 8 / 4, an error when not even 4 images fit; never more than the call's own image count;
    the fused SGR layout (no word-node query / aggregate rows in memory) is about half the step-by-step chain's."""
    lib = _lib.load()
    args = (5000, 25000, 325623, 5192, 1024, 256)                # the 5k x 25k evaluation
    sizes = {}
    for mod in (0, 1):
        for ib in (64, 32, 16, 8, 4):
            sizes[mod, ib] = lib.itr_sgraf_workspace_bytes(*args, mod, ib, 0)
        assert sizes[mod, 64] > sizes[mod, 32] > sizes[mod, 16] > sizes[mod, 8] > sizes[mod, 4] > 0
        assert lib.itr_sgraf_workspace_bytes(*args, mod, 0, 0) == sizes[mod, 64]          # 0 = the default block
        got = C.c_size_t(0)
        for ib in (64, 32, 16, 8, 4):
            assert lib.itr_sgraf_pick_image_block(*args, mod, 0, sizes[mod, ib], C.byref(got)) == ib and got.value == sizes[mod, ib]
            if ib > 4:
                assert lib.itr_sgraf_pick_image_block(*args, mod, 0, sizes[mod, ib] - 1, None) == ib // 2
        assert lib.itr_sgraf_pick_image_block(*args, mod, 0, sizes[mod, 4] - 1, None) == -2        # ITR_ERR_UNSUPPORTED
        assert b"4-image block" in lib.itr_last_error()
    assert sizes[1, 64] < 50e9 and sizes[0, 64] < 45e9                                        # (round 4: 85 GB / 38 GB)
    chain = lib.itr_sgraf_workspace_bytes(*args, 1, 64, 1)                                     # ITR_SGRAF_UNFUSED_STEPS
    assert chain > 1.8 * sizes[1, 64]
    # a 10-image call never pays for 64 images
    small = (10, 50, 600, 10, 1024, 256)
    assert lib.itr_sgraf_workspace_bytes(*small, 1, 64, 0) == lib.itr_sgraf_workspace_bytes(*small, 1, 12, 0)
    assert lib.itr_sgraf_pick_image_block(*small, 1, 0, 1 << 40, None) == 12
