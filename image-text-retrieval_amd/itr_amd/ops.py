"""Thin torch-tensor front end of the C ABI (include/itr_hip.h).

torch is plumbing here: it owns device memory and the current HIP stream; every op below hands
raw device pointers + sizes to libitr_hip.so.  CPU tensors are rejected -- there is no fallback.
"""
import ctypes as C
import weakref

import numpy as np
import torch

from . import _lib
from .settings import SETTINGS

SCAN_NT = 64
SCAN_R = 36            # regions per image the fused SCAN / SGRAF kernels are built for (csrc/scan_common.h SC_R)
_NORMS = {'clipped_l2norm': 0, 'l2norm': 1, 'softmax': 2, 'no_norm': 3, 'clipped': 4, 'l1norm': 5,
          'clipped_l1norm': 6}
_AGGS = {'LogSumExp': 0, 'Max': 1, 'Sum': 2, 'Mean': 3}
_ACTS = {None: 0, 'none': 0, 'relu': 1, 'tanh': 2, 'sigmoid': 3, 'gelu': 4, 'leaky_relu': 5, 'nan_to_zero': 6}


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _dev(t, dtype=torch.float32, name="tensor"):
    if not torch.is_tensor(t):
        raise TypeError("%s must be a torch tensor" % name)
    if not t.is_cuda:
        raise RuntimeError("%s is on %s: the itr_amd ops only run on the GPU (no CPU fallback)" % (name, t.device))
    if t.dtype != dtype:
        raise TypeError("%s must be %s, got %s" % (name, dtype, t.dtype))
    return t.contiguous()


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def h2d(arr, device, dtype=None):
    """Small host array -> device tensor through PINNED staging with a non-blocking copy.  A copy from pageable memory
    (torch.as_tensor(list, device=...), .to(device) of a numpy view) is synchronous: the host stops until the stream
    has drained, which put the whole training step in lock-step with the GPU (1.8 ms per call measured)."""
    t = torch.as_tensor(np.ascontiguousarray(arr)) if not torch.is_tensor(arr) else arr
    if dtype is not None and t.dtype != dtype:
        t = t.to(dtype)
    if torch.device(device).type != 'cuda':
        return t.to(device)
    return t.pin_memory().to(device, non_blocking=True)


def _host_i32(a):
    arr = np.ascontiguousarray(np.asarray([int(x) for x in a] if not isinstance(a, np.ndarray) else a, dtype=np.int32))
    return arr


# ------------------------------------------------------------------------------------------
def l2norm(x, dim=-1, eps=1e-8):
    """utils.l2norm (eps after the sqrt).  Any `dim`: the tensor is viewed with `dim` last."""
    return _norm(x, dim, eps, 0)


def l1norm(x, dim=-1, eps=1e-8):
    return _norm(x, dim, eps, 1)


def normalize(x, dim=-1, eps=1e-12):
    """F.normalize(p=2) semantics: x / max(||x||, eps)."""
    return _norm(x, dim, eps, 2)


def _norm(x, dim, eps, kind, take_abs=False):
    lib = _lib.load()
    x = _dev(x, name="x")
    dim = dim % x.dim()
    moved = dim != x.dim() - 1
    xs = x.movedim(dim, -1).contiguous() if moved else x
    y = torch.empty_like(xs)
    rows = xs.numel() // xs.shape[-1] if xs.numel() else 0
    _lib.check(lib.itr_l2norm_rows(_p(xs), _p(y), rows, xs.shape[-1], eps, kind, int(take_abs), _stream()))
    return y.movedim(-1, dim).contiguous() if moved else y


def mean_mid(x):
    """torch.mean(x, 1) for x (B, R, F)."""
    lib = _lib.load()
    x = _dev(x, name="x")
    B, R, F_ = x.shape
    y = torch.empty(B, F_, device=x.device, dtype=torch.float32)
    for b0 in range(0, B, 65535):
        b1 = min(B, b0 + 65535)
        _lib.check(lib.itr_mean_mid(_p(x[b0:b1]), _p(y[b0:b1]), b1 - b0, R, F_, _stream()))
    return y


# STUDY switches (STUDY_SPLIT_PRECISION.md): True routes the plain tower / score GEMMs (linear, linear_strided, cosine_scores) through the
# split-bf16 kernel (terms = 3, ~1e-6 of fp32) / its fp16-plane variant.  Set by the study tools and their tests in their own
# process, never from the environment: the product GEMM is exact fp32.
BF16X3 = False
FP16X3 = False


def _weight_planes(w):
    """hi / lo planes of a weight tensor.  Not cached: call sites hand over detached views and temporaries whose address
    and version counter do not identify their contents (a cached plane of a recycled temporary gave wrong CAMERA scores);
    splitting costs one read + one write of the weight, ~1 % of the GEMM it feeds."""
    return split_bf16(w)


GEMM_ALGOS = {None: 0, "auto": 0, "tile": 1, "stream_plain": 2, "stream_xcd": 3, "skinny": 4}


def linear(x, weight, bias=None, act=None, algo=None):
    """act(x @ weight^T + bias) on the fp32 MFMA GEMM.  x (..., K), weight (N, K).  algo: None = the library's own choice;
    "tile" / "stream_plain" / "stream_xcd" force a kernel (itr_gemm_nt_algo: cross-checks, bit-identical results)."""
    lib = _lib.load()
    x = _dev(x, name="x")
    weight = _dev(weight, name="weight")
    K = x.shape[-1]
    if weight.shape[1] != K:
        raise ValueError("linear: x (..., %d) vs weight %s" % (K, tuple(weight.shape)))
    M = x.numel() // K if K else 0
    N = weight.shape[0]
    b = _dev(bias, name="bias") if bias is not None else None
    out = torch.empty(x.shape[:-1] + (N,), device=x.device, dtype=torch.float32)
    if FP16X3 and M and N and K % 32 == 0:
        gemm_nt_f16x3(split_f16(x.reshape(M, K)), split_f16(weight), b, act, out=out.view(M, N))
        return out
    if BF16X3 and M and N and K % 32 == 0:
        gemm_nt_bf16(split_bf16(x.reshape(M, K)), _weight_planes(weight), b, 3, act, out=out.view(M, N))
        return out
    if algo is not None:
        _lib.check(lib.itr_gemm_nt_algo(_p(x), K, _p(weight), K, _p(b), _p(out), N, M, N, K, _ACTS[act], GEMM_ALGOS[algo], _stream()))
        return out
    _lib.check(lib.itr_gemm_nt(_p(x), K, _p(weight), K, _p(b), _p(out), N, M, N, K, _ACTS[act], _stream()))
    return out


def linear_strided(base, lda, M, K, weight, bias=None, act=None, out=None):
    """act(A @ weight^T + bias) where row m of A starts at base.data_ptr() + m*lda floats and is K long (rows may
    overlap: SAEM's Conv2d(1, C, (k, 768)) over a token sequence is this GEMM with lda = 768, K = k*768)."""
    lib = _lib.load()
    base = _dev(base, name="base")
    weight = _dev(weight, name="weight")
    if weight.shape[1] != K:
        raise ValueError("linear_strided: K=%d vs weight %s" % (K, tuple(weight.shape)))
    if (M - 1) * lda + K > base.numel():
        raise ValueError("linear_strided: the last row reads past the end of the buffer")
    N = weight.shape[0]
    b = _dev(bias, name="bias") if bias is not None else None
    if out is None:
        out = torch.empty(M, N, device=base.device, dtype=torch.float32)
    if FP16X3 and M and N and K % 32 == 0 and lda % 32 == 0:
        return gemm_nt_f16x3(split_f16(base), split_f16(weight), b, act, M=M, K=K, lda=lda, out=out)
    if BF16X3 and M and N and K % 32 == 0 and lda % 32 == 0:
        return gemm_nt_bf16(split_bf16(base), _weight_planes(weight), b, 3, act, M=M, K=K, lda=lda, out=out)
    _lib.check(lib.itr_gemm_nt(_p(base), lda, _p(weight), K, _p(b), _p(out), out.stride(0), M, N, K, _ACTS[act], _stream()))
    return out


def proj_l2norm(images, weight, bias, no_imgnorm=False, use_abs=False):
    """EncoderImagePrecomp.forward (ImgEncoder.py:133-147)."""
    lib = _lib.load()
    images = _dev(images, name="images")
    weight = _dev(weight, name="fc.weight")
    bias = _dev(bias, name="fc.bias")
    F_, D = images.shape[-1], weight.shape[0]
    if weight.shape[1] != F_:
        raise ValueError("proj_l2norm: feature dim %d vs fc.weight %s" % (F_, tuple(weight.shape)))
    rows = images.numel() // F_
    out = torch.empty(images.shape[:-1] + (D,), device=images.device, dtype=torch.float32)
    _lib.check(lib.itr_proj_l2norm(_p(images), _p(weight), _p(bias), _p(out), rows, F_, D, int(no_imgnorm),
                                   int(use_abs), _stream()))
    return out


def _score_out(out, Ni, Nc, dev, name):
    """The (Ni, Nc) result buffer of a score op: a fresh tensor, or the caller's -- possibly a column block of a wider score
    matrix (row stride > Nc; the sharded evaluation scores caption ranges into their columns, evalpipe.py)."""
    if out is None:
        return torch.empty(Ni, Nc, device=dev, dtype=torch.float32)
    if not (out.is_cuda and out.dtype == torch.float32 and tuple(out.shape) == (Ni, Nc) and (Nc <= 1 or out.stride(1) == 1)):
        raise ValueError("%s: out must be a (%d, %d) fp32 CUDA view with contiguous rows" % (name, Ni, Nc))
    return out


def cosine_scores(im, s, out=None):
    """cosine_sim (Objectives.py:18-21): im @ s^T."""
    lib = _lib.load()
    im = _dev(im, name="im")
    s = _dev(s, name="s")
    if im.dim() != 2 or s.dim() != 2 or im.shape[1] != s.shape[1]:
        raise ValueError("cosine_scores: expected (Ni, D) and (Nc, D), got %s and %s" % (tuple(im.shape), tuple(s.shape)))
    if FP16X3 and im.shape[0] and s.shape[0] and im.shape[1] % 32 == 0:
        return gemm_nt_f16x3(split_f16(im), split_f16(s), None, out=out)
    if BF16X3 and im.shape[0] and s.shape[0] and im.shape[1] % 32 == 0:
        return gemm_nt_bf16(split_bf16(im), split_bf16(s), None, 3, out=out)
    S = _score_out(out, im.shape[0], s.shape[0], im.device, "cosine_scores")
    _lib.check(lib.itr_cosine_scores(_p(im), _p(s), _p(S), im.shape[0], s.shape[0], im.shape[1], S.stride(0) if S.shape[0] > 1 else max(S.shape[1], S.stride(0)), _stream()))
    return S


def pdist_cos(x1, x2, out=None):
    """Objectives.pdist_cos (:310-323): rows / ||row|| (no eps), mm, NaN -> 0."""
    lib = _lib.load()
    a = _norm(x1, -1, 0.0, 3)
    b = _norm(x2, -1, 0.0, 3)
    if a.dim() != 2 or b.dim() != 2 or a.shape[1] != b.shape[1]:
        raise ValueError("pdist_cos: expected (Ni, D) and (Nc, D), got %s and %s" % (tuple(a.shape), tuple(b.shape)))
    if FP16X3 or BF16X3:      # study routing: the split-precision GEMMs have no NaN epilogue
        return torch.nan_to_num_(cosine_scores(a, b, out=out), nan=0.0, posinf=float('inf'), neginf=float('-inf'))
    # `res[res != res] = 0` (:321) rides in the GEMM's epilogue (activation code 6): no second pass over the matrix
    S = _score_out(out, a.shape[0], b.shape[0], a.device, "pdist_cos")
    ldS = S.stride(0) if S.shape[0] > 1 else max(S.shape[1], S.stride(0))
    _lib.check(lib.itr_gemm_nt(_p(a), a.shape[1], _p(b), a.shape[1], _p(None), _p(S), ldS, a.shape[0], b.shape[0], a.shape[1], _ACTS['nan_to_zero'],
                               _stream()))
    return S


def split_bf16(x):
    """fp32 [..., K] -> split-bf16 operand [rows, 2 K] (int16 storage): per row and 32-wide chunk 32 hi values then 32 lo values,
    x = hi + lo + O(2^-17 |x|).  K must be a multiple of 32."""
    lib = _lib.load()
    x = _dev(x, name="x")
    K = x.shape[-1]
    rows = x.numel() // K if K else 0
    out = torch.empty(rows, 2 * K, device=x.device, dtype=torch.int16)
    _lib.check(lib.itr_split_bf16(_p(x), _p(out), rows, K, _stream()))
    return out


def gemm_nt_bf16(a_il, b_il, bias=None, terms=3, act=None, M=None, K=None, lda=None, out=None):
    """STUDY / opt-in (STUDY_SPLIT_PRECISION.md): A B^T on the bf16 matrix core from split operands (split_bf16), fp32 accumulation.
    terms = 3: hi.hi + hi.lo + lo.hi; terms = 1: hi.hi.  The product GEMM (linear / cosine_scores) is exact fp32.
    M, K, lda (fp32 elements) describe strided / overlapping rows of a flat operand (linear_strided)."""
    lib = _lib.load()
    if b_il.dim() != 2 or (M is None and (a_il.dim() != 2 or a_il.shape[1] != b_il.shape[1])) or (K is not None and 2 * K != b_il.shape[1]):
        raise ValueError("gemm_nt_bf16: A %s vs B %s" % (tuple(a_il.shape), tuple(b_il.shape)))
    if M is None:
        M, K, lda = a_il.shape[0], a_il.shape[1] // 2, a_il.shape[1] // 2
    N = b_il.shape[0]
    b = _dev(bias, name="bias") if bias is not None else None
    if out is None:
        out = torch.empty(M, N, device=a_il.device, dtype=torch.float32)
    _lib.check(lib.itr_gemm_nt_bf16(_p(a_il), 2 * lda, _p(b_il), b_il.shape[1], _p(b), _p(out), out.stride(0), M, N, K,
                                    _ACTS[act], int(terms), _stream()))
    return out


def split_f16(x):
    """fp32 [..., K] -> (fp16x3 operand [rows, 2 K] int16, scale_state [2] float32 on the device): per row and 32-wide chunk 32 hi then
    32 scaled-lo fp16 values of x * s, s = the power of two that puts the tensor's absmax in [2^14, 2^15)."""
    lib = _lib.load()
    x = _dev(x, name="x")
    K = x.shape[-1]
    rows = x.numel() // K if K else 0
    out = torch.empty(rows, 2 * K, device=x.device, dtype=torch.int16)
    state = torch.empty(2, device=x.device, dtype=torch.float32)
    _lib.check(lib.itr_split_f16(_p(x), _p(out), _p(state), rows, K, _stream()))
    return out, state


def gemm_nt_f16x3(a, b, bias=None, act=None, M=None, K=None, lda=None, out=None):
    """STUDY / opt-in (STUDY_SPLIT_PRECISION.md): A B^T from split_f16 operands on the fp16 matrix core, fp32 accumulation, error at the fp32
    rounding level.  M, K, lda (fp32 elements) describe strided / overlapping rows of a flat operand (linear_strided)."""
    lib = _lib.load()
    (a_il, sa), (b_il, sb) = a, b
    if b_il.dim() != 2 or (M is None and (a_il.dim() != 2 or a_il.shape[1] != b_il.shape[1])) or (K is not None and 2 * K != b_il.shape[1]):
        raise ValueError("gemm_nt_f16x3: A %s vs B %s" % (tuple(a_il.shape), tuple(b_il.shape)))
    if M is None:
        M, K, lda = a_il.shape[0], a_il.shape[1] // 2, a_il.shape[1] // 2
    N = b_il.shape[0]
    bb = _dev(bias, name="bias") if bias is not None else None
    if out is None:
        out = torch.empty(M, N, device=a_il.device, dtype=torch.float32)
    _lib.check(lib.itr_gemm_nt_f16x3(_p(a_il), 2 * lda, _p(sa), _p(b_il), b_il.shape[1], _p(sb), _p(bb), _p(out), out.stride(0), M, N, K,
                                     _ACTS[act], _stream()))
    return out


def order_scores(im, s, out=None):
    """order_sim (Objectives.py:24-30): -sqrt(sum_d max(0, s - im)^2) for every (image, sentence) pair."""
    lib = _lib.load()
    im, s = _dev(im, name="im"), _dev(s, name="s")
    if im.dim() != 2 or s.dim() != 2 or im.shape[1] != s.shape[1]:
        raise ValueError("order_scores: im %s vs s %s" % (tuple(im.shape), tuple(s.shape)))
    res = torch.empty(im.shape[0], s.shape[0], device=im.device, dtype=torch.float32)
    _lib.check(lib.itr_order_scores(_p(im), _p(s), _p(res), im.shape[0], s.shape[0], im.shape[1], _stream()))
    if out is not None:          # (the order kernel writes dense rows: a column block of a wider matrix is filled by a copy)
        _score_out(out, im.shape[0], s.shape[0], im.device, "order_scores").copy_(res)
        return out
    return res


def pdist(x1, x2):
    """SAEM's euclidean `pdist` (Objectives.py:297-307): sqrt(|x1|^2 - 2 x1.x2 + |x2|^2 + 1e-4)."""
    lib = _lib.load()
    x1, x2 = _dev(x1, name="x1"), _dev(x2, name="x2")
    S = cosine_scores(x1, x2)
    n1 = torch.empty(x1.shape[0], device=x1.device, dtype=torch.float32)
    n2 = torch.empty(x2.shape[0], device=x1.device, dtype=torch.float32)
    _lib.check(lib.itr_row_sqnorm(_p(x1), _p(n1), x1.shape[0], x1.shape[1], _stream()))
    _lib.check(lib.itr_row_sqnorm(_p(x2), _p(n2), x2.shape[0], x2.shape[1], _stream()))
    _lib.check(lib.itr_pdist_finish(_p(S), _p(n1), _p(n2), S.shape[0], S.shape[1], _stream()))
    return S


def mvm_scores(imgs, caps, out=None):
    """MultiViewMatching.forward (Fusionmodule.py:674-692)."""
    lib = _lib.load()
    imgs = _dev(imgs, name="imgs")
    caps = _dev(caps, name="caps")
    if imgs.dim() != 3 or caps.dim() != 2 or imgs.shape[2] != caps.shape[1]:
        raise ValueError("mvm_scores: expected (Ni, k, D) and (Nc, D)")
    Ni, k, D = imgs.shape
    S = _score_out(out, Ni, caps.shape[0], imgs.device, "mvm_scores")
    _lib.check(lib.itr_mvm_scores(_p(imgs), _p(caps), _p(S), Ni, caps.shape[0], k, D, S.stride(0) if Ni > 1 else max(S.shape[1], S.stride(0)), _stream()))
    return S


# ------------------------------------------------------------------------------------------
def hinge_fwd(scores, margin, max_violation):
    """-> (loss[1], row_arg, col_arg)."""
    lib = _lib.load()
    scores = _dev(scores, name="scores")
    if scores.dim() != 2 or scores.shape[0] != scores.shape[1]:
        raise ValueError("hinge loss needs a square score matrix, got %s" % (tuple(scores.shape),))
    B = scores.shape[0]
    loss = torch.empty(1, device=scores.device, dtype=torch.float32)
    row_arg = torch.empty(B, device=scores.device, dtype=torch.int32)
    col_arg = torch.empty(B, device=scores.device, dtype=torch.int32)
    ws = torch.empty(2 * B, device=scores.device, dtype=torch.float32)
    _lib.check(lib.itr_hinge_maxviol_fwd(_p(scores), B, B, float(margin), int(bool(max_violation)), _p(loss),
                                         _p(row_arg), _p(col_arg), _p(ws), _stream()))
    return loss, row_arg, col_arg


def hinge_bwd(scores, margin, max_violation, row_arg, col_arg, grad_loss):
    lib = _lib.load()
    scores = _dev(scores, name="scores")
    B = scores.shape[0]
    g = _dev(grad_loss.reshape(1), name="grad_loss")
    dS = torch.empty_like(scores)
    _lib.check(lib.itr_hinge_maxviol_bwd(_p(scores), B, B, float(margin), int(bool(max_violation)), _p(row_arg),
                                         _p(col_arg), _p(g), _p(dS), B, _stream()))
    return dS


class _HingeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, scores, margin, max_violation):
        scores = scores.contiguous()
        loss, ra, ca = hinge_fwd(scores, margin, max_violation)
        ctx.save_for_backward(scores, ra, ca)
        ctx.margin, ctx.mv = margin, max_violation
        return loss.reshape(())

    @staticmethod
    def backward(ctx, grad):
        scores, ra, ca = ctx.saved_tensors
        return hinge_bwd(scores, ctx.margin, ctx.mv, ra, ca, grad.contiguous()), None, None


def hinge_loss(scores, margin=0.2, max_violation=False):
    """Differentiable (w.r.t. scores) bidirectional hinge loss, HIP forward and backward."""
    return _HingeFn.apply(scores, margin, max_violation)


# ------------------------------------------------------------------------------------------
class ScanPlan:
    """Column-tile plan of a caption set for the SCAN kernel (host-side, cheap, reusable).

    The fused kernel packs whole captions into 64-column tiles, so a caption of more than 64 words does not fit
    (Flickr30k has a few, up to 82 tokens).  Such captions are planned separately: the kernel scores the others and
    scan_xattn_scores fills their columns with the one-workgroup-per-pair forward of the training path
    (csrc/scan_train*.hip, up to 96 words).  `Nc` is always the full caption count."""

    def __init__(self, cap_off, cap_len, n_rows, device, max_kernel_len=None):
        lib = _lib.load()
        self.len_host = _host_i32(cap_len)
        self.off_host = np.asarray(cap_off, dtype=np.int64)
        self.Nc = len(self.len_host)
        self.n_rows = int(n_rows)
        self.device = device
        long_mask = self.len_host > (SCAN_NT if max_kernel_len is None else max_kernel_len)      # (SGRAF: 63 words + the global node)
        self.long_idx = np.nonzero(long_mask)[0] if long_mask.any() else None
        if self.long_idx is None:
            k_len, k_off = self.len_host, self.off_host
        else:
            self.short_idx = np.nonzero(~long_mask)[0]
            k_len, k_off = np.ascontiguousarray(self.len_host[self.short_idx]), self.off_host[self.short_idx]
        self.Nc_kernel = len(k_len)
        tb = np.zeros(self.Nc_kernel + 1, dtype=np.int32)
        order = np.zeros(max(self.Nc_kernel, 1), dtype=np.int32)
        nt = C.c_int64(0)
        _lib.check(lib.itr_scan_plan_tiles(k_len.ctypes.data_as(C.c_void_p), self.Nc_kernel, SCAN_NT,
                                           tb.ctypes.data_as(C.c_void_p), order.ctypes.data_as(C.c_void_p),
                                           C.byref(nt)))
        self.n_tiles = int(nt.value)
        self.tile_begin = h2d(tb[:self.n_tiles + 1].copy(), device)
        self.cap_order = h2d(order, device)
        self.cap_len = h2d(k_len.copy(), device)
        self.cap_off = h2d(np.asarray(k_off, dtype=np.int64), device)
        self._k_len = k_len
        self._node_groups = None

    def node_groups(self):
        """SGRAF-SGR's fused graph steps (csrc/sgr_fused.hip): groups of whole captions binned by NODE count (words + the global
        node), at most 64 node rows and 16 captions per group (itr_sgr_plan_node_groups, exact fill).
        -> (group_begin int32[n + 1], group_order int32[Nc_kernel], n) on the device, or None when a caption has more than 63
        words (its graph does not fit one workgroup)."""
        if self._node_groups is None:
            lib = _lib.load()
            if self.Nc_kernel == 0 or int(self._k_len.max()) + 1 > SCAN_NT:
                self._node_groups = False
            else:
                lens = np.ascontiguousarray(self._k_len.astype(np.int32))
                tb = np.zeros(self.Nc_kernel + 1, dtype=np.int32)
                order = np.zeros(self.Nc_kernel, dtype=np.int32)
                nt = C.c_int64(0)
                # SETTINGS.sgr_group_rows = 32 (cross-check): captions of <= 31 words in groups of <= 32 node rows, two workgroups per CU
                small = 32 if SETTINGS.sgr_group_rows == 32 else SCAN_NT
                _lib.check(lib.itr_sgr_plan_node_groups(lens.ctypes.data_as(C.c_void_p), self.Nc_kernel, small, tb.ctypes.data_as(C.c_void_p),
                                                        order.ctypes.data_as(C.c_void_p), C.byref(nt)))
                n = int(nt.value)
                # the bounds the device checks again (a group that breaks them is refused there and its columns come back NaN)
                cnt = np.diff(tb[:n + 1])
                rows = np.add.reduceat(lens[order] + 1, tb[:n]) if n else np.zeros(0, np.int64)
                assert n == 0 or (cnt.min() >= 1 and cnt.max() <= 16 and rows.max() <= SCAN_NT and int(cnt.sum()) == self.Nc_kernel)
                self._node_groups = (h2d(tb[:n + 1].copy(), self.device), h2d(order, self.device), n)
        return self._node_groups or None


def scan_prepare(images, words, plan, cross_attn='t2i'):
    """Per (image block, caption set) precompute for the SCAN kernel: tile-packed words, Gram matrices,
    norms.  Returns the prepared workspace (a uint8 tensor); None when the images do not have the fused kernel's 36 regions
    (scan_xattn_scores takes the pair kernels then, which prepare per block)."""
    lib = _lib.load()
    images = _dev(images, name="images")
    words = _dev(words, name="words")
    Ni, R, D = images.shape
    if R != SCAN_R:
        return None
    wsb = lib.itr_scan_workspace_bytes(Ni, R, words.shape[0], plan.Nc_kernel, plan.n_tiles, D)
    ws = torch.empty(wsb, device=images.device, dtype=torch.uint8)
    if Ni == 0 or plan.Nc_kernel == 0:
        return ws                      # nothing to prepare: the score matrix is empty
    _lib.check(lib.itr_scan_prepare(_p(images), _p(words), _p(plan.cap_off), _p(plan.cap_len), _p(plan.tile_begin),
                                    _p(plan.cap_order), plan.n_tiles, Ni, plan.Nc_kernel, words.shape[0], R, D,
                                    0 if cross_attn == 't2i' else 1, _p(ws), wsb, _stream()))
    return ws


def scan_xattn_scores(images, words, plan, cross_attn='t2i', raw_feature_norm='clipped_l2norm',
                      agg_func='LogSumExp', lambda_lse=6.0, lambda_softmax=9.0, out=None, workspace=None, precision='fp32'):
    """xattn_score_t2i / _i2t (Objectives.py:329-417).  images (Ni, R, D); words (n_rows, D) with the
    caption layout described by `plan` (ScanPlan).  -> (Ni, Nc).  R = 36 (every reference configuration): the fused kernel;
    any other R <= 100: the one-workgroup-per-pair kernels of the training path (_scan_scores_pairwise), same results.
    precision='bf16x3' / 'fp16x3' (opt-in study variants, STUDY_SPLIT_PRECISION.md): the region x word dot products run on the 16-bit
    matrix core from split operands (hi.hi + hi.lo + lo.hi, fp32 accumulation; bf16 planes: ~3e-6, fp16 planes: ~1e-7 of the
    fp32 result, fp16 needs |x| <= 65504); everything else is unchanged."""
    lib = _lib.load()
    if cross_attn not in ('t2i', 'i2t'):
        raise ValueError("unknown first norm type:", raw_feature_norm)  # the reference's message (Objectives.py:71)
    if raw_feature_norm not in _NORMS:
        raise ValueError("unknown first norm type:", raw_feature_norm)
    if agg_func not in _AGGS:
        raise ValueError("unknown aggfunc: {}".format(agg_func))
    images = _dev(images, name="images")
    words = _dev(words, name="words")
    Ni, R, D = images.shape
    n_rows = words.shape[0]
    if out is None:
        out = torch.empty(Ni, plan.Nc, device=images.device, dtype=torch.float32)
    if Ni == 0 or plan.Nc == 0:
        return out
    if R != SCAN_R:
        # the fused kernel is built for the 36 regions of every reference configuration (4 images = 144 rows = 9 MFMA row tiles);
        # any other region count takes the one-workgroup-per-pair kernels of the training path (1..100 regions)
        if precision != 'fp32':
            raise NotImplementedError("scan_xattn_scores: precision=%r with %d regions per image (the fused kernel: %d)" % (precision, R, SCAN_R))
        return _scan_scores_pairwise(images, words, plan, np.arange(plan.Nc), cross_attn, raw_feature_norm, agg_func, lambda_lse,
                                     lambda_softmax, out)
    if plan.long_idx is not None:
        if precision != 'fp32':
            raise NotImplementedError("scan_xattn_scores: precision=%r with captions of more than %d words" % (precision, SCAN_NT))
        return _scan_scores_with_long_captions(images, words, plan, cross_attn, raw_feature_norm, agg_func, lambda_lse, lambda_softmax,
                                               out, workspace)
    ws = workspace if workspace is not None else scan_prepare(images, words, plan, cross_attn)
    if precision in ('bf16x3', 'fp16x3'):
        bsz = lib.itr_scan_bf16_workspace_bytes(Ni, R, plan.n_tiles, D)
        bws = torch.empty(bsz, device=images.device, dtype=torch.uint8)
        _lib.check(lib.itr_scan_xattn_scores_bf16x3(
            _p(images), plan.n_tiles, Ni, plan.Nc, n_rows, R, D, 0 if cross_attn == 't2i' else 1,
            _NORMS[raw_feature_norm], _AGGS[agg_func], float(lambda_softmax), float(lambda_lse), _p(out), out.stride(0),
            _p(ws), ws.numel(), _p(bws), bsz, 1 if precision == 'fp16x3' else 0, _stream()))
        return out
    if precision != 'fp32':
        raise ValueError("scan_xattn_scores: precision must be 'fp32', 'bf16x3' or 'fp16x3'")
    _lib.check(lib.itr_scan_xattn_scores(
        _p(images), plan.n_tiles, Ni, plan.Nc, n_rows, R, D, 0 if cross_attn == 't2i' else 1,
        _NORMS[raw_feature_norm], _AGGS[agg_func], float(lambda_softmax), float(lambda_lse), _p(out), out.stride(0),
        _p(ws), ws.numel(), _stream()))
    return out


def scan_clock_probe(images, words, plan, cross_attn='t2i', raw_feature_norm='clipped_l2norm', agg_func='LogSumExp', lambda_lse=6.0,
                     lambda_softmax=9.0, workspace=None):
    """Shader clock [MHz] the chip sustains UNDER the SCAN kernel: one extra INSTRUMENTED launch (itr_debug_scan_clock_probe; the
    stamps cost the kernel a few percent, so it is never the timed launch).  Every workgroup adds its s_memtime (shader cycles) and
    s_memrealtime (100 MHz) deltas into eight counters at the head of a scratch buffer; clock = 100 * cycles / ticks."""
    lib = _lib.load()
    images = _dev(images, name="images")
    words = _dev(words, name="words")
    Ni, R, D = images.shape
    if R != SCAN_R or plan.long_idx is not None or Ni == 0 or plan.Nc == 0:
        return None
    ws = workspace if workspace is not None else scan_prepare(images, words, plan, cross_attn)
    ld = plan.Nc + 64 + (plan.Nc & 1)                      # an even row width: the int64 view of the head exists whatever Nc is
    scratch = torch.zeros(Ni, ld, device=images.device, dtype=torch.float32)
    _lib.check(lib.itr_debug_scan_clock_probe(
        _p(images), plan.n_tiles, Ni, plan.Nc, words.shape[0], R, D, 0 if cross_attn == 't2i' else 1, _NORMS[raw_feature_norm],
        _AGGS[agg_func], float(lambda_softmax), float(lambda_lse), _p(scratch), ld, _p(ws), ws.numel(), _stream()))
    torch.cuda.synchronize()
    c = scratch.view(torch.int64).flatten()[:8].cpu().numpy().astype(np.float64)
    return 100.0 * float(c[:7].sum()) / float(c[7]) if c[7] > 0 else None


def _scan_scores_with_long_captions(images, words, plan, cross_attn, norm, agg, lambda_lse, lambda_softmax, out, workspace):
    """Captions of 65..96 words (ScanPlan docstring): fused kernel for the rest, pair kernels for these."""
    from . import autograd
    lib = _lib.load()
    Ni, R, D = images.shape
    dev = images.device
    if plan.Nc_kernel:
        ws = workspace if workspace is not None else scan_prepare(images, words, plan, cross_attn)
        part = torch.empty(Ni, plan.Nc_kernel, device=dev, dtype=torch.float32)
        _lib.check(lib.itr_scan_xattn_scores(
            _p(images), plan.n_tiles, Ni, plan.Nc_kernel, words.shape[0], R, D, 0 if cross_attn == 't2i' else 1, _NORMS[norm],
            _AGGS[agg], float(lambda_softmax), float(lambda_lse), _p(part), part.stride(0), _p(ws), ws.numel(), _stream()))
        out[:, torch.from_numpy(plan.short_idx).to(dev)] = part
    return _scan_scores_pairwise(images, words, plan, plan.long_idx, cross_attn, norm, agg, lambda_lse, lambda_softmax, out)


SCAN_PAIR_MAXW, SCAN_PAIR_RMAX = 96, 100          # csrc/scan_train.hip: ST_MAXW, ST_RMAX


def _scan_scores_pairwise(images, words, plan, cap_idx, cross_attn, norm, agg, lambda_lse, lambda_softmax, out, budget_bytes=1 << 30):
    """out[:, cap_idx] through the forward of the training path (csrc/scan_train*.hip: one GEMM for the raw dot products of a block
    of pairs, one workgroup per (image, caption) pair for the rest) -- the captions the fused kernel does not take (65..96 words) and
    every caption when the images do not have the 36 regions it is built for.  The pairs are cut into blocks whose dot-product
    matrix [images x R, words] stays under `budget_bytes`."""
    from . import autograd
    Ni, R, D = images.shape
    dev = images.device
    cap_idx = np.asarray(cap_idx, dtype=np.int64)
    lens_all = plan.len_host[cap_idx]
    if R > SCAN_PAIR_RMAX or (len(lens_all) and int(lens_all.max()) > SCAN_PAIR_MAXW):
        raise NotImplementedError("scan_xattn_scores: %d regions per image / captions of %d words (supported: <= %d regions; "
                                  "captions of <= %d words)" % (R, int(lens_all.max()) if len(lens_all) else 0, SCAN_PAIR_RMAX, SCAN_PAIR_MAXW))
    fn = autograd.scan_t2i_scores if cross_attn == 't2i' else autograd.scan_i2t_scores
    bi = int(min(Ni, 1024, 65535 // R))                               # grid.y and the i2t kernels' 65535-row limits
    tok_budget = max(int(budget_bytes // (4 * bi * R)), SCAN_PAIR_MAXW)
    csum = np.concatenate([[0], np.cumsum(lens_all)])
    with torch.no_grad():
        c0 = 0
        while c0 < len(cap_idx):
            c1 = int(np.searchsorted(csum, csum[c0] + tok_budget, side='right')) - 1
            c1 = min(max(c1, c0 + 1), len(cap_idx))
            ids, lens = cap_idx[c0:c1], lens_all[c0:c1]
            rows = np.concatenate([np.arange(plan.off_host[c], plan.off_host[c] + plan.len_host[c]) for c in ids])
            w_blk = words[h2d(rows, dev)].contiguous()
            off = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.int64)
            cols = h2d(ids, dev)
            for i0 in range(0, Ni, bi):
                out[i0:i0 + bi, cols] = fn(images[i0:i0 + bi], w_blk, off, lens, norm, agg, lambda_lse, lambda_softmax)
            c0 = c1
    return out


def scan_xattn_padded(images, captions, cap_lens, **kw):
    """Reference call shape: captions (Nc, L, D) padded + cap_lens (Objectives.py:329)."""
    Nc, L, D = captions.shape
    lens = [int(x) for x in cap_lens][:Nc]
    plan = ScanPlan(np.arange(Nc, dtype=np.int64) * L, lens, Nc * L, captions.device)
    return scan_xattn_scores(images, _dev(captions, name="captions").reshape(Nc * L, D), plan, **kw)


# ------------------------------------------------------------------------------------------ transformer blocks
def bert_embed_ln(ids, type_ids, word_emb, pos_emb, type_emb, gamma, beta, eps=1e-12):
    lib = _lib.load()
    ids = _dev(ids, torch.int64, "input_ids")
    ty = _dev(type_ids, torch.int64, "token_type_ids") if type_ids is not None else None
    B, L = ids.shape
    H = word_emb.shape[1]
    out = torch.empty(B, L, H, device=ids.device, dtype=torch.float32)
    _lib.check(lib.itr_bert_embed_ln(_p(ids), _p(ty), _p(_dev(word_emb)), _p(_dev(pos_emb)), _p(_dev(type_emb)),
                                     _p(_dev(gamma)), _p(_dev(beta)), _p(out), B, L, H, word_emb.shape[0],
                                     pos_emb.shape[0], type_emb.shape[0], eps, _stream()))
    return out


def add_layernorm(x, residual, gamma, beta, eps=1e-12):
    """BERTLayerNorm(x + residual) (bert.py:113-126): epsilon inside the sqrt."""
    lib = _lib.load()
    x = _dev(x, name="x")
    r = _dev(residual, name="residual") if residual is not None else None
    H = x.shape[-1]
    out = torch.empty_like(x)
    _lib.check(lib.itr_add_layernorm(_p(x), _p(r), _p(_dev(gamma)), _p(_dev(beta)), _p(out), x.numel() // H, H, eps,
                                     _stream()))
    return out


def mha_small(q, k, v, mask, B, L, heads, dk, scale, out=None):
    """q, k, v: 2-D views [B*L, *] (last dim contiguous, head h at columns h*dk .. h*dk+dk-1; they may be column
    slices of one fused QKV buffer).  mask: (B, L) float 0/1 or None.  -> (B*L, heads*dk).  `out`: a (B*L, heads*dk) view with a
    contiguous last dim (a column block of a wider buffer; rows that are not 16-byte aligned take the LDS-staged kernel)."""
    lib = _lib.load()
    for t in (q, k, v):
        if not (t.is_cuda and t.dtype == torch.float32 and t.stride(-1) == 1):
            raise ValueError("mha_small: q/k/v must be fp32 CUDA tensors with a contiguous last dim")
    m = _dev(mask.to(torch.float32), name="mask") if mask is not None else None
    if out is None:
        out = torch.empty(B * L, heads * dk, device=q.device, dtype=torch.float32)
    elif not (out.is_cuda and out.dtype == torch.float32 and out.stride(-1) == 1 and tuple(out.shape) == (B * L, heads * dk)):
        raise ValueError("mha_small: out must be a fp32 CUDA (B*L, heads*dk) view with a contiguous last dim")
    _lib.check(lib.itr_mha_small(_p(q), _p(k), _p(v), q.stride(0), k.stride(0), v.stride(0), _p(m), _p(out),
                                 out.stride(0), B, L, heads, dk, float(scale), _stream()))
    return out


def relu_maxpool(x, valid, out=None):
    """x (B, L, C) -> max over t < valid of relu(x) -> (B, C)."""
    lib = _lib.load()
    x = _dev(x, name="x")
    B, L, Cc = x.shape
    if out is None:
        out = torch.empty(B, Cc, device=x.device, dtype=torch.float32)
    for b0 in range(0, B, 65535):
        b1 = min(B, b0 + 65535)
        _lib.check(lib.itr_relu_maxpool(_p(x[b0:b1]), _p(out[b0:b1]), out.stride(0), b1 - b0, L, Cc, int(valid), _stream()))
    return out


# ------------------------------------------------------------------------------------------ CAMERA helpers
def mul_rows(a, b2d):
    """a (..., C) * b where b is a 2-D (R, C) view with contiguous last dim (may be a column slice)."""
    lib = _lib.load()
    a = _dev(a, name="a")
    Cc = a.shape[-1]
    R = a.numel() // Cc
    if not (b2d.is_cuda and b2d.dim() == 2 and b2d.stride(1) == 1 and b2d.shape[0] == R and b2d.shape[1] == Cc):
        raise ValueError("mul_rows: b must be a (R, C) CUDA view with a contiguous last dim")
    out = torch.empty_like(a)
    _lib.check(lib.itr_mul_rows(_p(a), _p(b2d), b2d.stride(0), _p(out), R, Cc, _stream()))
    return out


def agsa_gate(q2, k2, fc_q, fc_k, fc_g):
    """GatedQueryAttLayer's gate (camera_.py:36-44) on q2, k2 [rows, dk] (rows = positions x heads; dk = 16 or 32), fc_* = (weight,
    bias) of the three Linear layers -> (q2 * M[:, :dk], k2 * M[:, dk:]) with M = sigmoid(fc_g(fc_q(q2) * fc_k(k2))).  One kernel
    (csrc/agsa_gate.hip) instead of three skinny GEMMs and three elementwise products."""
    lib = _lib.load()
    q2, k2 = _dev(q2, name="q"), _dev(k2, name="k")
    rows, dk = q2.shape
    w = [_dev(t.detach(), name="gate weight") for pair in (fc_q, fc_k, fc_g) for t in pair]
    if w[0].shape != (dk, dk) or w[2].shape != (dk, dk) or w[4].shape != (2 * dk, dk) or k2.shape != q2.shape:
        raise ValueError("agsa_gate: q / k [rows, dk], fc_q / fc_k Linear(dk, dk), fc_g Linear(dk, 2 dk)")
    qo, ko = torch.empty_like(q2), torch.empty_like(k2)
    _lib.check(lib.itr_agsa_gate(_p(q2), _p(k2), rows, dk, _p(w[0]), _p(w[1]), _p(w[2]), _p(w[3]), _p(w[4]), _p(w[5]), _p(qo), _p(ko), _stream()))
    return qo, ko


def affine_cols(x, scale=None, shift=None, residual=None, act=None):
    """act(x*scale[c] + shift[c]) + residual (eval-mode BatchNorm1d folded to scale / shift)."""
    lib = _lib.load()
    x = _dev(x, name="x")
    Cc = x.shape[-1]
    out = torch.empty_like(x)
    sc = _dev(scale) if scale is not None else None
    sh = _dev(shift) if shift is not None else None
    rs = _dev(residual) if residual is not None else None
    _lib.check(lib.itr_affine_cols(_p(x), _p(sc), _p(sh), _p(rs), _p(out), x.numel() // Cc, Cc, _ACTS[act], _stream()))
    return out


def gemm_acc(x2d, lda, M, K, weight, bias, out, act=None):
    """out = act(out + A @ weight^T + bias), A rows at x2d.data_ptr() + m*lda."""
    lib = _lib.load()
    weight = _dev(weight, name="weight")
    b = _dev(bias) if bias is not None else None
    _lib.check(lib.itr_gemm_nt_acc(_p(x2d), lda, _p(weight), weight.shape[1], _p(b), _p(out), out.stride(0), M,
                                   weight.shape[0], K, _ACTS[act], _stream()))
    return out


def gemm_residual(x2d, weight, bias, residual, act=None):
    """act(residual + x2d @ weight^T + bias) -> new tensor (a residual connection without copying the residual into the output
    first; same arithmetic as gemm_acc on a clone)."""
    lib = _lib.load()
    x2d, weight, residual = _dev(x2d, name="x"), _dev(weight, name="weight"), _dev(residual, name="residual")
    M, K = x2d.shape
    N = weight.shape[0]
    if weight.shape[1] != K or tuple(residual.shape) != (M, N):
        raise ValueError("gemm_residual: x (%d, %d), weight %s, residual %s" % (M, K, tuple(weight.shape), tuple(residual.shape)))
    b = _dev(bias) if bias is not None else None
    out = torch.empty(M, N, device=x2d.device, dtype=torch.float32)
    _lib.check(lib.itr_gemm_nt_residual(_p(x2d), K, _p(weight), K, _p(b), _p(residual), N, _p(out), N, M, N, K, _ACTS[act], _stream()))
    return out


def gcn_relation(tpg, n_img, n_regions, channels):
    """Rs_GCN between its convolutions (vsrn_.py:50-67): tpg [n_img*N, 3C] rows = theta | phi | g ->
    y [n_img*N, C] = (theta phi^T / N) g per image."""
    lib = _lib.load()
    tpg = _dev(tpg, name="tpg")
    if tpg.dim() != 2 or tpg.shape[0] != n_img * n_regions or tpg.shape[1] != 3 * channels:
        raise ValueError("gcn_relation: tpg %s for %d images x %d regions x 3*%d" % (tuple(tpg.shape), n_img, n_regions, channels))
    y = torch.empty(n_img * n_regions, channels, device=tpg.device, dtype=torch.float32)
    _lib.check(lib.itr_gcn_relation(_p(tpg), tpg.shape[1], _p(y), channels, n_img, n_regions, channels, _stream()))
    return y


def camera_posenc(boxes, imgs_wh):
    lib = _lib.load()
    boxes = _dev(boxes.to(torch.float32), name="boxes")
    wh = _dev(imgs_wh.to(torch.float32), name="imgs_wh")
    B, R = boxes.shape[:2]
    out = torch.empty(B, R, 6, device=boxes.device, dtype=torch.float32)
    _lib.check(lib.itr_camera_posenc(_p(boxes), _p(wh), _p(out), B, R, _stream()))
    return out


def camera_summarize(smry_mat, X):
    """softmax over regions of smry (B, R, k); (L^T X) -> F.normalize -> (B, k, D)."""
    lib = _lib.load()
    smry_mat = _dev(smry_mat, name="smry_mat")
    X = _dev(X, name="X")
    B, R, k = smry_mat.shape
    D = X.shape[2]
    out = torch.empty(B, k, D, device=X.device, dtype=torch.float32)
    _lib.check(lib.itr_camera_summarize(_p(smry_mat), _p(X), _p(out), B, R, k, D, _stream()))
    return out


# ------------------------------------------------------------------------------------------
_SGRAF_MAP = {
    "v_loc_w": "v_global_w.embedding_local.0.weight", "v_loc_b": "v_global_w.embedding_local.0.bias",
    "v_loc_bn_w": "v_global_w.embedding_local.1.weight", "v_loc_bn_b": "v_global_w.embedding_local.1.bias",
    "v_loc_bn_mean": "v_global_w.embedding_local.1.running_mean", "v_loc_bn_var": "v_global_w.embedding_local.1.running_var",
    "v_glo_w": "v_global_w.embedding_global.0.weight", "v_glo_b": "v_global_w.embedding_global.0.bias",
    "v_glo_bn_w": "v_global_w.embedding_global.1.weight", "v_glo_bn_b": "v_global_w.embedding_global.1.bias",
    "v_glo_bn_mean": "v_global_w.embedding_global.1.running_mean", "v_glo_bn_var": "v_global_w.embedding_global.1.running_var",
    "v_com_w": "v_global_w.embedding_common.0.weight", "v_com_b": "v_global_w.embedding_common.0.bias",
    "t_loc_w": "t_global_w.embedding_local.0.weight", "t_loc_b": "t_global_w.embedding_local.0.bias",
    "t_glo_w": "t_global_w.embedding_global.0.weight", "t_glo_b": "t_global_w.embedding_global.0.bias",
    "t_com_w": "t_global_w.embedding_common.0.weight", "t_com_b": "t_global_w.embedding_common.0.bias",
    "loc_w": "sim_tranloc_w.weight", "loc_b": "sim_tranloc_w.bias", "glo_w": "sim_tranglo_w.weight",
    "glo_b": "sim_tranglo_w.bias", "eval_w": "sim_eval_w.weight", "eval_b": "sim_eval_w.bias",
}
_SAF_MAP = {"saf_w": "SAF_module.attn_sim_w.weight", "saf_b": "SAF_module.attn_sim_w.bias",
            "saf_bn_w": "SAF_module.bn.weight", "saf_bn_b": "SAF_module.bn.bias",
            "saf_bn_mean": "SAF_module.bn.running_mean", "saf_bn_var": "SAF_module.bn.running_var"}


SGRAF_MAX_WORDS = 63      # fused pair kernels: 63 words + the global node = one 64-row tile
SGRAF_COMPOSED_MAX_WORDS = 191   # the per-caption composition: 191 words + the global node = the 192 graph nodes itr_smry_fwd holds


SGRAF_FLAGS = {"unfused_steps": 1, "non_persistent": 2}       # include/itr_hip.h: ITR_SGRAF_UNFUSED_STEPS, ITR_SGRAF_NON_PERSISTENT
SGRAF_LAST_BLOCK = {}          # diagnostics: what the last sgraf_scores call ran with (image_block, workspace_bytes, budget)


def _sgraf_workspace(lib, dev, Ni, Nc, n_rows, n_tiles, D, S_dim, mod, flags, image_block, max_workspace_bytes):
    """Pick the pair stage's image block for the memory this process can have and allocate the workspace (a
    validation pass inside a training process -- optimizer state and activations resident -- or a co-tenant must shrink the
    block, not die).  Budget = `max_workspace_bytes` if given, else 90 % of (free device memory + what torch's caching allocator
    holds unused); the largest of 64 / 32 / 16 / 8 / 4 images whose workspace fits is taken (itr_sgraf_pick_image_block), and an
    allocation that still fails falls back to the next smaller block.  image_block / SETTINGS.sgraf_image_block pin the block instead.
    -> (workspace tensor, image_block)."""
    if image_block is None and SETTINGS.sgraf_image_block:
        image_block = int(SETTINGS.sgraf_image_block)          # explicit knob (several ranks sharing ONE GPU in the tests)
    if image_block is not None:
        wsb = lib.itr_sgraf_workspace_bytes(Ni, Nc, n_rows, n_tiles, D, S_dim, mod, int(image_block), flags)
        SGRAF_LAST_BLOCK.update(image_block=int(image_block), workspace_bytes=int(wsb), budget_bytes=None, pinned=True)
        return torch.empty(wsb, device=dev, dtype=torch.uint8), int(image_block)
    if max_workspace_bytes is None:
        free, _total = torch.cuda.mem_get_info(dev)
        cached = torch.cuda.memory_reserved(dev) - torch.cuda.memory_allocated(dev)
        budget = int(0.9 * (free + cached))
    else:
        budget = int(max_workspace_bytes)
    while True:
        wsb = C.c_size_t(0)
        ib = lib.itr_sgraf_pick_image_block(Ni, Nc, n_rows, n_tiles, D, S_dim, mod, flags, max(0, budget), C.byref(wsb))
        if ib < 0:
            raise torch.cuda.OutOfMemoryError("sgraf_scores: %s" % lib.itr_last_error().decode())
        try:
            ws = torch.empty(wsb.value, device=dev, dtype=torch.uint8)
        except torch.cuda.OutOfMemoryError:
            if max_workspace_bytes is not None and budget < wsb.value:
                raise
            torch.cuda.empty_cache()
            budget = wsb.value - 1                             # the next smaller block
            continue
        SGRAF_LAST_BLOCK.update(image_block=int(ib), workspace_bytes=int(wsb.value), budget_bytes=budget, pinned=False)
        return ws, int(ib)


def sgraf_scores(images, words, plan, weights, module_name='SAF', sgr_step=3, out=None, image_block=None, max_workspace_bytes=None,
                 variant=()):
    """EncoderSimilarity.forward (Fusionmodule.py:406-451), eval mode.  images (Ni, 36, D); words (n_rows, D)
    with the caption layout of `plan` (ScanPlan on the WORD lengths); weights: the module's state_dict.
    image_block / max_workspace_bytes: see _sgraf_workspace (default: the largest block the free memory admits).
    variant: names from SGRAF_FLAGS -- the step-by-step / non-persistent forms the fused kernel is cross-checked against."""
    lib = _lib.load()
    if module_name not in ('SAF', 'SGR'):
        raise ValueError('Invalid input of config.module_name in configs.py')
    images = _dev(images, name="images")
    words = _dev(words, name="words")
    Ni, R, D = images.shape
    keep = []
    st = _lib.SgrafWeights()

    def ptr(name):
        t = _dev(weights[name].detach(), name=name)
        keep.append(t)
        return t.data_ptr()

    for f, name in _SGRAF_MAP.items():
        setattr(st, f, ptr(name))
    S_dim = weights["sim_tranloc_w.weight"].shape[0]
    if module_name == 'SAF':
        for f, name in _SAF_MAP.items():
            setattr(st, f, ptr(name))
    else:
        for k in range(sgr_step):
            pre = "SGR_module.sgr%d." % k
            st.sgr_q_w[k] = ptr(pre + "graph_query_w.weight"); st.sgr_q_b[k] = ptr(pre + "graph_query_w.bias")
            st.sgr_k_w[k] = ptr(pre + "graph_key_w.weight"); st.sgr_k_b[k] = ptr(pre + "graph_key_w.bias")
            st.sgr_g_w[k] = ptr(pre + "sim_graph_w.weight"); st.sgr_g_b[k] = ptr(pre + "sim_graph_w.bias")
    mod = 0 if module_name == 'SAF' else 1
    flags = 0
    for v in ((variant,) if isinstance(variant, str) else variant):
        flags |= SGRAF_FLAGS[v]
    dev = images.device
    if out is None:
        out = torch.empty(Ni, plan.Nc, device=dev, dtype=torch.float32)
    if plan.Nc and int(plan.len_host.max()) > SGRAF_MAX_WORDS:
        plan = ScanPlan(plan.off_host, plan.len_host, plan.n_rows, plan.device, max_kernel_len=SGRAF_MAX_WORDS)

    def fused_call(Nc_k, max_len, dst):
        grp = plan.node_groups() if mod == 1 else None      # SGR: the graph steps of a group of captions run in one workgroup
        fl = flags | (SGRAF_FLAGS["unfused_steps"] if (mod == 1 and not grp) else 0)
        ws, ib = _sgraf_workspace(lib, dev, Ni, Nc_k, words.shape[0], plan.n_tiles, D, S_dim, mod, fl, image_block, max_workspace_bytes)
        _lib.check(lib.itr_sgraf_scores(_p(images), _p(words), _p(plan.cap_off), _p(plan.cap_len), _p(plan.tile_begin),
                                        _p(plan.cap_order), plan.n_tiles, Ni, Nc_k, words.shape[0], max_len, R, D, S_dim,
                                        mod, int(sgr_step), C.byref(st), _p(grp[0]) if grp else None, _p(grp[1]) if grp else None,
                                        grp[2] if grp else 0, ib, fl, _p(dst), dst.stride(0), _p(ws), ws.numel(), _stream()))

    if plan.long_idx is not None:
        # Captions of more than 63 words (Flickr30k has a few, up to 82 tokens) do not fit the 64-node tiles of the fused pair
        # kernels: the others are scored by the fused path, these by the per-caption composition of the training path run in
        # evaluation mode (Fusionmodule.encoder_similarity_train(training=False): same arithmetic, HIP kernels, up to 191 words).
        from .modalmodule import Fusionmodule
        if plan.Nc_kernel:
            part = torch.empty(Ni, plan.Nc_kernel, device=dev, dtype=torch.float32)
            fused_call(plan.Nc_kernel, int(plan.len_host[plan.short_idx].max()), part)
            out[:, torch.from_numpy(plan.short_idx).to(dev)] = part
        sim_enc = Fusionmodule.EncoderSimilarity(D, S_dim, module_name, sgr_step)
        own = sim_enc.state_dict()
        sim_enc.load_state_dict({k: (weights[k].detach() if k in weights else own[k]) for k in own if k in weights or k.endswith('num_batches_tracked')})
        sim_enc.to(dev).eval()
        lens = [int(plan.len_host[c]) for c in plan.long_idx]
        if max(lens) > SGRAF_COMPOSED_MAX_WORDS:
            raise NotImplementedError("sgraf_scores: captions of at most %d words are supported" % SGRAF_COMPOSED_MAX_WORDS)
        rows = np.concatenate([np.arange(plan.off_host[c], plan.off_host[c] + plan.len_host[c]) for c in plan.long_idx])
        w_long = words[torch.from_numpy(rows).to(dev)].contiguous()
        off = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.int64)
        with torch.no_grad():
            out[:, torch.from_numpy(plan.long_idx).to(dev)] = Fusionmodule.encoder_similarity_train(sim_enc, images, w_long, off, lens, None,
                                                                                                   training=False)
        return out
    fused_call(plan.Nc, int(plan.len_host.max()) if plan.Nc else 1, out)
    return out


def sgraf_padded(images, captions, cap_lens, weights, module_name='SAF', sgr_step=3, **kw):
    """Reference call shape: captions (Nc, L, D) padded + cap_lens (Fusionmodule.py:406).  kw: sgraf_scores' image_block /
    max_workspace_bytes / variant."""
    Nc, L, D = captions.shape
    lens = [int(x) for x in cap_lens][:Nc]
    plan = ScanPlan(np.arange(Nc, dtype=np.int64) * L, lens, Nc * L, captions.device)
    return sgraf_scores(images, _dev(captions, name="captions").reshape(Nc * L, D), plan, weights, module_name, sgr_step, **kw)


_TOKENS_CHECKED = {}         # id(tensor) -> (weak reference, (version counter, vocabulary size)) of token tensors whose id range has been checked


def gru_encode(tokens_packed, tok_off, lengths, weights, bidirectional, no_txtnorm=False, use_abs=False,
               gather_last=False, out=None, batch_invariant=False, launch_form=None, chains=0):
    """EncoderText.forward on packed captions (TextEncoder.py:38-70).
    tokens_packed (n_tok,) int64 cuda; tok_off (B,) int64; lengths: host list sorted descending.
    weights: dict with the reference's state_dict names.  -> (n_tok, D) packed word embeddings, or
    (B, D) when gather_last.  `out`: the caller's contiguous (n_tok, D) / (B, D) result buffer -- e.g. the head of the
    send buffer of the sharded evaluation's all-gather (evalpipe.py).  batch_invariant: a caption's embedding does not
    depend, bit for bit, on which other captions share its batch (ITR_GRU_BATCH_INVARIANT, include/itr_hip.h).
    launch_form: None (default), "paired" or "input_after_fork" -- the bi-GRU launch orders kept as bit-identical cross-checks
    (ITR_GRU_PAIRED_DIRECTIONS / ITR_GRU_INPUT_AFTER_FORK flag bits); "per_token": the input projection as one GEMM row per TOKEN
    even where the call would project the vocabulary once (ITR_GRU_PER_TOKEN_INPUT; >= 2 tokens per vocabulary word) -- the
    cross-check of that form; "first_step_gemm": the recurrence GEMM of step 0 is launched although h = 0 makes it b_hh
    (ITR_GRU_FIRST_STEP_GEMM), the cross-check of skipping it.  chains: interleaved caption chains of the last-state recurrence (ITR_GRU_CHAINS; 0 = the library's choice)."""
    lib = _lib.load()
    form_bits = {None: 0, "paired": 4, "input_after_fork": 8, "per_token": 16, "paired+per_token": 20, "first_step_gemm": 256,
                 "per_token+first_step_gemm": 272}[launch_form]
    tokens_packed = _dev(tokens_packed, torch.int64, "tokens")
    len_host = _host_i32(lengths)
    B = len(len_host)
    n_tok = int(tokens_packed.numel())
    dev = tokens_packed.device
    tok_off = _dev(tok_off, torch.int64, "tok_off")
    len_dev = h2d(len_host.copy(), dev)
    emb = _dev(weights['embed.weight'], name="embed.weight")
    V, E = emb.shape
    if n_tok:
        # nn.Embedding raises on ids outside [0, V) (TextEncoder.py:41).  The check costs a device -> host round trip, so a token tensor
        # that has been checked is not checked again until it is written to (an evaluation loop encodes the same tokens every step)
        # (The cache trusts torch's version counter: writes through .data, raw pointers or DLPack do not bump it -- such a caller must
        # pass a fresh tensor.  Inference-mode tensors have no version counter: they are checked every time.)
        ver = None if tokens_packed.is_inference() else tokens_packed._version
        ent = _TOKENS_CHECKED.get(id(tokens_packed)) if ver is not None else None
        if not (ent is not None and ent[0]() is tokens_packed and ent[1] == (ver, int(V))):
            lo, hi = torch.aminmax(tokens_packed)
            if int(lo) < 0 or int(hi) >= V:
                raise IndexError("index out of range in self")
            if ver is not None:
                if len(_TOKENS_CHECKED) >= 16:
                    _TOKENS_CHECKED.clear()
                _TOKENS_CHECKED[id(tokens_packed)] = (weakref.ref(tokens_packed), (ver, int(V)))
    w_ih = _dev(weights['rnn.weight_ih_l0'])
    w_hh = _dev(weights['rnn.weight_hh_l0'])
    b_ih = _dev(weights['rnn.bias_ih_l0'])
    b_hh = _dev(weights['rnn.bias_hh_l0'])
    D = w_hh.shape[1]
    rev = [None] * 4
    if bidirectional:
        rev = [_dev(weights['rnn.weight_ih_l0_reverse']), _dev(weights['rnn.weight_hh_l0_reverse']),
               _dev(weights['rnn.bias_ih_l0_reverse']), _dev(weights['rnn.bias_hh_l0_reverse'])]
    wsb = lib.itr_gru_workspace_bytes(n_tok, B, E, D, int(bool(bidirectional)))
    ws = torch.empty(wsb, device=dev, dtype=torch.uint8)
    if out is not None and not (out.is_cuda and out.dtype == torch.float32 and out.is_contiguous()
                                and tuple(out.shape) == ((B, D) if gather_last else (n_tok, D))):
        raise ValueError("gru_encode: out must be a contiguous fp32 CUDA tensor of shape %s" % (((B, D) if gather_last else (n_tok, D)),))
    res = out
    out = None if gather_last else (res if res is not None else torch.empty(n_tok, D, device=dev, dtype=torch.float32))
    out_last = (res if res is not None else torch.empty(B, D, device=dev, dtype=torch.float32)) if gather_last else None
    _lib.check(lib.itr_gru_fwd(_p(tokens_packed), _p(tok_off), _p(len_dev), len_host.ctypes.data_as(C.c_void_p), B,
                               n_tok, _p(emb), V, E, D, _p(w_ih), _p(w_hh), _p(b_ih), _p(b_hh), _p(rev[0]),
                               _p(rev[1]), _p(rev[2]), _p(rev[3]), int(no_txtnorm), int(use_abs), int(bool(gather_last)) | (2 if batch_invariant else 0) | form_bits | ((int(chains) & 7) << 5),
                               _p(out), _p(out_last), _p(ws), wsb, _stream()))
    return out_last if gather_last else out


# ------------------------------------------------------------------------------------------
def rank_counts(S, im_div=5, row0=0, s_gt=None, t2i_rank=None, t2i_best=None):
    """Sort-free ranks of a (local row block of a) similarity matrix, ONE pass over S for both directions.
    -> (i2t_rank int32[n_rows], i2t_top1 int32[n_rows], t2i_rank int32[Nc], t2i_best uint64-as-int64[Nc], s_gt)
    For a single GPU (row0 = 0, all rows local) t2i_rank is final and t2i_top1 = t2i_best & 0xffffffff.
    s_gt None: when the block holds every ground-truth row (the single-GPU call) the library reads the GT scores from S itself
    (no gather launch; the returned s_gt is then None); otherwise they are gathered here for the local rows (the sharded
    evaluation gathers, max-reduces over ranks and passes them in).  t2i_rank / t2i_best None: allocated here and zeroed INSIDE
    the call (ITR_RANK_INIT_COLUMNS); pass them to accumulate over several row blocks.
    A float64 matrix (the reference's cal_sims output, an ensemble average) is ranked in float64: `rank_counts_f64`."""
    lib = _lib.load()
    S = _dev(S, name="S")
    n_rows, Nc = S.shape
    dev = S.device
    if s_gt is None and not (row0 == 0 and n_rows * im_div >= Nc):
        s_gt = torch.full((Nc,), float('-inf'), device=dev, dtype=torch.float32)
        _lib.check(lib.itr_rank_gather_gt(_p(S), S.stride(0), row0, n_rows, Nc, im_div, _p(s_gt), _stream()))
    i2t_rank = torch.empty(n_rows, device=dev, dtype=torch.int32)
    i2t_top1 = torch.empty(n_rows, device=dev, dtype=torch.int32)
    if (t2i_rank is None) != (t2i_best is None):
        raise ValueError("rank_counts: pass both t2i_rank and t2i_best (accumulators of several row blocks) or neither")
    flags = 0
    if t2i_rank is None:
        t2i_rank = torch.empty(Nc, device=dev, dtype=torch.int32)
        t2i_best = torch.empty(Nc, device=dev, dtype=torch.int64)
        flags = 1                      # ITR_RANK_INIT_COLUMNS
    wsb = lib.itr_rank_workspace_bytes(n_rows, Nc)
    ws = torch.empty(wsb // 8 + 1, device=dev, dtype=torch.int64)
    _lib.check(lib.itr_rank_counts(_p(S), S.stride(0), row0, n_rows, Nc, im_div, _p(s_gt), _p(i2t_rank), _p(i2t_top1),
                                   _p(t2i_rank), _p(t2i_best), flags, _p(ws), wsb, _stream()))
    return i2t_rank, i2t_top1, t2i_rank, t2i_best, s_gt


def rank_counts_f64(S, im_div=5):
    """The same counts on a float64 similarity matrix (all rows local): index-exact against the reference's argsort of
    float64 rows / columns (evaluation.py:169, :209; ensemble averages: :380, :398).
    -> (i2t_rank, i2t_top1, t2i_rank, t2i_top1) int32."""
    lib = _lib.load()
    S = _dev(S, dtype=torch.float64, name="S")
    n_rows, Nc = S.shape
    dev = S.device
    s_gt = torch.full((Nc,), float('-inf'), device=dev, dtype=torch.float64)
    _lib.check(lib.itr_rank_gather_gt_f64(_p(S), S.stride(0), 0, n_rows, Nc, im_div, _p(s_gt), _stream()))
    i2t_rank = torch.empty(n_rows, device=dev, dtype=torch.int32)
    i2t_top1 = torch.empty(n_rows, device=dev, dtype=torch.int32)
    t2i_rank = torch.zeros(Nc, device=dev, dtype=torch.int32)
    t2i_key = torch.zeros(Nc, device=dev, dtype=torch.int64)
    t2i_top1 = torch.full((Nc,), -1, device=dev, dtype=torch.int32)
    _lib.check(lib.itr_rank_counts_f64(_p(S), S.stride(0), 0, n_rows, Nc, im_div, _p(s_gt), _p(i2t_rank), _p(i2t_top1),
                                       _p(t2i_rank), _p(t2i_key), _stream()))
    _lib.check(lib.itr_rank_t2i_top1_f64(_p(S), S.stride(0), 0, n_rows, Nc, _p(t2i_key), _p(t2i_top1), _stream()))
    return i2t_rank, i2t_top1, t2i_rank, t2i_top1


def gather_gt(S, im_div=5, row0=0, s_gt=None):
    lib = _lib.load()
    S = _dev(S, name="S")
    n_rows, Nc = S.shape
    if s_gt is None:
        s_gt = torch.full((Nc,), float('-inf'), device=S.device, dtype=torch.float32)
    _lib.check(lib.itr_rank_gather_gt(_p(S), S.stride(0), row0, n_rows, Nc, im_div, _p(s_gt), _stream()))
    return s_gt


def recall_from_ranks(ranks):
    """(r1, r5, r10, medr, meanr) from 0-based integer ranks (evaluation.py:181-185); host side."""
    lib = _lib.load()
    r = np.ascontiguousarray(np.asarray(ranks, dtype=np.int32))
    out = (C.c_double * 5)()
    _lib.check(lib.itr_recall_from_ranks(r.ctypes.data_as(C.c_void_p), len(r), out))
    return tuple(float(v) for v in out)
