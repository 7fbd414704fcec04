"""Differentiable wrappers of the HIP kernels for `model.train_emb` (itr/modalmodule/Models.py:115-145, :198-225).

torch.autograd is used as the tape only: every forward and every backward below is a call into libitr_hip.so
(MFMA GEMMs on transposed operands + the small kernels of csrc/train.hip, gru_train.hip, scan_train.hip).
There is no CPU path: tensors must live on the GPU.

    linear(x, W, b)              y = x W^T + b                          dW = dy^T x, db = colsum(dy), dx = dy W
    l2norm_rows(x)               z = x / (||x|| + eps)                  utils.py:10-15
    gru_sequence(...)            packed (bi)GRU, raw outputs            TextEncoder.py:38-56
    gather_rows(x, idx)          rows of x                              last valid GRU step (TextEncoder.py:57-60)
    cosine_scores(im, s)         im s^T                                 Objectives.py:18-21
    scan_t2i_scores(...)         SCAN t2i similarity matrix             Objectives.py:329-372
    Adam / clip_grad_norm        torch.optim.Adam + clip_grad_norm_     Models.py:98, :178, :223-225
    dp_gather_rows(...)          rank-major all-gather of row blocks    data-parallel train_emb
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from .ops import _dev, _p, _stream, _host_i32, _NORMS, _AGGS, h2d


def _f32(*shape, dev):
    return torch.empty(*shape, device=dev, dtype=torch.float32)


def transpose2d(x):
    lib = _lib.load()
    x = _dev(x, name="x")
    rows, cols = x.shape
    out = _f32(cols, rows, dev=x.device)
    _lib.check(lib.itr_transpose2d(_p(x), _p(out), rows, cols, _stream()))
    return out


def colsum(x, out=None, accumulate=False):
    lib = _lib.load()
    x = _dev(x, name="x")
    rows, cols = x.shape
    if out is None:
        out = _f32(cols, dev=x.device)
    wsb = lib.itr_colsum_workspace_bytes(rows, cols)
    ws = torch.empty(wsb, device=x.device, dtype=torch.uint8)
    _lib.check(lib.itr_colsum(_p(x), _p(out), rows, cols, int(accumulate), _p(ws), wsb, _stream()))
    return out


def _peel_plan(M, N, K):
    """A dense product of the tape whose 128 x 128 output tiles are a round and a bit of the chip: which part to give to a launch
    of its own.  The tile kernel keeps at most 2 x 256 workgroups resident (two per CU share the matrix pipe), so 576 tiles -- VSRN's
    4 608 x 2 048 layers -- take as long as 1 024 (398 us for a 246 us product: round 6).  Peeling whole rows or columns of tiles so that
    the main launch is exactly one or two per CU, and cutting the few peeled tiles along K so that THEY fill the chip
    (itr_gemm_nt_splitk), costs a launch and a slice sum.  Returns None, ('rows', main_rows) or ('cols', main_cols); a model in units
    of one tile's time on a CU of its own.  Training tape only: an element's summation order then depends on the shape of the call."""
    tm, tn = (M + 127) // 128, (N + 127) // 128
    tiles = tm * tn
    if tiles <= 256 or tiles >= 1024 or K < 512:       # (from 1 024 tiles the streaming kernel takes the product)
        return None

    def cost(t):
        return 0.0 if t == 0 else (1.0 if t <= 256 else 2.0 * ((t + 511) // 512))

    def tail_cost(t):
        if t >= 128:
            return cost(t) + 0.1
        s = min(8, 512 // t)
        while s > 1 and K // s < 128:
            s -= 1
        w = t * s
        return (1.0 if w <= 256 else 2.0 * ((w + 511) // 512)) / s + 0.15

    best, plan = cost(tiles) - 0.3, None
    for r in range(1, tm):
        c = cost((tm - r) * tn) + tail_cost(r * tn)
        if c < best:
            best, plan = c, ('rows', (tm - r) * 128)
    for q in range(1, tn):
        c = cost(tm * (tn - q)) + tail_cost(tm * q)
        if c < best:
            best, plan = c, ('cols', (tn - q) * 128)
    return plan


def _nt_call(lib, a_ptr, w_ptr, b_ptr, o_ptr, ldc, M, N, K, dev, sliced):
    """One launch: C [M, N] (row stride ldc) = A [M, K] . W [N, K]^T + b; sliced: cut K so that few tiles fill the chip."""
    vp = C.c_void_p
    if sliced:
        wsb = lib.itr_gemm_nt_splitk_workspace_bytes(M, N, K)
        ws = torch.empty(max(wsb, 1), device=dev, dtype=torch.uint8)
        _lib.check(lib.itr_gemm_nt_splitk(vp(a_ptr), K, vp(w_ptr), K, vp(b_ptr), vp(o_ptr), ldc, M, N, K, 0, _p(ws), wsb, _stream()))
    else:
        _lib.check(lib.itr_gemm_nt(vp(a_ptr), K, vp(w_ptr), K, vp(b_ptr), vp(o_ptr), ldc, M, N, K, 0, _stream()))


def _nt_tape(lib, a, w, b, out):
    """out [M, N] = a [M, K] . w [N, K]^T + b for the training tape (contiguous operands), by the shape of the product: <= 128 rows ->
    16-column strips (algo 4); few output tiles -> slices of K; a round and a bit of tiles -> peeled (_peel_plan); else the plain call.
    (The evaluation keeps ONE kernel family per product so that a row's result does not depend on M.)"""
    M, K = a.shape
    N = w.shape[0]
    if M == 0 or N == 0:
        return out
    pa, pw, po = a.data_ptr(), w.data_ptr(), out.data_ptr()
    pb = b.data_ptr() if b is not None else 0
    if M <= 128 and K >= 256 and 64 <= N < 2048:      # a decoder step's layers: K slices of 16-column strips + one slice sum (the library
        _nt_call(lib, pa, pw, pb, po, N, M, N, K, a.device, True)      # routes <= 128 rows of itr_gemm_nt_splitk there)
    elif M <= 128:
        _lib.check(lib.itr_gemm_nt_algo(_p(a), K, _p(w), K, _p(b), _p(out), N, M, N, K, 0, 4, _stream()))
    elif K >= 512 and ((M + 127) // 128) * ((N + 127) // 128) < 128:      # CAMERA's convolutions as GEMMs: 36 tiles over K = 10 240;
        _nt_call(lib, pa, pw, pb, po, N, M, N, K, a.device, True)         # BERT's 768-wide layers at batch 64: 96 tiles
    else:
        plan = _peel_plan(M, N, K)
        if plan is None:
            _nt_call(lib, pa, pw, pb, po, N, M, N, K, a.device, False)
        elif plan[0] == 'rows':
            m0 = plan[1]
            _nt_call(lib, pa, pw, pb, po, N, m0, N, K, a.device, False)
            _nt_call(lib, pa + 4 * m0 * K, pw, pb, po + 4 * m0 * N, N, M - m0, N, K, a.device, True)
        else:
            n0 = plan[1]
            _nt_call(lib, pa, pw, pb, po, N, M, n0, K, a.device, False)
            _nt_call(lib, pa, pw + 4 * n0 * K, pb + 4 * n0 if pb else 0, po + 4 * n0, N, M, N - n0, K, a.device, True)
    return out


def _gemm_nt(a, b, out=None, accumulate=False):
    """a [M, K] . b [N, K]^T -> [M, N] on the MFMA GEMM."""
    lib = _lib.load()
    M, K = a.shape
    N = b.shape[0]
    assert b.shape[1] == K
    if out is None:
        out = _f32(M, N, dev=a.device)
    if not accumulate:
        if a.is_contiguous() and b.is_contiguous() and out.is_contiguous():
            return _nt_tape(lib, a, b, None, out)
        _lib.check(lib.itr_gemm_nt(_p(a), K, _p(b), K, _p(None), _p(out), N, M, N, K, 0, _stream()))
        return out
    _lib.check(lib.itr_gemm_nt_acc(_p(a), K, _p(b), K, _p(None), _p(out), N, M, N, K, 0, _stream()))
    return out


def _gemm_tn(a, b, out=None, accumulate=False, colsum_a=None):
    """a [R, P]^T . b [R, Q] -> [P, Q]: the weight gradient dY^T X, reduced over the rows in slices (csrc/gemm_tn.hip); colsum_a [P]
    (optional) receives the column sums of a -- the bias gradient -- from the same pass."""
    lib = _lib.load()
    R, P = a.shape
    Q = b.shape[1]
    assert b.shape[0] == R
    if out is None:
        out = _f32(P, Q, dev=a.device)
    wsb = lib.itr_gemm_tn_workspace_bytes(R, P, Q)
    ws = torch.empty(max(wsb, 1), device=a.device, dtype=torch.uint8)
    _lib.check(lib.itr_gemm_tn(_p(a), a.stride(0), _p(b), b.stride(0), _p(out), out.stride(0), R, P, Q, int(accumulate), _p(colsum_a), _p(ws), wsb, _stream()))
    return out


class _Linear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        lib = _lib.load()
        x2 = _dev(x, name="x").reshape(-1, x.shape[-1])
        w = _dev(weight, name="weight")
        b = _dev(bias, name="bias") if bias is not None else None
        M, K = x2.shape
        N = w.shape[0]
        if w.dim() != 2 or w.shape[1] != K:
            raise ValueError("linear: x (..., %d) vs weight %s" % (K, tuple(w.shape)))
        out = _f32(M, N, dev=x.device)
        _nt_tape(lib, x2.contiguous(), w.contiguous(), b, out)
        ctx.save_for_backward(x2, w)
        ctx.has_bias = bias is not None
        ctx.xshape = x.shape
        return out.reshape(x.shape[:-1] + (N,))

    @staticmethod
    def backward(ctx, dy):
        x2, w = ctx.saved_tensors
        dy2 = dy.contiguous().reshape(-1, dy.shape[-1])
        dx = dw = db = None
        want_db = ctx.has_bias and ctx.needs_input_grad[2]
        M, N = dy2.shape
        if N % 4 != 0 and M % 4 == 0 and M >= 512 and N >= 512 and x2.shape[1] % 4 == 0:
            # An output width that is not a multiple of 4 floats (the caption decoder's vocabulary projection, 7 552 x 9 487): dy's rows do
            # not start on 16 bytes, and both gradient products -- which reduce over or stream along N -- would run on the scalar-load
            # kernels (2.2 + 1.5 ms per VSRN step).  With dy^T [N, M] (one transpose, rows of M floats) both are aligned again:
            #   dx [M, K] = (dy^T)^T W  = the TN product over the N rows of dy^T and W;   dW [N, K] = dy^T . (x^T [K, M])^T.
            dyT = transpose2d(dy2)
            if ctx.needs_input_grad[0]:
                dx = _gemm_tn(dyT, w).reshape(ctx.xshape)
            if ctx.needs_input_grad[1]:
                dw = _gemm_nt(dyT, transpose2d(x2))
            if want_db:
                db = colsum(dy2)
            return dx, dw, db
        if ctx.needs_input_grad[0]:
            dx = _gemm_nt(dy2, transpose2d(w)).reshape(ctx.xshape)        # dy [M, N] . (W^T [K, N])^T
        if ctx.needs_input_grad[1]:
            db = _f32(dy2.shape[1], dev=dy2.device) if want_db else None
            dw = _gemm_tn(dy2, x2, colsum_a=db)                           # dy^T [N, M] . x [M, K], split over the rows; db = colsum(dy) rides along
        elif want_db:
            db = colsum(dy2)
        return dx, dw, db


def linear(x, weight, bias=None):
    return _Linear.apply(x, weight, bias)


class _L2Norm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, eps):
        lib = _lib.load()
        x2 = _dev(x, name="x").reshape(-1, x.shape[-1])
        rows, dim = x2.shape
        z = torch.empty_like(x2)
        n = _f32(rows, dev=x.device)
        _lib.check(lib.itr_l2norm_fwd_save(_p(x2), _p(z), _p(n), rows, dim, float(eps), _stream()))
        ctx.save_for_backward(z, n)
        ctx.eps = float(eps)
        return z.reshape(x.shape)

    @staticmethod
    def backward(ctx, dz):
        lib = _lib.load()
        z, n = ctx.saved_tensors
        dz2 = dz.contiguous().reshape(z.shape)
        dx = torch.empty_like(z)
        _lib.check(lib.itr_l2norm_bwd(_p(dz2), _p(z), _p(n), _p(dx), z.shape[0], z.shape[1], ctx.eps, _stream()))
        return dx.reshape(dz.shape), None


def l2norm_rows(x, eps=1e-8):
    """utils.l2norm over the last axis, differentiable."""
    return _L2Norm.apply(x, eps)


class _GatherRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, idx):
        lib = _lib.load()
        x = _dev(x, name="x")
        idx = _dev(idx, torch.int64, "idx")
        out = _f32(idx.numel(), x.shape[1], dev=x.device)
        out.zero_()
        # gather == scatter-add into a zero buffer with the roles swapped is not expressible; use the embedding path:
        # out[r] = x[idx[r]]  is the forward of an embedding whose table is x
        bad = torch.zeros(1, device=x.device, dtype=torch.int32)
        _lib.check(lib.itr_gather_rows(_p(idx), idx.numel(), _p(x), x.shape[0], x.shape[1], _p(out), _p(bad), _stream()))
        ctx.save_for_backward(idx)
        ctx.n_rows = x.shape[0]
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        (idx,) = ctx.saved_tensors
        dout = dout.contiguous()
        dx = torch.zeros(ctx.n_rows, dout.shape[1], device=dout.device, dtype=torch.float32)
        _lib.check(lib.itr_embed_scatter_add(_p(idx), _p(dout), idx.numel(), ctx.n_rows, dout.shape[1], _p(dx), _stream()))
        return dx, None


def gather_rows(x, idx):
    return _GatherRows.apply(x, idx)


class _Gru(torch.autograd.Function):
    """inputs: tokens / tok_off / len (non-differentiable), then embed, w_ih, w_hh, b_ih, b_hh[, reverse x4]."""

    @staticmethod
    def forward(ctx, tokens, tok_off, len_dev, len_host, embed, *w):
        lib = _lib.load()
        bi = len(w) == 8
        tokens = _dev(tokens, torch.int64, "tokens")
        tok_off = _dev(tok_off, torch.int64, "tok_off")
        embed = _dev(embed, name="embed.weight")
        w = [_dev(t) for t in w]
        B, n_tok = len(len_host), int(tokens.numel())
        V, E = embed.shape
        D = w[1].shape[1]
        dev = tokens.device
        save_b = lib.itr_gru_train_save_bytes(n_tok, D, int(bi))
        ws_b = lib.itr_gru_train_workspace_bytes(n_tok, B, E, D)
        save = torch.empty(save_b, device=dev, dtype=torch.uint8)
        ws = torch.empty(ws_b, device=dev, dtype=torch.uint8)
        out = _f32(n_tok, D, dev=dev)
        rev = w[4:] if bi else [None] * 4
        _lib.check(lib.itr_gru_fwd_train(_p(tokens), _p(tok_off), _p(len_dev), len_host.ctypes.data_as(C.c_void_p), B, n_tok,
                                         _p(embed), V, E, D, _p(w[0]), _p(w[1]), _p(w[2]), _p(w[3]), _p(rev[0]), _p(rev[1]),
                                         _p(rev[2]), _p(rev[3]), _p(out), _p(save), save_b, _p(ws), ws_b, _stream()))
        ctx.save_for_backward(tokens, tok_off, len_dev, embed, save, *w)
        ctx.len_host, ctx.bi, ctx.dims = len_host, bi, (B, n_tok, V, E, D)
        return out

    @staticmethod
    def backward(ctx, d_out):
        lib = _lib.load()
        tokens, tok_off, len_dev, embed, save = ctx.saved_tensors[:5]
        w = ctx.saved_tensors[5:]
        B, n_tok, V, E, D = ctx.dims
        bi = ctx.bi
        dev = tokens.device
        d_out = d_out.contiguous()
        ws_b = lib.itr_gru_train_workspace_bytes(n_tok, B, E, D)
        ws = torch.empty(ws_b, device=dev, dtype=torch.uint8)
        d_embed = torch.zeros_like(embed)
        grads = [torch.empty_like(t) for t in w]
        rev_w = [w[4], w[5]] if bi else [None, None]
        rev_g = grads[4:] if bi else [None] * 4
        _lib.check(lib.itr_gru_bwd(_p(tokens), _p(tok_off), _p(len_dev), ctx.len_host.ctypes.data_as(C.c_void_p), B, n_tok, _p(embed),
                                   V, E, D, _p(w[0]), _p(w[1]), _p(rev_w[0]), _p(rev_w[1]), _p(save), _p(d_out), _p(d_embed),
                                   _p(grads[0]), _p(grads[1]), _p(grads[2]), _p(grads[3]), _p(rev_g[0]), _p(rev_g[1]), _p(rev_g[2]),
                                   _p(rev_g[3]), _p(ws), ws_b, _stream()))
        return (None, None, None, None, d_embed) + tuple(grads)


def gru_sequence(tokens_packed, tok_off, lengths, embed, rnn_params, bidirectional):
    """Raw packed GRU outputs (n_tok, D) = (fwd + bwd) / 2 for a bi-GRU, differentiable w.r.t. the weights.
    rnn_params: dict of nn.GRU parameters (weight_ih_l0, weight_hh_l0, bias_ih_l0, bias_hh_l0[, *_reverse])."""
    len_host = _host_i32(lengths)
    len_dev = h2d(len_host.copy(), tokens_packed.device)
    names = ['weight_ih_l0', 'weight_hh_l0', 'bias_ih_l0', 'bias_hh_l0']
    w = [rnn_params[n] for n in names]
    if bidirectional:
        w += [rnn_params[n + '_reverse'] for n in names]
    return _Gru.apply(tokens_packed, tok_off, len_dev, len_host, embed, *w)


class _Cosine(torch.autograd.Function):
    @staticmethod
    def forward(ctx, im, s):
        im, s = _dev(im, name="im"), _dev(s, name="s")
        if im.dim() != 2 or s.dim() != 2 or im.shape[1] != s.shape[1]:
            raise ValueError("cosine_scores: im %s vs s %s" % (tuple(im.shape), tuple(s.shape)))
        ctx.save_for_backward(im, s)
        return _gemm_nt(im, s)

    @staticmethod
    def backward(ctx, dS):
        im, s = ctx.saved_tensors
        dS = dS.contiguous()
        d_im = _gemm_nt(dS, transpose2d(s)) if ctx.needs_input_grad[0] else None          # dS [Ni, Nc] . s [Nc, D]
        d_s = _gemm_tn(dS, im) if ctx.needs_input_grad[1] else None                       # dS^T [Nc, Ni] . im [Ni, D]
        return d_im, d_s


def cosine_scores(im, s):
    return _Cosine.apply(im, s)


class _Order(torch.autograd.Function):
    """order_sim (Objectives.py:24-30) under autograd."""

    @staticmethod
    def forward(ctx, im, s):
        from . import ops
        S = ops.order_scores(im, s)
        ctx.save_for_backward(_dev(im, name="im"), _dev(s, name="s"), S)
        return S

    @staticmethod
    def backward(ctx, dS):
        lib = _lib.load()
        im, s, S = ctx.saved_tensors
        dS = dS.contiguous()
        d_im, d_s = torch.empty_like(im), torch.empty_like(s)
        _lib.check(lib.itr_order_bwd(_p(im), _p(s), _p(S), _p(dS), _p(d_im), _p(d_s), im.shape[0], s.shape[0], im.shape[1], _stream()))
        return d_im, d_s


def order_scores(im, s):
    return _Order.apply(im, s)


class _ScanT2I(torch.autograd.Function):
    """The word axis is zero-padded to a multiple of 32 (ntp) so that every GEMM whose contraction runs over the words
    (dV = dA E) takes the branch-free kernel; padded columns of A / dA are zero and never read by the pair kernels."""

    @staticmethod
    def forward(ctx, V, E, cap_off, cap_len, max_len, norm, agg, ls, ll):
        lib = _lib.load()
        V = _dev(V, name="images")
        E = _dev(E, name="words")
        Bi, R, D = V.shape
        if E.dim() != 2 or E.shape[1] != D:
            raise ValueError("scan_t2i_scores: images (.., %d) vs words %s" % (D, tuple(E.shape)))
        n_tok = E.shape[0]
        ntp = (n_tok + 31) // 32 * 32
        Bc = cap_len.numel()
        dev = V.device
        V2 = V.reshape(Bi * R, D)
        Ep = torch.zeros(ntp, D, device=dev, dtype=torch.float32)
        Ep[:n_tok] = E
        A = _gemm_nt(V2, Ep)                                          # raw dot products of all pairs, [Bi*36, ntp]
        G = _f32(Bi, R, R, dev=dev)
        enorm = _f32(ntp, dev=dev)
        _lib.check(lib.itr_scan_train_prepare(_p(V2), _p(Ep), Bi, ntp, R, D, _p(G), _p(enorm), _stream()))
        S = _f32(Bi, Bc, dev=dev)
        _lib.check(lib.itr_scan_train_fwd(_p(A), ntp, _p(G), _p(enorm), _p(cap_off), _p(cap_len), Bi, Bc, ntp, R, D, max_len,
                                          norm, agg, ls, ll, _p(S), _stream()))
        ctx.save_for_backward(V2, Ep, A, G, enorm, cap_off, cap_len)
        ctx.opts = (Bi, Bc, n_tok, ntp, R, D, max_len, norm, agg, ls, ll)
        return S

    @staticmethod
    def backward(ctx, dS):
        lib = _lib.load()
        V2, Ep, A, G, enorm, cap_off, cap_len = ctx.saved_tensors
        Bi, Bc, n_tok, ntp, R, D, max_len, norm, agg, ls, ll = ctx.opts
        dev = V2.device
        dS = dS.contiguous()
        dA = torch.zeros_like(A)
        dGp = _f32(Bi, Bc, R, R, dev=dev)
        denp = torch.zeros(Bi, ntp, device=dev, dtype=torch.float32)
        _lib.check(lib.itr_scan_train_bwd(_p(A), ntp, _p(G), _p(enorm), _p(cap_off), _p(cap_len), Bi, Bc, ntp, R, D, max_len, norm,
                                          agg, ls, ll, _p(dS), _p(dA), _p(dGp), _p(denp), _stream()))
        dV = _gemm_nt(dA, transpose2d(Ep))                            # dA [Bi*36, ntp] . E [ntp, D]
        dE = _gemm_tn(dA, V2)                                         # dA^T [ntp, Bi*36] . V [Bi*36, D]
        den = colsum(denp)
        _lib.check(lib.itr_scan_train_finish(_p(dGp), Bi, Bc, _p(V2), _p(Ep), _p(enorm), _p(den), ntp, R, D, _p(dV), _p(dE), _stream()))
        return dV.reshape(Bi, R, D), dE[:n_tok], None, None, None, None, None, None, None


def scan_t2i_scores(images, words_packed, cap_off, cap_lens, raw_feature_norm='clipped_l2norm', agg_func='LogSumExp',
                    lambda_lse=6.0, lambda_softmax=9.0):
    """xattn_score_t2i on a training batch -> (n_img, n_cap), differentiable w.r.t. images and words."""
    if raw_feature_norm not in _NORMS:
        raise ValueError("unknown first norm type:", raw_feature_norm)
    if agg_func not in _AGGS:
        raise ValueError("unknown aggfunc: {}".format(agg_func))
    lens = _host_i32(cap_lens)
    dev = images.device
    off = cap_off if torch.is_tensor(cap_off) else h2d(np.asarray(cap_off, np.int64), dev)
    return _ScanT2I.apply(images, words_packed, _dev(off, torch.int64, "cap_off"), h2d(lens.copy(), dev), int(lens.max()),
                          _NORMS[raw_feature_norm], _AGGS[agg_func], float(lambda_softmax), float(lambda_lse))


class _ScanI2T(torch.autograd.Function):
    """xattn_score_i2t on a training batch (scan_train_i2t.hip); same padding of the word axis as _ScanT2I."""

    @staticmethod
    def forward(ctx, V, E, cap_off, cap_len, h_off, h_total, max_len, norm, agg, ls, ll):
        lib = _lib.load()
        V = _dev(V, name="images")
        E = _dev(E, name="words")
        Bi, R, D = V.shape
        if E.dim() != 2 or E.shape[1] != D:
            raise ValueError("scan_i2t_scores: images (.., %d) vs words %s" % (D, tuple(E.shape)))
        n_tok = E.shape[0]
        ntp = (n_tok + 31) // 32 * 32
        Bc = cap_len.numel()
        dev = V.device
        V2 = V.reshape(Bi * R, D)
        Ep = torch.zeros(ntp, D, device=dev, dtype=torch.float32)
        Ep[:n_tok] = E
        A = _gemm_nt(V2, Ep)
        H = _f32(h_total, dev=dev)
        vnorm = _f32(Bi * R, dev=dev)
        _lib.check(lib.itr_scan_train_i2t_prepare(_p(V2), _p(Ep), _p(cap_off), _p(cap_len), _p(h_off), Bi, Bc, R, D, _p(H), _p(vnorm), _stream()))
        S = _f32(Bi, Bc, dev=dev)
        _lib.check(lib.itr_scan_train_i2t_fwd(_p(A), ntp, _p(H), _p(h_off), _p(vnorm), _p(cap_off), _p(cap_len), Bi, Bc, ntp, R, D, max_len,
                                              norm, agg, ls, ll, _p(S), _stream()))
        ctx.save_for_backward(V2, Ep, A, H, vnorm, cap_off, cap_len, h_off)
        ctx.opts = (Bi, Bc, n_tok, ntp, R, D, max_len, norm, agg, ls, ll, h_total)
        return S

    @staticmethod
    def backward(ctx, dS):
        lib = _lib.load()
        V2, Ep, A, H, vnorm, cap_off, cap_len, h_off = ctx.saved_tensors
        Bi, Bc, n_tok, ntp, R, D, max_len, norm, agg, ls, ll, h_total = ctx.opts
        dev = V2.device
        dS = dS.contiguous()
        dA = torch.zeros_like(A)
        dHp = _f32(Bi, h_total, dev=dev)
        dvnp = _f32(Bc, Bi * R, dev=dev)
        _lib.check(lib.itr_scan_train_i2t_bwd(_p(A), ntp, _p(H), _p(h_off), h_total, _p(vnorm), _p(cap_off), _p(cap_len), Bi, Bc, ntp, R, D,
                                              max_len, norm, agg, ls, ll, _p(dS), _p(dA), _p(dHp), _p(dvnp), _stream()))
        dV = _gemm_nt(dA, transpose2d(Ep))
        dE = _gemm_tn(dA, V2)
        dH = colsum(dHp)
        dvn = colsum(dvnp)
        _lib.check(lib.itr_scan_train_i2t_finish(_p(dH), _p(h_off), _p(cap_off), _p(cap_len), Bc, _p(Ep), _p(V2), _p(vnorm), _p(dvn), Bi, R, D,
                                                 _p(dV), _p(dE), _stream()))
        return (dV.reshape(Bi, R, D), dE[:n_tok]) + (None,) * 9


def scan_i2t_scores(images, words_packed, cap_off, cap_lens, raw_feature_norm='clipped_l2norm', agg_func='LogSumExp',
                    lambda_lse=6.0, lambda_softmax=9.0):
    """xattn_score_i2t on a training batch -> (n_img, n_cap), differentiable w.r.t. images and words."""
    if raw_feature_norm not in _NORMS:
        raise ValueError("unknown first norm type:", raw_feature_norm)
    if agg_func not in _AGGS:
        raise ValueError("unknown aggfunc: {}".format(agg_func))
    lens = _host_i32(cap_lens)
    dev = images.device
    off = cap_off if torch.is_tensor(cap_off) else h2d(np.asarray(cap_off, np.int64), dev)
    sq = lens.astype(np.int64) ** 2
    h_off = np.concatenate([[0], np.cumsum(sq)[:-1]]).astype(np.int64)
    return _ScanI2T.apply(images, words_packed, _dev(off, torch.int64, "cap_off"), h2d(lens.copy(), dev),
                          h2d(h_off, dev), int(sq.sum()), int(lens.max()), _NORMS[raw_feature_norm], _AGGS[agg_func],
                          float(lambda_softmax), float(lambda_lse))


# ------------------------------------------------------------------------------------------ optimizer
# ---------------------------------------------------------------------------------------------------------------------
# transformer towers (SAEM): dropout, residual LayerNorm, gelu, short-sequence attention, relu + max-pool, mean over regions
class _Dropout(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, p, seed):
        lib = _lib.load()
        x = _dev(x, name="x")
        y = torch.empty_like(x)
        _lib.check(lib.itr_dropout(_p(x), _p(y), x.numel(), float(p), int(seed), 0, _stream()))
        ctx.p, ctx.seed = float(p), int(seed)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        dy = dy.contiguous()
        dx = torch.empty_like(dy)
        _lib.check(lib.itr_dropout(_p(dy), _p(dx), dy.numel(), ctx.p, ctx.seed, 0, _stream()))
        return dx, None, None


class DropoutSeeds(object):
    """Per-step seeds of the dropout sites: torch's generator supplies one base seed per training step (so
    torch.manual_seed makes runs repeatable), every call site takes the next counter value."""

    def __init__(self):
        self.base, self.n = 0, 0

    def new_step(self):
        self.base = int(torch.randint(0, 2 ** 31 - 1, (1,)).item())
        self.n = 0

    def next(self):
        self.n += 1
        return self.base * 4096 + self.n


def dropout(x, p, seeds, training=True):
    """nn.Dropout(p)(x): identity when not training or p == 0."""
    if not training or p <= 0.0:
        return x
    return _Dropout.apply(x, p, seeds.next())


class _AddLayerNorm(torch.autograd.Function):
    """BERTLayerNorm(x + residual) (bert.py:113-126)."""

    @staticmethod
    def forward(ctx, x, residual, gamma, beta, eps):
        lib = _lib.load()
        x = _dev(x, name="x")
        H = x.shape[-1]
        rows = x.numel() // H
        res = _dev(residual, name="residual") if residual is not None else None
        g, b = _dev(gamma, name="gamma"), _dev(beta, name="beta")
        z, out = torch.empty_like(x), torch.empty_like(x)
        mean, rstd = _f32(rows, dev=x.device), _f32(rows, dev=x.device)
        _lib.check(lib.itr_add_ln_fwd(_p(x), _p(res), _p(g), _p(b), _p(z), _p(out), _p(mean), _p(rstd), rows, H, float(eps), _stream()))
        ctx.save_for_backward(z, mean, rstd, g)
        ctx.has_res = residual is not None
        return out

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        z, mean, rstd, g = ctx.saved_tensors
        dy = dy.contiguous()
        H = z.shape[-1]
        rows = z.numel() // H
        dz, t = torch.empty_like(z), torch.empty_like(z)
        _lib.check(lib.itr_ln_bwd(_p(dy), _p(z), _p(mean), _p(rstd), _p(g), _p(dz), _p(t), rows, H, _stream()))
        dgamma = colsum(t.view(rows, H)) if ctx.needs_input_grad[2] else None
        dbeta = colsum(dy.view(rows, H)) if ctx.needs_input_grad[3] else None
        return dz, (dz if ctx.has_res else None), dgamma, dbeta, None


def add_layernorm(x, residual, gamma, beta, eps=1e-12):
    return _AddLayerNorm.apply(x, residual, gamma, beta, eps)


class _Gelu(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        lib = _lib.load()
        x = _dev(x, name="x")
        y = torch.empty_like(x)
        _lib.check(lib.itr_gelu(_p(x), None, _p(y), x.numel(), _stream()))
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x, = ctx.saved_tensors
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        _lib.check(lib.itr_gelu(_p(x), _p(dy), _p(dx), x.numel(), _stream()))
        return dx


def gelu(x):
    return _Gelu.apply(x)


class _Mha(torch.autograd.Function):
    """BERTSelfAttention core (bert.py:175-215) on the fused QKV projection output qkv [B*L, 3A] (thirds Q | K | V)."""

    @staticmethod
    def forward(ctx, qkv, mask01, B, L, heads, p_drop, seed):
        lib = _lib.load()
        qkv = _dev(qkv, name="qkv")
        A = qkv.shape[1] // 3
        dk = A // heads
        m = _dev(mask01.to(torch.float32), name="mask") if mask01 is not None else None
        P = _f32(B, heads, L, L, dev=qkv.device)
        out = _f32(B * L, A, dev=qkv.device)
        base = qkv.data_ptr()
        scale = 1.0 / float(np.sqrt(dk))
        _lib.check(lib.itr_mha_train_fwd(C.c_void_p(base), C.c_void_p(base + 4 * A), C.c_void_p(base + 8 * A), 3 * A, _p(m), B, L, heads, dk,
                                         scale, float(p_drop), int(seed), _p(P), _p(out), A, _stream()))
        ctx.save_for_backward(qkv, P, m if m is not None else torch.empty(0, device=qkv.device))
        ctx.cfg = (B, L, heads, dk, A, scale, float(p_drop), int(seed), m is not None)
        return out

    @staticmethod
    def backward(ctx, dctx):
        lib = _lib.load()
        qkv, P, m = ctx.saved_tensors
        B, L, heads, dk, A, scale, p_drop, seed, has_mask = ctx.cfg
        dctx = dctx.contiguous()
        dqkv = torch.empty_like(qkv)
        base, gb = qkv.data_ptr(), dqkv.data_ptr()
        _lib.check(lib.itr_mha_train_bwd(C.c_void_p(base), C.c_void_p(base + 4 * A), C.c_void_p(base + 8 * A), 3 * A, _p(m) if has_mask else None,
                                         B, L, heads, dk, scale, p_drop, seed, _p(P), _p(dctx), A, C.c_void_p(gb), C.c_void_p(gb + 4 * A),
                                         C.c_void_p(gb + 8 * A), 3 * A, _stream()))
        return dqkv, None, None, None, None, None, None


def mha(qkv, mask01, B, L, heads, p_drop=0.0, seed=0):
    return _Mha.apply(qkv, mask01, B, L, heads, p_drop, seed)


class _ReluMaxpool(torch.autograd.Function):
    """max over positions of relu(x), x [B, npos, C] (TextEncoder.py:122-124)."""

    @staticmethod
    def forward(ctx, x):
        lib = _lib.load()
        x = _dev(x, name="x")
        B, npos, Cc = x.shape
        out = _f32(B, Cc, dev=x.device)
        arg = torch.empty(B, Cc, device=x.device, dtype=torch.int32)
        _lib.check(lib.itr_relu_maxpool_arg(_p(x), B, npos, Cc, _p(out), _p(arg), _stream()))
        ctx.save_for_backward(arg)
        ctx.npos = npos
        return out

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        arg, = ctx.saved_tensors
        dy = dy.contiguous()
        B, Cc = dy.shape
        dx = _f32(B, ctx.npos, Cc, dev=dy.device)
        _lib.check(lib.itr_relu_maxpool_bwd(_p(dy), _p(arg), B, ctx.npos, Cc, _p(dx), _stream()))
        return dx


def relu_maxpool(x):
    return _ReluMaxpool.apply(x)


class _MeanMid(torch.autograd.Function):
    """torch.mean(x, 1) of x [B, R, F]."""

    @staticmethod
    def forward(ctx, x):
        from . import ops
        ctx.R = x.shape[1]
        return ops.mean_mid(x)

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        dy = dy.contiguous()
        B, F = dy.shape
        dx = _f32(B, ctx.R, F, dev=dy.device)
        _lib.check(lib.itr_bcast_mid(_p(dy), _p(dx), B, ctx.R, F, 1.0 / ctx.R, _stream()))
        return dx


def mean_mid(x):
    return _MeanMid.apply(x)


class _Angular(torch.autograd.Function):
    """AngularLoss.angular_loss on M1 = anchors others^T, M2 = positives others^T, Q = anchors positives^T (Objectives.py:262-290)."""

    @staticmethod
    def forward(ctx, M1, M2, Q, angle_bound, max_violation):
        lib = _lib.load()
        M1, M2, Q = _dev(M1, name="M1"), _dev(M2, name="M2"), _dev(Q, name="Q")
        n = M1.shape[0]
        if not (tuple(M1.shape) == tuple(M2.shape) == tuple(Q.shape) == (n, n)):
            raise ValueError("angular_loss: needs three n x n matrices, got %s %s %s" % (tuple(M1.shape), tuple(M2.shape), tuple(Q.shape)))
        dev = M1.device
        loss, row, stat, den = _f32(1, dev=dev), _f32(n, dev=dev), _f32(n, dev=dev), _f32(n, dev=dev)
        arg = torch.empty(n, device=dev, dtype=torch.int32)
        _lib.check(lib.itr_angular_fwd(_p(M1), _p(M2), _p(Q), n, float(angle_bound), int(bool(max_violation)), _p(loss), _p(row), _p(stat),
                                       _p(den), _p(arg), _stream()))
        ctx.save_for_backward(M1, M2, Q, stat, den, arg)
        ctx.cfg = (float(angle_bound), int(bool(max_violation)))
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        M1, M2, Q, stat, den, arg = ctx.saved_tensors
        n = M1.shape[0]
        dM, dQ = torch.empty_like(M1), torch.empty_like(Q)
        _lib.check(lib.itr_angular_bwd(_p(M1), _p(M2), _p(Q), n, ctx.cfg[0], ctx.cfg[1], _p(stat), _p(den), _p(arg),
                                       _p(g.contiguous().reshape(1).float()), _p(dM), _p(dQ), _stream()))
        return dM, dM, dQ, None, None


def angular_loss(anchors, positives, others, angle_bound=1.0, max_violation=True):
    """One direction of SAEM's AngularLoss: three GEMMs + the row kernel, differentiable w.r.t. all three inputs."""
    return _Angular.apply(linear(anchors, others), linear(positives, others), linear(anchors, positives), angle_bound, max_violation)


class _Diversity(torch.autograd.Function):
    """DiversityRegularization (Objectives.py:521-542) of smry [B, R, K]."""

    @staticmethod
    def forward(ctx, smry):
        lib = _lib.load()
        smry = _dev(smry, name="smry_mat")
        B, R, K = smry.shape
        part, loss = _f32(max(B, 1), dev=smry.device), _f32(1, dev=smry.device)
        _lib.check(lib.itr_diversity_fwd(_p(smry), B, R, K, _p(part), _p(loss), _stream()))
        ctx.save_for_backward(smry)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        smry, = ctx.saved_tensors
        B, R, K = smry.shape
        d = torch.empty_like(smry)
        _lib.check(lib.itr_diversity_bwd(_p(smry), B, R, K, _p(g.contiguous().reshape(1).float()), _p(d), _stream()))
        return d


def diversity_reg(smry_mat):
    return _Diversity.apply(smry_mat)


# ---------------------------------------------------------------------------------------------------------------------
# CAMERA towers: elementwise product, activations, attention gate, training-mode BatchNorm, l2norm across regions,
# multi-view summarisation, multi-view matching
def _ew_mul(a, b):
    lib = _lib.load()
    out = torch.empty_like(a)
    _lib.check(lib.itr_ew_mul(_p(a), _p(b), _p(out), a.numel(), _stream()))
    return out


class _Mul(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        a, b = _dev(a, name="a"), _dev(b, name="b")
        if a.shape != b.shape:
            raise ValueError("mul: shapes %s vs %s" % (tuple(a.shape), tuple(b.shape)))
        ctx.save_for_backward(a, b)
        return _ew_mul(a, b)

    @staticmethod
    def backward(ctx, dy):
        a, b = ctx.saved_tensors
        dy = dy.contiguous()
        return _ew_mul(dy, b), _ew_mul(dy, a)


def mul(a, b):
    return _Mul.apply(a, b)


_ACT_CODE = {'relu': 1, 'tanh': 2, 'sigmoid': 3, 'leaky_relu': 4}


class _Act(torch.autograd.Function):
    """relu / tanh / sigmoid; the derivative is taken from the output."""

    @staticmethod
    def forward(ctx, x, kind):
        from . import ops
        y = ops.affine_cols(_dev(x, name="x"), act=kind)
        ctx.save_for_backward(y)
        ctx.kind = kind
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        y, = ctx.saved_tensors
        dy = dy.contiguous()
        dx = torch.empty_like(y)
        _lib.check(lib.itr_act_bwd(_p(y), _p(dy), _p(dx), y.numel(), _ACT_CODE[ctx.kind], _stream()))
        return dx, None


def act(x, kind):
    return _Act.apply(x, kind)


class _AddBcastMidAct(torch.autograd.Function):
    """act(x [B, N, H] + v [B, H] broadcast over N); the derivative is taken from the output."""

    @staticmethod
    def forward(ctx, x, v, kind):
        lib = _lib.load()
        x, v = _dev(x, name="x"), _dev(v, name="v")
        B, N, H = x.shape
        if tuple(v.shape) != (B, H):
            raise ValueError("add_bcast_mid_act: x %s vs v %s" % (tuple(x.shape), tuple(v.shape)))
        y = torch.empty_like(x)
        code = 0 if kind is None else _ACT_CODE[kind]
        _lib.check(lib.itr_add_bcast_mid_act(_p(x), _p(v), _p(y), B, N, H, code, _stream()))
        ctx.save_for_backward(y)
        ctx.code = code
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        y, = ctx.saved_tensors
        B, N, H = y.shape
        dx, dv = torch.empty_like(y), _f32(B, H, dev=y.device)
        _lib.check(lib.itr_add_bcast_mid_act_bwd(_p(y), _p(dy.contiguous()), _p(dx), _p(dv), B, N, H, ctx.code, _stream()))
        return dx, dv, None


def add_bcast_mid_act(x, v, kind=None):
    return _AddBcastMidAct.apply(x, v, kind)


class _AddAttnScore(torch.autograd.Function):
    """e[b, n] = sum_h w[h] tanh(x[b, n, h] + v[b, h]): the decoder attention's linear2(tanh(linear1(cat(enc, hidden)))) with the encoder
    half x computed once per batch (Attention.forward, Fusionmodule.py:136-141).  The tanh tensor is never stored: the backward recomputes it."""

    @staticmethod
    def forward(ctx, x, v, w):
        lib = _lib.load()
        x, v, w = _dev(x, name="x"), _dev(v, name="v"), _dev(w, name="w")
        B, N, H = x.shape
        if tuple(v.shape) != (B, H) or w.numel() != H:
            raise ValueError("addattn_score: x %s vs v %s, w %s" % (tuple(x.shape), tuple(v.shape), tuple(w.shape)))
        e = _f32(B, N, dev=x.device)
        _lib.check(lib.itr_addattn_score(_p(x), _p(v), _p(w), _p(e), B, N, H, _stream()))
        ctx.save_for_backward(x, v, w)
        return e

    @staticmethod
    def backward(ctx, de):
        lib = _lib.load()
        x, v, w = ctx.saved_tensors
        B, N, H = x.shape
        dx, dv, dwp = torch.empty_like(x), _f32(B, H, dev=x.device), _f32(B, H, dev=x.device)
        _lib.check(lib.itr_addattn_score_bwd(_p(x), _p(v), _p(w), _p(de.contiguous()), _p(dx), _p(dv), _p(dwp), B, N, H, _stream()))
        return dx, dv, colsum(dwp).reshape(w.shape)


def addattn_score(x, v, w):
    return _AddAttnScore.apply(x, v, w)


class _GateApply(torch.autograd.Function):
    """q' = q * M[:, :dk], k' = k * M[:, dk:]  (camera_.py:41-44); q, k [rows, dk], M [rows, 2 dk]."""

    @staticmethod
    def forward(ctx, q, k, M):
        lib = _lib.load()
        q, k, M = _dev(q, name="q"), _dev(k, name="k"), _dev(M, name="M")
        rows, dk = q.shape
        qo, ko = torch.empty_like(q), torch.empty_like(k)
        _lib.check(lib.itr_gate_apply(_p(q), _p(k), _p(M), _p(qo), _p(ko), rows, dk, _stream()))
        ctx.save_for_backward(q, k, M)
        return qo, ko

    @staticmethod
    def backward(ctx, dqo, dko):
        lib = _lib.load()
        q, k, M = ctx.saved_tensors
        rows, dk = q.shape
        dq, dkk, dM = torch.empty_like(q), torch.empty_like(k), torch.empty_like(M)
        _lib.check(lib.itr_gate_apply_bwd(_p(q), _p(k), _p(M), _p(dqo.contiguous()), _p(dko.contiguous()), _p(dq), _p(dkk), _p(dM), rows, dk,
                                          _stream()))
        return dq, dkk, dM


def gate_apply(q, k, M):
    return _GateApply.apply(q, k, M)


class _BatchNormTrain(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps, stats_out):
        lib = _lib.load()
        x = _dev(x, name="x")
        N, Cc = x.shape
        g, b = _dev(gamma, name="gamma"), _dev(beta, name="beta")
        y = torch.empty_like(x)
        mean, invstd = _f32(Cc, dev=x.device), _f32(Cc, dev=x.device)
        scratch = torch.empty(lib.itr_bn_train_scratch_bytes(N, Cc), device=x.device, dtype=torch.uint8)
        _lib.check(lib.itr_bn_train_fwd(_p(x), _p(g), _p(b), _p(y), _p(mean), _p(invstd), N, Cc, float(eps), _p(scratch), _stream()))
        ctx.save_for_backward(x, mean, invstd, g)
        stats_out.append((mean, invstd))
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x, mean, invstd, g = ctx.saved_tensors
        N, Cc = x.shape
        dx, dg, db = torch.empty_like(x), _f32(Cc, dev=x.device), _f32(Cc, dev=x.device)
        scratch = torch.empty(lib.itr_bn_train_scratch_bytes(N, Cc), device=x.device, dtype=torch.uint8)
        _lib.check(lib.itr_bn_train_bwd(_p(dy.contiguous()), _p(x), _p(mean), _p(invstd), _p(g), _p(dx), _p(dg), _p(db), N, Cc, _p(scratch),
                                        _stream()))
        return dx, dg, db, None, None


class _SyncBatchNormTrain(torch.autograd.Function):
    """_BatchNormTrain on a row shard of the batch: the statistics of the GLOBAL batch from one all-reduce of the per-rank
    (count, sum, sum of squares) in the forward and one of (sum dy, sum dy*xhat) in the backward, so that a data-parallel step
    normalises exactly like one process on the whole batch.  The local kernels do the row reductions; gamma / beta keep their
    LOCAL gradients (Adam.step sums parameter gradients over the ranks)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, stats_out, comm):
        import torch.distributed as dist
        from . import ops
        lib = _lib.load()
        x = _dev(x, name="x")
        N, Cc = x.shape
        g, b = _dev(gamma, name="gamma"), _dev(beta, name="beta")
        mean, invstd = _f32(Cc, dev=x.device), _f32(Cc, dev=x.device)
        scratch = torch.empty(lib.itr_bn_train_scratch_bytes(N, Cc), device=x.device, dtype=torch.uint8)
        _lib.check(lib.itr_bn_train_fwd(_p(x), _p(g), _p(b), _p(torch.empty_like(x)), _p(mean), _p(invstd), N, Cc, float(eps), _p(scratch),
                                        _stream()))
        m64 = mean.double()
        var64 = 1.0 / (invstd.double() * invstd.double()) - eps                  # biased variance of the local rows
        acc = torch.cat([torch.full((1,), float(N), device=x.device, dtype=torch.float64), m64 * N, (var64 + m64 * m64) * N])
        comm.all_reduce(acc, dist.ReduceOp.SUM)
        n_glob = float(acc[0])
        gm = acc[1:1 + Cc] / n_glob
        gvar = (acc[1 + Cc:] / n_glob - gm * gm).clamp_min(0.0)
        ginv = torch.rsqrt(gvar + eps)
        mean, invstd = gm.float(), ginv.float()
        scale = (g.double() * ginv).float()
        y = ops.affine_cols(x, scale=scale, shift=(b.double() - gm * g.double() * ginv).float())
        ctx.save_for_backward(x, mean, invstd, g)
        ctx.comm, ctx.n_glob = comm, n_glob
        stats_out.append((mean, invstd, n_glob))
        return y

    @staticmethod
    def backward(ctx, dy):
        import torch.distributed as dist
        from . import ops
        lib = _lib.load()
        x, mean, invstd, g = ctx.saved_tensors
        N, Cc = x.shape
        dy = dy.contiguous()
        dg, db = _f32(Cc, dev=x.device), _f32(Cc, dev=x.device)
        scratch = torch.empty(lib.itr_bn_train_scratch_bytes(N, Cc), device=x.device, dtype=torch.uint8)
        # the kernel's row sums with the GLOBAL mean / invstd: dg = sum dy*xhat, db = sum dy over the local rows (its dx, which
        # would close the formula with local sums only, is discarded)
        _lib.check(lib.itr_bn_train_bwd(_p(dy), _p(x), _p(mean), _p(invstd), _p(g), _p(torch.empty_like(x)), _p(dg), _p(db), N, Cc, _p(scratch),
                                        _stream()))
        tot = torch.cat([dg, db]).double()
        ctx.comm.all_reduce(tot, dist.ReduceOp.SUM)
        dg_g, db_g = tot[:Cc] / ctx.n_glob, tot[Cc:] / ctx.n_glob
        gi = g.double() * invstd.double()
        # dx = gamma*invstd * (dy - mean(dy) - xhat * mean(dy*xhat)),  xhat = (x - mean) * invstd
        s_x = -gi * invstd.double() * dg_g
        c = gi * (mean.double() * invstd.double() * dg_g - db_g)
        dx = ops.affine_cols(x, scale=s_x.float(), shift=c.float(), residual=ops.affine_cols(dy, scale=gi.float(), shift=torch.zeros_like(g)))
        return dx, dg, db, None, None, None


_BN_SYNC = [None]


class bn_sync:
    """`with bn_sync(comm):` -- batch_norm_train inside normalises over the rows of ALL ranks (data-parallel train_emb)."""

    def __init__(self, comm):
        self.comm = comm if (comm is not None and comm.on) else None

    def __enter__(self):
        self.prev, _BN_SYNC[0] = _BN_SYNC[0], self.comm
        return self

    def __exit__(self, *exc):
        _BN_SYNC[0] = self.prev
        return False


def batch_norm_train(x2d, bn):
    """nn.BatchNorm1d(x2d [N, C]) in training mode: batch statistics, and the module's running statistics updated like torch
    (momentum, unbiased variance, num_batches_tracked).  Inside `bn_sync(comm)` the batch is the rows of all ranks."""
    stats = []
    if _BN_SYNC[0] is not None:
        y = _SyncBatchNormTrain.apply(x2d, bn.weight, bn.bias, bn.eps, stats, _BN_SYNC[0])
        mean, invstd, N = stats[0]
    else:
        y = _BatchNormTrain.apply(x2d, bn.weight, bn.bias, bn.eps, stats)
        (mean, invstd), N = stats[0], x2d.shape[0]
    with torch.no_grad():
        var_b = 1.0 / (invstd * invstd) - bn.eps
        mom = bn.momentum if bn.momentum is not None else 0.1
        bn.running_mean.mul_(1 - mom).add_(mean, alpha=mom)
        bn.running_var.mul_(1 - mom).add_(var_b * (N / max(N - 1, 1)), alpha=mom)
        bn.num_batches_tracked += 1
    return y


class _L2NormMid(torch.autograd.Function):
    """utils.l2norm(x) with the reference's default dim=1 on x [B, R, D]."""

    @staticmethod
    def forward(ctx, x, eps):
        lib = _lib.load()
        x = _dev(x, name="x")
        B, R, D = x.shape
        z, n = torch.empty_like(x), _f32(B, D, dev=x.device)
        _lib.check(lib.itr_l2norm_mid_fwd(_p(x), _p(z), _p(n), B, R, D, float(eps), _stream()))
        ctx.save_for_backward(z, n)
        ctx.eps = float(eps)
        return z

    @staticmethod
    def backward(ctx, dz):
        lib = _lib.load()
        z, n = ctx.saved_tensors
        B, R, D = z.shape
        dx = torch.empty_like(z)
        _lib.check(lib.itr_l2norm_mid_bwd(_p(dz.contiguous()), _p(z), _p(n), _p(dx), B, R, D, ctx.eps, _stream()))
        return dx, None


def l2norm_mid(x, eps=1e-8):
    return _L2NormMid.apply(x, eps)


class _Smry(torch.autograd.Function):
    """softmax(smry_mat, dim=1)^T x  (ImgEncoder.py:386-387): smry [B, R, K], x [B, R, D] -> [B, K, D]."""

    @staticmethod
    def forward(ctx, smry, x):
        lib = _lib.load()
        smry, x = _dev(smry, name="smry"), _dev(x, name="x")
        B, R, K = smry.shape
        D = x.shape[2]
        L, out = torch.empty_like(smry), _f32(B, K, D, dev=x.device)
        _lib.check(lib.itr_smry_fwd(_p(smry), _p(x), _p(L), _p(out), B, R, K, D, _stream()))
        ctx.save_for_backward(x, L)
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        x, L = ctx.saved_tensors
        B, R, K = L.shape
        D = x.shape[2]
        dx, dsm, scratch = torch.empty_like(x), torch.empty_like(L), torch.empty_like(L)
        _lib.check(lib.itr_smry_bwd(_p(x), _p(L), _p(dout.contiguous()), _p(dx), _p(dsm), _p(scratch), B, R, K, D, _stream()))
        return dsm, dx


def summarize(smry, x):
    return _Smry.apply(smry, x)


def _bmm(A, B, ta, tb, M, N, K):
    """C[b] (M x N) = op(A[b]) op(B[b]) for a batch of small operands.  On the matrix cores where a batched kernel of the library has
    the shape (round 6: VSRN's relation products theta phi^T / R g and their gradients were 8.6 ms of its step on the thread-per-element
    itr_bmm_small): A B^T with <= 64 output columns = itr_sgt_dp; A^T B = itr_gemm_tn_batched; A B = the same on a transposed copy of the
    (small) left operand."""
    lib = _lib.load()
    Bn = A.shape[0]
    out = _f32(Bn, M, N, dev=A.device)
    if Bn <= 65535 and not ta and tb and N <= 64 and K % 4 == 0:            # A [Bn, M, K] . B [Bn, N, K]^T
        _lib.check(lib.itr_sgt_dp(_p(A), _p(B), Bn, M, N, K, _p(out), _stream()))
        return out
    if Bn <= 65535 and not tb:
        if not ta:
            A = A.transpose(1, 2).contiguous()                              # [Bn, K, M]: the reduced index becomes the row
        _lib.check(lib.itr_gemm_tn_batched(_p(A), M, K * M, _p(B), N, K * N, _p(out), N, M * N, K, M, N, Bn, _stream()))
        return out
    _lib.check(lib.itr_bmm_small(_p(A), _p(B), _p(out), Bn, M, N, K, int(ta), int(tb), _stream()))
    return out


class _BmmNT(torch.autograd.Function):
    """C[b] = A[b] B[b]^T for small operands: A [Bn, M, K], B [Bn, N, K] -> [Bn, M, N]  (torch.bmm(q, k.permute(0, 2, 1)))."""

    @staticmethod
    def forward(ctx, A, B):
        A, B = _dev(A, name="A"), _dev(B, name="B")
        if A.dim() != 3 or B.dim() != 3 or A.shape[0] != B.shape[0] or A.shape[2] != B.shape[2]:
            raise ValueError("bmm_nt: A %s vs B %s" % (tuple(A.shape), tuple(B.shape)))
        ctx.save_for_backward(A, B)
        return _bmm(A, B, 0, 1, A.shape[1], B.shape[1], A.shape[2])

    @staticmethod
    def backward(ctx, dC):
        A, B = ctx.saved_tensors
        dC = dC.contiguous()
        M, N, K = A.shape[1], B.shape[1], A.shape[2]
        dA = _bmm(dC, B, 0, 0, M, K, N)            # dC [M, N] . B [N, K]
        dB = _bmm(dC, A, 1, 0, N, K, M)            # dC^T [N, M] . A [M, K]
        return dA, dB


def bmm_nt(A, B):
    return _BmmNT.apply(A, B)


class _BmmNN(torch.autograd.Function):
    """C[b] = A[b] B[b] for small operands: A [Bn, M, K], B [Bn, K, N] -> [Bn, M, N]."""

    @staticmethod
    def forward(ctx, A, B):
        A, B = _dev(A, name="A"), _dev(B, name="B")
        if A.dim() != 3 or B.dim() != 3 or A.shape[0] != B.shape[0] or A.shape[2] != B.shape[1]:
            raise ValueError("bmm_nn: A %s vs B %s" % (tuple(A.shape), tuple(B.shape)))
        ctx.save_for_backward(A, B)
        return _bmm(A, B, 0, 0, A.shape[1], B.shape[2], A.shape[2])

    @staticmethod
    def backward(ctx, dC):
        A, B = ctx.saved_tensors
        dC = dC.contiguous()
        M, K, N = A.shape[1], A.shape[2], B.shape[2]
        dA = _bmm(dC, B, 0, 1, M, K, N)            # dC [M, N] . B^T [N, K]
        dB = _bmm(A, dC, 1, 0, K, N, M)            # A^T [K, M] . dC [M, N]
        return dA, dB


def bmm_nn(A, B):
    return _BmmNN.apply(A, B)


class _GruCell(torch.autograd.Function):
    """One nn.GRU step from the two projections (gate order r, z, n)."""

    @staticmethod
    def forward(ctx, gi, gh, h):
        lib = _lib.load()
        gi, gh, h = _dev(gi, name="gi"), _dev(gh, name="gh"), _dev(h, name="h")
        B, H = h.shape
        hn, gates = torch.empty_like(h), torch.empty_like(gi)
        _lib.check(lib.itr_gru_cell_fwd(_p(gi), _p(gh), _p(h), _p(hn), _p(gates), B, H, _stream()))
        ctx.save_for_backward(gates, gh, h)
        return hn

    @staticmethod
    def backward(ctx, dhn):
        lib = _lib.load()
        gates, gh, h = ctx.saved_tensors
        B, H = h.shape
        dgi, dgh, dh = torch.empty_like(gates), torch.empty_like(gates), torch.empty_like(h)
        _lib.check(lib.itr_gru_cell_bwd(_p(dhn.contiguous()), _p(gates), _p(gh), _p(h), _p(dgi), _p(dgh), _p(dh), B, H, _stream()))
        return dgi, dgh, dh


def gru_cell(x, h, rnn):
    """h' = nn.GRU step (single layer, `rnn` supplies weight_ih_l0 / weight_hh_l0 / bias_ih_l0 / bias_hh_l0)."""
    gi = linear(x, rnn.weight_ih_l0, rnn.bias_ih_l0)
    gh = linear(h, rnn.weight_hh_l0, rnn.bias_hh_l0)
    return _GruCell.apply(gi, gh, h)


class _NllLogSoftmax(torch.autograd.Function):
    """-mask[b] * log_softmax(logits[b])[target[b]] per row."""

    @staticmethod
    def forward(ctx, logits, target, mask):
        lib = _lib.load()
        logits = _dev(logits, name="logits")
        target = _dev(target, torch.int64, "target")
        mask = _dev(mask.to(torch.float32), name="mask")
        B, V = logits.shape
        loss, lse = _f32(B, dev=logits.device), _f32(B, dev=logits.device)
        _lib.check(lib.itr_nll_logsoftmax_fwd(_p(logits), _p(target), _p(mask), _p(loss), _p(lse), B, V, _stream()))
        ctx.save_for_backward(logits, target, mask, lse)
        return loss

    @staticmethod
    def backward(ctx, dloss):
        lib = _lib.load()
        logits, target, mask, lse = ctx.saved_tensors
        B, V = logits.shape
        dlogits = torch.empty_like(logits)
        _lib.check(lib.itr_nll_logsoftmax_bwd(_p(logits), _p(target), _p(mask), _p(lse), _p(dloss.contiguous()), _p(dlogits), B, V, _stream()))
        return dlogits, None, None


def nll_logsoftmax(logits, target, mask):
    return _NllLogSoftmax.apply(logits, target, mask)


class _GroupMax(torch.autograd.Function):
    """max over the k view rows of every image: T [Ni * k, Nc] -> [Ni, Nc]."""

    @staticmethod
    def forward(ctx, T, k):
        lib = _lib.load()
        T = _dev(T, name="T")
        Ni, Nc = T.shape[0] // k, T.shape[1]
        S = _f32(Ni, Nc, dev=T.device)
        arg = torch.empty(Ni, Nc, device=T.device, dtype=torch.int32)
        _lib.check(lib.itr_groupmax_fwd(_p(T), Ni, k, Nc, _p(S), _p(arg), _stream()))
        ctx.save_for_backward(arg)
        ctx.k = k
        return S

    @staticmethod
    def backward(ctx, dS):
        lib = _lib.load()
        arg, = ctx.saved_tensors
        Ni, Nc = arg.shape
        dT = _f32(Ni * ctx.k, Nc, dev=dS.device)
        _lib.check(lib.itr_groupmax_bwd(_p(dS.contiguous()), _p(arg), Ni, ctx.k, Nc, _p(dT), _stream()))
        return dT, None


def mvm_scores(img_views, caps):
    """MultiViewMatching (Fusionmodule.py:674-692) on the tape: img_views [Ni, k, D], caps [Nc, D] -> [Ni, Nc]."""
    Ni, k, D = img_views.shape
    return _GroupMax.apply(cosine_scores(img_views.reshape(Ni * k, D), caps), k)


class _DPGatherRows(torch.autograd.Function):
    """Row blocks of every rank concatenated rank-major (one RCCL all-gather).  Two kinds of backward:
    reduce=True   the gathered rows feed a computation that DIFFERS per rank (every rank scores its own image rows
                  against all caption embeddings): each rank holds a partial gradient for all rows -> sum all-reduce,
                  then the rank's own rows;
    reduce=False  the gathered rows feed a computation REPLICATED on every rank (the hinge over the gathered score
                  matrix): the gradient is already complete -> the rank's own rows, no exchange."""

    @staticmethod
    def forward(ctx, local, comm, counts, reduce):
        counts = [int(c) for c in counts]
        if local.shape[0] != counts[comm.rank]:
            raise ValueError("dp_gather_rows: %d local rows, counts[%d] = %d" % (local.shape[0], comm.rank, counts[comm.rank]))
        buf, maxrows = comm.all_gather_rows(local.contiguous(), counts)
        ctx.comm, ctx.counts, ctx.reduce = comm, counts, reduce
        if all(c == maxrows for c in counts):
            return buf
        return torch.cat([buf[q * maxrows:q * maxrows + counts[q]] for q in range(comm.world)], 0)

    @staticmethod
    def backward(ctx, g):
        import torch.distributed as dist
        comm, counts = ctx.comm, ctx.counts
        lo = sum(counts[:comm.rank])
        if ctx.reduce:
            g = g.clone()
            comm.all_reduce(g, dist.ReduceOp.SUM)
        return g[lo:lo + counts[comm.rank]].contiguous(), None, None, None


def dp_gather_rows(local, comm, counts, reduce):
    return _DPGatherRows.apply(local, comm, counts, reduce)


class Adam(object):
    """torch.optim.Adam(params, lr) as the reference builds it (betas (0.9, 0.999), eps 1e-8, no weight decay),
    one fused kernel per tensor; `step(max_norm)` folds clip_grad_norm_(params, max_norm) into the update."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        self.params = [p for p in params]
        self.param_groups = [{'lr': lr, 'betas': betas, 'eps': eps, 'params': self.params}]
        self.state = {}
        self.t = 0
        self.last_grad_norm = None
        self.comm = None       # evalpipe.Comm of a data-parallel run: step() sums the gradients over the ranks first

    def zero_grad(self):
        for p in self.params:
            p.grad = None

    def _sync_grads(self):
        """Data parallel: ONE sum all-reduce of all gradients as a flat bucket (the loss is a sum over the global batch,
        so the global gradient is the sum of the ranks' -- no averaging); p.grad become views of the reduced bucket."""
        import torch.distributed as dist
        ps = [p for p in self.params if p.requires_grad]
        for p in ps:
            if p.grad is None:       # a rank whose shard never touched p still has to take part in the collective
                p.grad = torch.zeros_like(p.data)
        flat = torch.cat([p.grad.reshape(-1) for p in ps])
        self.comm.all_reduce(flat, dist.ReduceOp.SUM)
        o = 0
        for p in ps:
            n = p.numel()
            p.grad = flat[o:o + n].view(p.shape)
            o += n

    def _tables(self, ps):
        """Block maps of the two multi-tensor launches for this set of tensors (sizes only: cached) and the per-step pointer table."""
        lib = _lib.load()
        key = tuple((id(p), p.numel()) for p in ps)
        if getattr(self, '_map_key', None) != key:
            nsq = [lib.itr_sq_sum_blocks(p.numel()) for p in ps]
            nad = [(p.numel() + 255) // 256 for p in ps]
            sq_first = np.concatenate([[0], np.cumsum(nsq)]).astype(np.int64)
            ad_first = np.concatenate([[0], np.cumsum(nad)]).astype(np.int64)
            dev = ps[0].device
            self._maps = dict(nsq=nsq, sq_first=sq_first, n_sq=int(sq_first[-1]), n_ad=int(ad_first[-1]),
                              sq_blk=h2d(np.repeat(np.arange(len(ps)), nsq).astype(np.int32), dev),
                              ad_blk=h2d(np.repeat(np.arange(len(ps)), nad).astype(np.int32), dev),
                              ad_first=h2d(ad_first[:-1].astype(np.int32), dev))
            self._map_key = key
        m = self._maps
        tab = np.zeros((len(ps), 6), dtype=np.int64)          # 48-byte records: p, g, m, v, n, (first_blk | nblk << 32)
        self._keep = []
        for i, p in enumerate(ps):
            st = self.state.get(p)
            if st is None:
                st = self.state[p] = {'exp_avg': torch.zeros_like(p.data), 'exp_avg_sq': torch.zeros_like(p.data)}
            g = p.grad.contiguous()
            self._keep.append(g)
            tab[i] = (p.data.data_ptr(), g.data_ptr(), st['exp_avg'].data_ptr(), st['exp_avg_sq'].data_ptr(), p.numel(),
                      int(m['sq_first'][i]) | (int(m['nsq'][i]) << 32))
        return m, h2d(tab.reshape(-1), ps[0].device)

    def step(self, max_norm=0.0):
        """clip_grad_norm_(params, max_norm) + Adam over every tensor that has a gradient: the squared norms, the clip coefficient and the
        update are three launches for ALL tensors (itr_sq_sum_multi, itr_clip_coef, itr_adam_step_multi)."""
        lib = _lib.load()
        self.t += 1
        g0 = self.param_groups[0]
        if self.comm is not None and self.comm.on:
            self._sync_grads()
        ps = [p for p in self.params if p.grad is not None]
        if not ps:
            return
        for p in ps:
            if not p.is_cuda:
                raise RuntimeError("Adam: parameter on %s (no CPU fallback)" % p.device)
            if not p.data.is_contiguous():
                raise RuntimeError("Adam: non-contiguous parameter")
        dev = ps[0].device
        m, tab = self._tables(ps)
        part = torch.empty(m['n_sq'], device=dev, dtype=torch.float32)
        _lib.check(lib.itr_sq_sum_multi(_p(tab), _p(m['sq_blk']), m['n_sq'], _p(part), _stream()))
        coef = torch.empty(2, device=dev, dtype=torch.float32)
        _lib.check(lib.itr_clip_coef(_p(part), part.numel(), float(max_norm), _p(coef), _stream()))
        self.last_grad_norm = coef[1:]
        _lib.check(lib.itr_adam_step_multi(_p(tab), _p(m['ad_blk']), _p(m['ad_first']), m['n_ad'], float(g0['lr']), float(g0['betas'][0]),
                                           float(g0['betas'][1]), float(g0['eps']), self.t, _p(coef), _stream()))
        self._keep = None
