"""SGRAF.train_emb's similarity module on all (image, caption) pairs of the batch at once: differentiable wrappers of the ragged
kernels of csrc/sgraf_train.hip (EncoderSimilarity.forward in training mode, itr/modalmodule/Fusionmodule.py:406-451).

Layout (image-major, ragged over the captions; B images, C captions, T words in total, W_c words in caption c):
    local rows   b T + t                         attention weights P [B T, R], squared context difference X [B T, D], sim_loc [B T, S]
    node rows    b (T + C) + cap_off[c] + c + j  j = 0 the global alignment, j = 1 .. W_c the local ones           nodes [B (T + C), S]
    pair rows    b C + c                         sim_glo, sim_vec, the similarity matrix itself
torch.autograd is the tape only; every forward and backward is a call into libitr_hip.so.  No CPU path."""
import numpy as np
import torch

from . import _lib
from .ops import _dev, _p, _stream, h2d


def _f32(*shape, dev):
    return torch.empty(*shape, device=dev, dtype=torch.float32)


class Layout(object):
    """Device tables of one batch of captions (lengths in caption order)."""

    def __init__(self, lens, device):
        lens = np.asarray([int(x) for x in lens], dtype=np.int64)
        if lens.size == 0 or int(lens.min()) < 1:
            raise ValueError("sgraf_train.Layout: captions must have at least one word")
        self.lens = lens
        self.C = int(lens.size)
        off = np.concatenate([[0], np.cumsum(lens)])
        self.T = int(off[-1])
        self.NT = self.T + self.C
        self.Wmax = int(lens.max())
        self.nmax = self.Wmax + 1
        cap = np.arange(self.C)
        e_off = np.concatenate([[0], np.cumsum((lens + 1) ** 2)])
        self.E_per_image = int(e_off[-1])
        self.node0_col = off[:-1] + cap                                        # node column of the global alignment of caption c
        self.cap_off = h2d(off.astype(np.int32), device)
        self.e_off = h2d(e_off.astype(np.int32), device)
        self.node_cap = h2d(np.repeat(cap, lens + 1).astype(np.int32), device)
        self.device = device

    def node0_rows(self, B):
        """Rows of the global-alignment nodes, pair order (b, c)."""
        rows = (np.arange(B, dtype=np.int64)[:, None] * self.NT + self.node0_col[None, :]).reshape(-1)
        return h2d(rows, self.device)


def supported(D, R, S, lens):
    """Shapes the batched kernels take (csrc/sgraf_train.hip); beyond them SGRAF trains through the grouped path."""
    wmax = max(int(x) for x in lens)
    return D % 4 == 0 and D <= 2048 and R <= 64 and S % 4 == 0 and wmax <= 96 and (3 * R * wmax + R) * 4 <= 150 * 1024


class _LocAttn(torch.autograd.Function):
    """SCAN_attention's weights (Fusionmodule.py:641-652): A [B R, T] raw region . word products -> P [B T, R]."""

    @staticmethod
    def forward(ctx, A, lay, B, R, smooth):
        lib = _lib.load()
        A = _dev(A, name="A")
        P = _f32(B * lay.T, R, dev=A.device)
        _lib.check(lib.itr_sgt_attn_fwd(_p(A), A.stride(0), _p(lay.cap_off), B, lay.C, lay.T, R, lay.Wmax, float(smooth), 1e-8, _p(P), _stream()))
        ctx.save_for_backward(A, P)
        ctx.lay, ctx.B, ctx.R, ctx.smooth = lay, B, R, float(smooth)
        return P

    @staticmethod
    def backward(ctx, dP):
        lib = _lib.load()
        A, P = ctx.saved_tensors
        lay = ctx.lay
        dA = torch.empty_like(A)
        _lib.check(lib.itr_sgt_attn_bwd(_p(A), A.stride(0), _p(P), _p(dP.contiguous()), _p(lay.cap_off), ctx.B, lay.C, lay.T, ctx.R, lay.Wmax,
                                        ctx.smooth, 1e-8, _p(dA), _stream()))
        return dA, None, None, None, None


def loc_attn(A, lay, B, R, smooth=9.0):
    return _LocAttn.apply(A, lay, B, R, smooth)


class _LocCtx(torch.autograd.Function):
    """X[(b, t)] = (l2norm(sum_r P[(b, t), r] img[b, r]) - words[t])^2 (Fusionmodule.py:654-662, :426); the context itself is never stored."""

    @staticmethod
    def forward(ctx, P, img, words):
        lib = _lib.load()
        P, img, words = _dev(P, name="P"), _dev(img, name="img"), _dev(words, name="words")
        B, R, D = img.shape
        T = words.shape[0]
        X = _f32(B * T, D, dev=img.device)
        cnorm = _f32(B * T, dev=img.device)
        _lib.check(lib.itr_sgt_ctx_fwd(_p(P), _p(img), _p(words), B, T, R, D, 1e-8, _p(X), _p(cnorm), _stream()))
        ctx.save_for_backward(P, img, words, cnorm)
        return X

    @staticmethod
    def backward(ctx, dX):
        lib = _lib.load()
        P, img, words, cnorm = ctx.saved_tensors
        B, R, D = img.shape
        T = words.shape[0]
        dev = img.device
        dctx = _f32(B * T, D, dev=dev)
        dwords = _f32(T, D, dev=dev)
        wsb = lib.itr_sgt_ctx_bwd_workspace_bytes(B, T, D)
        ws = torch.empty(max(wsb, 1), device=dev, dtype=torch.uint8)
        _lib.check(lib.itr_sgt_ctx_bwd(_p(P), _p(img), _p(words), _p(cnorm), _p(dX.contiguous()), B, T, R, D, 1e-8, _p(dctx), _p(dwords), _p(ws),
                                       wsb, _stream()))
        dP = _f32(B * T, R, dev=dev)
        _lib.check(lib.itr_sgt_dp(_p(dctx), _p(img), B, T, R, D, _p(dP), _stream()))
        dimg = _f32(B, R, D, dev=dev)           # d regions[b] = P_b^T [R, T] . d ctx_b [T, D]
        _lib.check(lib.itr_gemm_tn_batched(_p(P), R, T * R, _p(dctx), D, T * D, _p(dimg), D, R * D, T, R, D, B, _stream()))
        return dP, dimg, dwords


def loc_ctx(P, img, words):
    return _LocCtx.apply(P, img, words)


class _PairSqdiff(torch.autograd.Function):
    """(img_glo[b] - cap_glo[c])^2 for every pair, rows b C + c (Fusionmodule.py:429)."""

    @staticmethod
    def forward(ctx, ig, cg):
        lib = _lib.load()
        ig, cg = _dev(ig, name="img_glo"), _dev(cg, name="cap_glo")
        B, D = ig.shape
        C = cg.shape[0]
        X = _f32(B * C, D, dev=ig.device)
        _lib.check(lib.itr_sgt_pair_sqdiff_fwd(_p(ig), _p(cg), B, C, D, _p(X), _stream()))
        ctx.save_for_backward(ig, cg)
        return X

    @staticmethod
    def backward(ctx, dX):
        lib = _lib.load()
        ig, cg = ctx.saved_tensors
        dig, dcg = torch.empty_like(ig), torch.empty_like(cg)
        _lib.check(lib.itr_sgt_pair_sqdiff_bwd(_p(ig), _p(cg), _p(dX.contiguous()), ig.shape[0], cg.shape[0], ig.shape[1], _p(dig), _p(dcg), _stream()))
        return dig, dcg


def pair_sqdiff(img_glo, cap_glo):
    return _PairSqdiff.apply(img_glo, cap_glo)


class _Nodes(torch.autograd.Function):
    """torch.cat([sim_glo.unsqueeze(1), sim_loc], 1) of every pair (Fusionmodule.py:433) into the ragged node matrix."""

    @staticmethod
    def forward(ctx, glo, loc, lay, B):
        lib = _lib.load()
        glo, loc = _dev(glo, name="sim_glo"), _dev(loc, name="sim_loc")
        S = glo.shape[1]
        nodes = _f32(B * lay.NT, S, dev=glo.device)
        _lib.check(lib.itr_sgt_nodes(_p(glo), _p(loc), _p(nodes), _p(lay.cap_off), _p(lay.node_cap), B, lay.C, lay.T, S, 0, _stream()))
        ctx.lay, ctx.B = lay, B
        return nodes

    @staticmethod
    def backward(ctx, dnodes):
        lib = _lib.load()
        lay, B = ctx.lay, ctx.B
        dnodes = dnodes.contiguous()
        S = dnodes.shape[1]
        dglo, dloc = _f32(B * lay.C, S, dev=dnodes.device), _f32(B * lay.T, S, dev=dnodes.device)
        _lib.check(lib.itr_sgt_nodes(_p(dglo), _p(dloc), _p(dnodes), _p(lay.cap_off), _p(lay.node_cap), B, lay.C, lay.T, S, 1, _stream()))
        return dglo, dloc, None, None


def assemble_nodes(sim_glo, sim_loc, lay, B):
    return _Nodes.apply(sim_glo, sim_loc, lay, B)


class _GraphAttn(torch.autograd.Function):
    """softmax(q k^T) x inside every pair's graph (GraphReasoning.forward, Fusionmodule.py:582-584)."""

    @staticmethod
    def forward(ctx, q, k, x, lay, B, row0):
        lib = _lib.load()
        q, k, x = _dev(q, name="q"), _dev(k, name="k"), _dev(x, name="x")
        S = x.shape[1]
        if q.shape[0] != (B * lay.C if row0 else B * lay.NT) or k.shape[0] != B * lay.NT or x.shape[0] != B * lay.NT:
            raise ValueError("graph_attn: q %s, k %s, x %s for %d images, %d captions, %d words" % (tuple(q.shape), tuple(k.shape), tuple(x.shape), B,
                                                                                                  lay.C, lay.T))
        E = _f32(B * (lay.NT if row0 else lay.E_per_image), dev=x.device)
        Z = torch.empty_like(q)
        _lib.check(lib.itr_sgt_graph_fwd(_p(q), _p(k), _p(x), _p(lay.cap_off), _p(lay.e_off), B, lay.C, lay.T, S, lay.nmax, int(row0), _p(E), _p(Z),
                                         _stream()))
        ctx.save_for_backward(q, k, x, E)
        ctx.lay, ctx.B, ctx.row0 = lay, B, int(row0)
        return Z

    @staticmethod
    def backward(ctx, dZ):
        lib = _lib.load()
        q, k, x, E = ctx.saved_tensors
        lay = ctx.lay
        dq, dk, dx = torch.empty_like(q), torch.empty_like(k), torch.empty_like(x)
        _lib.check(lib.itr_sgt_graph_bwd(_p(q), _p(k), _p(x), _p(E), _p(dZ.contiguous()), _p(lay.cap_off), _p(lay.e_off), ctx.B, lay.C, lay.T,
                                         x.shape[1], lay.nmax, ctx.row0, _p(dq), _p(dk), _p(dx), _stream()))
        return dq, dk, dx, None, None, None


def graph_attn(q, k, x, lay, B, row0=False):
    """row0: q [B C, S] holds only the query of node 0 of every pair (the last reasoning step) -> Z [B C, S]."""
    return _GraphAttn.apply(q, k, x, lay, B, row0)


class _SegBN(torch.autograd.Function):
    """AttentionFiltration's BatchNorm1d(1) in training mode, batch statistics per CAPTION (the reference calls the module once per
    caption, Fusionmodule.py:436-441): a [B (T + C)] attention logits in node order -> y; stats_out receives (mean, biased var) [C]."""

    @staticmethod
    def forward(ctx, a, gamma, beta, lay, B, eps, stats_out):
        lib = _lib.load()
        a = _dev(a, name="a")
        y = torch.empty_like(a)
        mean, var, invstd = (_f32(lay.C, dev=a.device) for _ in range(3))
        _lib.check(lib.itr_sgt_segbn_fwd(_p(a), _p(lay.cap_off), B, lay.C, lay.T, _p(gamma), _p(beta), float(eps), _p(y), _p(mean), _p(var), _p(invstd),
                                         _stream()))
        ctx.save_for_backward(a, gamma, mean, invstd)
        ctx.lay, ctx.B = lay, B
        stats_out.append((mean, var))
        return y

    @staticmethod
    def backward(ctx, dy):
        from .autograd import colsum
        lib = _lib.load()
        a, gamma, mean, invstd = ctx.saved_tensors
        lay = ctx.lay
        da = torch.empty_like(a)
        dg, db = _f32(lay.C, 1, dev=a.device), _f32(lay.C, 1, dev=a.device)
        _lib.check(lib.itr_sgt_segbn_bwd(_p(dy.contiguous()), _p(a), _p(lay.cap_off), ctx.B, lay.C, lay.T, _p(gamma), _p(mean), _p(invstd), _p(da), _p(dg),
                                         _p(db), _stream()))
        return da, colsum(dg), colsum(db), None, None, None, None


class _SafPool(torch.autograd.Function):
    """l1norm(sigmoid(y)) over the nodes of a pair, then the weighted node sum (Fusionmodule.py:614-616) -> [B C, S]."""

    @staticmethod
    def forward(ctx, y, nodes, lay, B):
        lib = _lib.load()
        y, nodes = _dev(y, name="y"), _dev(nodes, name="nodes")
        S = nodes.shape[1]
        out = _f32(B * lay.C, S, dev=nodes.device)
        _lib.check(lib.itr_sgt_saf_pool_fwd(_p(y), _p(nodes), _p(lay.cap_off), B, lay.C, lay.T, S, lay.nmax, 1e-8, _p(out), _stream()))
        ctx.save_for_backward(y, nodes)
        ctx.lay, ctx.B = lay, B
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        y, nodes = ctx.saved_tensors
        lay = ctx.lay
        dy, dnodes = torch.empty_like(y), torch.empty_like(nodes)
        _lib.check(lib.itr_sgt_saf_pool_bwd(_p(y), _p(nodes), _p(dout.contiguous()), _p(lay.cap_off), ctx.B, lay.C, lay.T, nodes.shape[1], lay.nmax, 1e-8,
                                            _p(dy), _p(dnodes), _stream()))
        return dy, dnodes, None, None


def saf_pool(y, nodes, lay, B):
    return _SafPool.apply(y, nodes, lay, B)


class _SegReduce(torch.autograd.Function):
    """words [T, D] -> the mean (or sum) over each caption's rows [C, D]."""

    @staticmethod
    def forward(ctx, x, lay, mean):
        lib = _lib.load()
        x = _dev(x, name="x")
        out = _f32(lay.C, x.shape[1], dev=x.device)
        _lib.check(lib.itr_sgt_seg_mean(_p(x), _p(lay.cap_off), lay.C, x.shape[1], _p(out), 0, int(mean), _stream()))
        ctx.lay, ctx.mean = lay, int(mean)
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        lay = ctx.lay
        dout = dout.contiguous()
        dx = _f32(lay.T, dout.shape[1], dev=dout.device)
        _lib.check(lib.itr_sgt_seg_mean(_p(dout), _p(lay.cap_off), lay.C, dout.shape[1], _p(dx), 1, ctx.mean, _stream()))
        return dx, None, None


class _SegSpread(torch.autograd.Function):
    """[C, D] -> [T, D]: every word row gets its caption's row (g_emb.unsqueeze(1).repeat, Fusionmodule.py:556); backward = the caption sums,
    in a fixed order (no atomics: two runs of a step give the same bits)."""

    @staticmethod
    def forward(ctx, x, lay):
        lib = _lib.load()
        x = _dev(x, name="x")
        out = _f32(lay.T, x.shape[1], dev=x.device)
        _lib.check(lib.itr_sgt_seg_mean(_p(x), _p(lay.cap_off), lay.C, x.shape[1], _p(out), 1, 0, _stream()))
        ctx.lay = lay
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        lay = ctx.lay
        dout = dout.contiguous()
        dx = _f32(lay.C, dout.shape[1], dev=dout.device)
        _lib.check(lib.itr_sgt_seg_mean(_p(dout), _p(lay.cap_off), lay.C, dout.shape[1], _p(dx), 0, 0, _stream()))
        return dx, None


def seg_mean(words, lay):
    return _SegReduce.apply(words, lay, True)


def seg_spread(x, lay):
    return _SegSpread.apply(x, lay)


class _SegSmry(torch.autograd.Function):
    """softmax(logit over the words of a caption) . words -> [C, D] (TextSA.forward, Fusionmodule.py:558-562)."""

    @staticmethod
    def forward(ctx, logit, words, lay):
        lib = _lib.load()
        logit, words = _dev(logit, name="logit"), _dev(words, name="words")
        D = words.shape[1]
        p = torch.empty_like(logit)
        out = _f32(lay.C, D, dev=words.device)
        _lib.check(lib.itr_sgt_seg_smry_fwd(_p(logit), _p(words), _p(lay.cap_off), lay.C, D, lay.Wmax, _p(p), _p(out), _stream()))
        ctx.save_for_backward(p, words)
        ctx.lay = lay
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        p, words = ctx.saved_tensors
        lay = ctx.lay
        dlogit, dwords = torch.empty_like(p), torch.empty_like(words)
        _lib.check(lib.itr_sgt_seg_smry_bwd(_p(p), _p(words), _p(dout.contiguous()), _p(lay.cap_off), lay.C, words.shape[1], lay.Wmax, _p(dlogit),
                                            _p(dwords), _stream()))
        return dlogit, dwords, None


def seg_smry(logit, words, lay):
    return _SegSmry.apply(logit, words, lay)


def seg_bn_train(a, bn, lay, B):
    """BatchNorm1d(1) of AttentionFiltration over the node logits, one set of batch statistics per caption, and the running statistics
    after C sequential momentum updates in caption order (closed form of the reference's per-caption calls)."""
    stats = []
    y = _SegBN.apply(a, bn.weight, bn.bias, lay, B, bn.eps, stats)
    mean, var = stats[0]
    with torch.no_grad():
        C = lay.C
        N = torch.as_tensor(B * (lay.lens + 1), dtype=torch.float32)
        unbias = h2d((N / torch.clamp(N - 1, min=1)).numpy(), a.device)
        var_u = var * unbias
        if bn.momentum is None:          # cumulative moving average
            for i in range(C):
                f = 1.0 / float(int(bn.num_batches_tracked) + i + 1)
                bn.running_mean.mul_(1 - f).add_(mean[i:i + 1], alpha=f)
                bn.running_var.mul_(1 - f).add_(var_u[i:i + 1], alpha=f)
        else:
            m = float(bn.momentum)
            w = h2d((m * (1.0 - m) ** np.arange(C - 1, -1, -1, dtype=np.float64)).astype(np.float32), a.device)
            keep = (1.0 - m) ** C
            bn.running_mean.mul_(keep).add_((w * mean).sum())
            bn.running_var.mul_(keep).add_((w * var_u).sum())
        bn.num_batches_tracked += C
    return y
