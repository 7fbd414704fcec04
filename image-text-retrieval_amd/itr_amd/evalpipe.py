"""Row-sharded encode -> score -> rank evaluation (the metric path of utils.validate_step /
evalrank_single: encode_data -> [::5] -> cal_sims -> i2t + t2i; itr/utils.py:144-168,
itr/metricmodule/evaluation.py:75-222), one process per GPU.

Partitioning:
  1. rank p encodes image rows [i0_p, i1_p) -- unique images only, no 5x redundancy -- and a
     contiguous caption slice [c0_p, c1_p);
  2. ONE exchange: all-gather of the packed word (or pooled caption) embeddings over RCCL;
  3. rank p scores its row block S_p = sim(img_p, all captions)  -- no communication;
  4. i2t ranks are row-local; t2i needs the GT score of every caption (max all-reduce of a
     -inf-initialised vector), then partial "greater-than" counts are summed (int32 sum
     all-reduce) and the best-row keys max-reduced.  Integer counts => the result is identical to
     the single-GPU one by construction.
With world_size == 1 every collective is skipped.
"""

import numpy as np
import torch
import torch.distributed as dist

from . import ops
from .settings import SETTINGS

_SIGN = -(1 << 63)  # XOR flips unsigned 64-bit order into signed int64 order for the max all-reduce


def block_range(n, world, rank, align=1):
    """Contiguous, `align`-aligned, near-equal partition of range(n)."""
    per = -(-n // world)
    per = -(-per // align) * align
    lo = min(n, rank * per)
    return lo, min(n, lo + per)


def caption_ranges(n_cap, world, weights=None):
    """Contiguous caption range of every rank.  With `weights` (token count of every caption) the ranges carry near-equal
    token sums -- the text towers' cost is per token and captions are ragged ("balance shards by token count");
    without, near-equal counts (BERT models: every caption is max_words ids, so counts ARE token counts)."""
    if weights is None or world == 1:
        return [block_range(n_cap, world, q) for q in range(world)]
    w = np.asarray(weights, dtype=np.float64)
    if len(w) != n_cap:
        raise ValueError("caption_ranges: %d weights for %d captions" % (len(w), n_cap))
    cs = np.concatenate([[0.0], np.cumsum(w)])
    bounds = [0]
    for q in range(1, world):
        b = int(np.searchsorted(cs, cs[-1] * q / world, side='left'))
        # every rank keeps at least one caption while there are enough of them (a few very long captions, or world close
        # to n_cap, would otherwise leave a rank with an empty range: its text tower has nothing to encode)
        lo = min(bounds[-1] + 1, n_cap)
        hi = max(lo, n_cap - (world - q))
        bounds.append(min(max(b, lo), hi))
    bounds.append(n_cap)
    return [(bounds[q], bounds[q + 1]) for q in range(world)]


class Comm:
    """Minimal collective layer: torch.distributed (backend nccl == RCCL over xGMI on ROCm, gloo in
    the CPU tests) or a no-op for a single process.

    Virtual caption split (`virtual_split="k[:v]"` or SETTINGS.virtual_split, single real rank only): the caption axis is treated as
    owned by k ranks of which this process is owner v.  The images stay whole (one real rank), the k - 1 other owners' embedding
    blocks are handed over in `peer_blocks`, and the exchange is a REAL asynchronous all-gather on the backend's stream (1-rank
    RCCL group when one is initialised, a side-stream copy otherwise) -- so the N > 1 order of work (gather in flight while the
    own columns are scored, wait(), the other owners' columns from the gathered buffer) executes on a 1-GPU box."""

    def __init__(self, group=None, virtual_split=None):
        self.on = dist.is_available() and dist.is_initialized() and (
            dist.get_world_size(group) > 1 or SETTINGS.force_collectives)
        self.group = group
        self.rank = dist.get_rank(group) if self.on else 0
        self.world = dist.get_world_size(group) if self.on else 1
        # gloo moves host memory: device tensors are staged through the host (tests: two ranks sharing ONE GPU, which
        # RCCL refuses; production runs use backend nccl = RCCL over xGMI and never take this branch)
        self.host_staged = self.on and dist.get_backend(group) == "gloo"
        self.cap_world, self.cap_rank = self.world, self.rank        # owners of the caption axis (= the ranks, unless virtual)
        self.peer_blocks = None                                      # virtual split: {owner q: its block}, set by the caller per step
        vs = virtual_split if virtual_split is not None else SETTINGS.virtual_split
        if vs:
            k, _, v = str(vs).partition(":")
            k = int(k)
            v = int(v) if v else k // 2
            if self.world != 1:
                raise ValueError("a virtual caption split needs a single real rank (world is %d)" % self.world)
            if not 0 <= v < k:
                raise ValueError("virtual split %r: owner index out of range" % (vs,))
            if k > 1:
                self.cap_world, self.cap_rank = k, v
        self.virtual = self.cap_world != self.world

    def all_gather_rows(self, local, counts):
        """Concatenate row blocks of different heights.  Returns (buffer [world*maxrows, ...], maxrows):
        rank q's rows live at buffer[q*maxrows : q*maxrows + counts[q]] (padded, no repacking pass)."""
        maxrows = int(max(counts))
        if not self.on:
            return local, maxrows
        pad = torch.zeros((maxrows,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        pad[:local.shape[0]] = local
        out = torch.empty((self.world * maxrows,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        if self.host_staged and local.is_cuda:
            parts = [torch.empty(pad.shape, dtype=pad.dtype) for _ in range(self.world)]
            dist.all_gather(parts, pad.cpu(), group=self.group)
            out.copy_(torch.cat(parts, 0))
        else:
            dist.all_gather_into_tensor(out, pad, group=self.group)
        return out, maxrows

    def all_gather_rows_async(self, send, counts):
        """The one exchange.  `send` is this owner's block ALREADY in a buffer of max(counts) rows (the towers write straight
        into it: no zero-filled staging copy; rows past the owner's count are never read).  Returns (buffer, maxrows, wait):
        owner q's rows live at buffer[q*maxrows : q*maxrows + counts[q]].  With RCCL the collective runs on the backend's own
        stream; kernels launched on the current stream before `wait()` overlap it (the caller scores the captions it already
        holds meanwhile).  gloo (tests) completes immediately."""
        maxrows = int(max(counts))
        if send.shape[0] != maxrows:
            raise ValueError("all_gather_rows_async: the send buffer has %d rows, the largest block %d" % (send.shape[0], maxrows))
        if self.virtual:
            return self._virtual_gather(send, counts, maxrows)
        if not self.on:
            return send, maxrows, (lambda: None)
        out = torch.empty((self.world * maxrows,) + tuple(send.shape[1:]), dtype=send.dtype, device=send.device)
        if self.host_staged and send.is_cuda:
            parts = [torch.empty(send.shape, dtype=send.dtype) for _ in range(self.world)]
            dist.all_gather(parts, send.cpu(), group=self.group)
            out.copy_(torch.cat(parts, 0))
            return out, maxrows, (lambda: None)
        if SETTINGS.exchange == "p2p":
            # Opt-in: the exchange as world - 1 point-to-point sends and receives per rank, batched -- on RCCL one
            # ncclGroup of ncclSend / ncclRecv pairs, i.e. every xGMI link of the fully connected node carries one block at once,
            # instead of whatever algorithm RCCL picks for all_gather.  Not the default: it has only ever run over gloo
            # (tests/test_distributed.py); which of the two is faster on 8 x MI355X is unmeasured.
            out[self.rank * maxrows:(self.rank + 1) * maxrows].copy_(send)
            ops_ = []
            for q in range(self.world):
                if q != self.rank:
                    ops_.append(dist.P2POp(dist.isend, send, q, group=self.group))
                    ops_.append(dist.P2POp(dist.irecv, out[q * maxrows:(q + 1) * maxrows], q, group=self.group))
            reqs = dist.batch_isend_irecv(ops_)

            def wait_p2p(_r=reqs, _k=send):
                for r in _r:
                    r.wait()
            return out, maxrows, wait_p2p
        if dist.get_backend(self.group) != "nccl":
            dist.all_gather_into_tensor(out, send, group=self.group)
            return out, maxrows, (lambda: None)
        work = dist.all_gather_into_tensor(out, send, group=self.group, async_op=True)

        def wait(_w=work, _k=send):     # (the send buffer outlives the collective through the closure)
            _w.wait()                   # current stream waits for the collective; the host does not block
        return out, maxrows, wait

    def _virtual_gather(self, send, counts, maxrows):
        k, v = self.cap_world, self.cap_rank
        if self.peer_blocks is None or set(self.peer_blocks) != set(q for q in range(k) if q != v and counts[q]):
            raise ValueError("virtual split: peer_blocks must hold the block of every other non-empty owner")
        stage = torch.empty((k * maxrows,) + tuple(send.shape[1:]), dtype=send.dtype, device=send.device)
        stage[v * maxrows:(v + 1) * maxrows] = send
        for q, blk in self.peer_blocks.items():
            if blk.shape[0] != counts[q]:
                raise ValueError("virtual split: owner %d hands over %d rows, expected %d" % (q, blk.shape[0], counts[q]))
            stage[q * maxrows:q * maxrows + counts[q]] = blk
        out = torch.empty_like(stage)
        if self.on and dist.get_backend(self.group) == "nccl":
            work = dist.all_gather_into_tensor(out, stage, group=self.group, async_op=True)     # 1 rank: out = stage, on RCCL's stream

            def wait(_w=work, _k=stage):
                _w.wait()
            return out, maxrows, wait
        if not send.is_cuda:               # (CPU tests of the split logic)
            out.copy_(stage)
            return out, maxrows, (lambda: None)
        side = torch.cuda.Stream(device=send.device)
        side.wait_stream(torch.cuda.current_stream(send.device))
        with torch.cuda.stream(side):
            out.copy_(stage, non_blocking=True)
        stage.record_stream(side)

        def wait(_s=side, _d=send.device):
            torch.cuda.current_stream(_d).wait_stream(_s)
        return out, maxrows, wait

    def all_gather_list(self, obj_array):
        """all-gather a small host int array (same length on every rank)."""
        if not self.on:
            return [np.asarray(obj_array)]
        t = torch.as_tensor(np.asarray(obj_array, dtype=np.int64))
        dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(self.group) == "nccl" else t.device
        t = t.to(dev)
        out = [torch.empty_like(t) for _ in range(self.world)]
        dist.all_gather(out, t, group=self.group)
        return [o.cpu().numpy() for o in out]

    def all_reduce(self, t, op):
        if self.on:
            if self.host_staged and t.is_cuda:
                h = t.cpu()
                dist.all_reduce(h, op=op, group=self.group)
                t.copy_(h)
            else:
                dist.all_reduce(t, op=op, group=self.group)
        return t


def finalize_ranks(comm, S_local, row0, n_img_total, im_div=5, rank_fn=None, gather_fn=None):
    """Step 4 above.  Returns host arrays (i2t_rank[Ni], i2t_top1[Ni], t2i_rank[Nc], t2i_top1[Nc]).
    rank_fn / gather_fn default to the HIP kernels; the gloo CPU tests inject reference versions to
    exercise exactly this collective logic."""
    n_local, Nc = S_local.shape
    if rank_fn is None and gather_fn is None and not comm.on and row0 == 0 and n_local * im_div >= Nc:
        # one process holding every row: two launches (preparation + the one pass over S); the library reads the ground-truth scores
        # from S itself and zeroes the column accumulators inside the call -- no gather, no fills
        i_rank, i_top, t_rank, t_best, _ = ops.rank_counts(S_local, im_div)
        return (i_rank.cpu().numpy().astype(np.int64), i_top.cpu().numpy().astype(np.int64), t_rank.cpu().numpy().astype(np.int64),
                (t_best & 0xffffffff).cpu().numpy())
    rank_fn = rank_fn or ops.rank_counts
    gather_fn = gather_fn or ops.gather_gt
    s_gt = torch.full((Nc,), float('-inf'), device=S_local.device, dtype=torch.float32)
    if n_local:
        gather_fn(S_local, im_div, row0, s_gt)
    comm.all_reduce(s_gt, dist.ReduceOp.MAX)
    t_rank = torch.zeros(Nc, device=S_local.device, dtype=torch.int32)
    t_best = torch.zeros(Nc, device=S_local.device, dtype=torch.int64)
    if n_local:
        i_rank, i_top, _, _, _ = rank_fn(S_local, im_div, row0, s_gt, t_rank, t_best)
    else:
        i_rank = torch.zeros(0, device=S_local.device, dtype=torch.int32)
        i_top = torch.zeros(0, device=S_local.device, dtype=torch.int32)
    comm.all_reduce(t_rank, dist.ReduceOp.SUM)
    if comm.on:
        t_best ^= _SIGN
        comm.all_reduce(t_best, dist.ReduceOp.MAX)
        t_best ^= _SIGN
    counts = [block_range(n_img_total, comm.world, q, _IMG_ALIGN)[1] - block_range(n_img_total, comm.world, q, _IMG_ALIGN)[0]
              for q in range(comm.world)]
    both = torch.stack([i_rank, i_top], 1)
    allb, maxrows = comm.all_gather_rows(both, counts)
    allb = allb.cpu().numpy()
    if comm.on:
        allb = np.concatenate([allb[q * maxrows:q * maxrows + counts[q]] for q in range(comm.world)], 0)
    t_top = (t_best & 0xffffffff).cpu().numpy()
    return allb[:, 0].astype(np.int64), allb[:, 1].astype(np.int64), t_rank.cpu().numpy().astype(np.int64), t_top


_IMG_ALIGN = 4  # the SCAN kernel works on 4 images per workgroup


class GruModelEval:
    """Sharded evaluation of the GRU model family (SCAN; VSE++ pooled cosine) on packed inputs."""

    def __init__(self, weights_img, weights_txt, config, comm=None):
        self.wi, self.wt, self.cfg = weights_img, weights_txt, config
        self.comm = comm or Comm()

    # -- step 1
    def encode_images(self, feats_local):
        return ops.proj_l2norm(feats_local, self.wi['fc.weight'], self.wi['fc.bias'],
                               no_imgnorm=self.cfg.get('no_imgnorm', False), use_abs=self.cfg.get('img_use_abs', False))

    def encode_captions(self, tokens_packed, tok_off, lengths_sorted, gather_last=False, out=None):
        return ops.gru_encode(tokens_packed, tok_off, lengths_sorted, self.wt, self.cfg.get('bi_gru', False),
                              no_txtnorm=self.cfg.get('no_txtnorm', False), use_abs=self.cfg.get('txt_use_abs', False),
                              gather_last=gather_last, out=out, batch_invariant=True)     # every partition of the captions: same bits

    # -- whole step for SCAN.  Local inputs:
    #   feats_local [n_img_local, 36, F]          unique images rows [i0, i1)
    #   tokens_packed / tok_off / lengths_sorted  this rank's caption slice, sorted by length (descending)
    #   order_local                               original LOCAL caption index of each sorted position
    #   n_img_total, n_cap_total
    def sgraf_eval(self, sim_weights, feats_local, tokens_packed, tok_off, lengths_sorted, order_local, n_img_total,
                   n_cap_total, im_div=5, timers=None):
        """Same sharding as scan_eval with the SGRAF similarity (EncoderSimilarity) as the scorer."""
        return self.scan_eval(feats_local, tokens_packed, tok_off, lengths_sorted, order_local, n_img_total,
                              n_cap_total, im_div, timers, sgraf_weights=sim_weights)

    def scan_eval(self, feats_local, tokens_packed, tok_off, lengths_sorted, order_local, n_img_total,
                  n_cap_total, im_div=5, timers=None, sgraf_weights=None, cap_ranges=None, all_lengths=None):
        """feats_local: this rank's region features [n_img_local, 36, F] in HBM, or an iterable of row blocks
        (r0, r1, tensor[r1 - r0, 36, F]) covering them in order (evaluate_precomp streams a memory-mapped file that way:
        block k is projected and scored while block k+1 crosses PCIe).
        cap_ranges: the caption range of EVERY caption owner (caption_ranges(); default: equal counts); this rank's inputs
        hold its own range.
        all_lengths: the length of EVERY caption of the evaluation in dataset order (host array).  Every rank of the bench
        and of the file path knows them (synthetic captions are generated, a split is tokenised, on every rank), and every
        owner packs its range in the loader's order (stable sort by length, descending: data_loader.py:146) -- so token counts,
        offsets and lengths of the other owners' blocks follow without any exchange.  Without it the lengths are exchanged
        first (one small device all-gather + host sync before the towers are queued).
        Order of work with several owners: towers (the text tower writes into the head of the max-sized send buffer) -> the
        all-gather of the packed word embeddings is started -> the columns of the captions this rank encoded itself are
        scored while the exchange is in flight -> wait -> the other owners' columns (one launch left and right of the own
        range).  A pair's score depends on that pair only (whatever tile the caption shares with others), so the matrix is
        bit-identical to the single-process one."""
        comm = self.comm
        cfg = self.cfg
        dev = tokens_packed.device
        kw, kr = comm.cap_world, comm.cap_rank
        ranges = cap_ranges or [block_range(n_cap_total, kw, q) for q in range(kw)]
        if len(ranges) != kw:
            raise ValueError("scan_eval: %d caption ranges for %d caption owners" % (len(ranges), kw))
        cap_counts = [hi - lo for lo, hi in ranges]
        # -- host metadata: lengths / offsets of this rank's captions in their original order
        lens_sorted = np.asarray(lengths_sorted, dtype=np.int64)
        off_sorted = np.concatenate([[0], np.cumsum(lens_sorted)[:-1]]) if len(lens_sorted) else np.zeros(0, np.int64)
        n_loc = len(lens_sorted)
        if n_loc != cap_counts[kr]:
            raise ValueError("scan_eval: %d local captions, range of owner %d holds %d" % (n_loc, kr, cap_counts[kr]))
        order_local = np.asarray(order_local, dtype=np.int64)
        len_loc = np.zeros(n_loc, np.int64)
        off_loc = np.zeros(n_loc, np.int64)
        len_loc[order_local] = lens_sorted       # back to the original caption order
        off_loc[order_local] = off_sorted
        n_tok = int(lens_sorted.sum())
        if kw == 1:
            tok_counts, cap_len, offs = [n_tok], len_loc, [off_loc]
        elif all_lengths is not None:
            cap_len = np.asarray(all_lengths, dtype=np.int64)
            c0, c1 = ranges[kr]
            if len(cap_len) != n_cap_total or not np.array_equal(cap_len[c0:c1], len_loc):
                raise ValueError("scan_eval: all_lengths does not match this rank's captions")
            if not np.array_equal(order_local, np.argsort(-len_loc, kind="stable")):
                raise ValueError("scan_eval: with all_lengths every owner packs its captions by stable descending length")
            tok_counts, offs = [], []
            for lo, hi in ranges:
                lq = cap_len[lo:hi]
                oq = np.argsort(-lq, kind="stable")
                ls = lq[oq]
                off_q = np.zeros(hi - lo, np.int64)
                off_q[oq] = np.cumsum(ls) - ls
                offs.append(off_q)
                tok_counts.append(int(lq.sum()))
        else:
            if comm.virtual:
                raise ValueError("scan_eval: a virtual caption split needs all_lengths")
            maxcap = max(cap_counts)
            meta = np.zeros(2 * maxcap + 1, np.int64)
            meta[0] = n_tok
            meta[1:1 + n_loc] = len_loc
            meta[1 + maxcap:1 + maxcap + n_loc] = off_loc
            metas = comm.all_gather_list(meta)
            tok_counts = [int(m[0]) for m in metas]
            cap_len = np.concatenate([m[1:1 + cap_counts[q]] for q, m in enumerate(metas)])
            offs = [m[1 + maxcap:1 + maxcap + cap_counts[q]] for q, m in enumerate(metas)]
        maxtok = max(tok_counts)
        # -- step 1: towers (text first: it needs no image, and the first feature block may still be on its way).  The word
        # embeddings land in the head of the send buffer of the exchange (maxtok rows).  The tail (at most a few rows: the owners'
        # token counts are balanced) is zeroed: it is shipped to every rank, and a consumer that reduces over all the gathered rows
        # (e.g. the tensor absmax of the opt-in split-precision variants) must not see uninitialised memory.
        D_emb = self.wt['rnn.weight_hh_l0'].shape[1]
        send = torch.empty(maxtok, D_emb, device=dev, dtype=torch.float32)
        if maxtok > n_tok:
            send[n_tok:].zero_()
        words = self.encode_captions(tokens_packed, tok_off, lengths_sorted, out=send[:n_tok])
        if torch.is_tensor(feats_local):
            blocks = [(0, feats_local.shape[0], feats_local)]
            n_img_local = feats_local.shape[0]
        else:
            blocks = feats_local
            n_img_local = block_range(n_img_total, comm.world, comm.rank, _IMG_ALIGN)
            n_img_local = n_img_local[1] - n_img_local[0]
        # -- step 2: start the one exchange
        words_all, maxtok, wait = comm.all_gather_rows_async(send, tok_counts)
        cap_off = np.concatenate([offs[q] + q * maxtok for q in range(kw)])
        xa = cfg.get('cross_attn', 't2i')
        # The own / left / right launches pack the captions into OTHER column tiles than one launch over all of them.  A
        # pair's score does not depend on the tile its caption sits in for SCAN t2i and SGRAF (every reduction runs in an
        # order relative to the caption's first word: bit-identical matrices, asserted by the sharded tests) -- but the i2t
        # kernel sums over a caption's words with indicator MFMAs whose association order follows the caption's column inside
        # the tile (differences of 1 ulp, a handful of near-tie rank flips at 5k x 25k).  i2t therefore waits for the exchange
        # and scores all columns in one launch from the same plan as a single process: identical ranks, the exchange
        # (~2 ms of a 130 ms step at 8 ranks) not hidden.
        overlap = kw > 1 and (sgraf_weights is not None or xa != 'i2t')

        if timers is not None:
            timers['segments'] = []      # (start, end) HIP events around every scoring launch of this step
        plans = {}

        def score(img, key, words_t, off, lens, out):
            if key not in plans:                       # the column-tile plan of a caption set is reused by every row block
                plans[key] = ops.ScanPlan(off, lens, words_t.shape[0], dev)
            plan_ = plans[key]
            ws = None if sgraf_weights is not None else ops.scan_prepare(img, words_t, plan_, xa)   # tile packing, Gram matrices: not the kernel timed
            if timers is not None:
                ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                ev[0].record()
            if sgraf_weights is not None:
                ops.sgraf_scores(img, words_t, plan_, sgraf_weights, cfg.get('module_name', 'SAF'), cfg.get('sgr_step', 3), out=out)
            else:
                ops.scan_xattn_scores(img, words_t, plan_, cross_attn=xa, raw_feature_norm=cfg.get('raw_feature_norm', 'clipped_l2norm'),
                                      agg_func=cfg.get('agg_func', 'LogSumExp'), lambda_lse=cfg.get('lambda_lse', 6.0),
                                      lambda_softmax=cfg.get('lambda_softmax', 9.0), out=out, workspace=ws,
                                      precision=cfg.get('scan_precision', 'fp32'))    # 'bf16x3': opt-in study variant (STUDY_SPLIT_PRECISION.md)
            if timers is not None:
                ev[1].record()
                timers['segments'].append(ev)
            return plan_

        # -- step 3: local row block(s).  One launch per row block when there is nothing to wait for; else the first block
        # scores this rank's own columns first
        S = torch.empty(n_img_local, n_cap_total, device=dev, dtype=torch.float32)
        plan = None
        waited = False
        for r0, r1, fblock in blocks:
            img = self.encode_images(fblock)
            if not overlap or waited:
                if not waited:
                    wait()
                    waited = True
                plan = score(img, 'all', words_all, cap_off, cap_len, S[r0:r1])
            else:
                c0, c1 = ranges[kr]
                if c1 > c0:
                    plan = score(img, 'own', words, off_loc, len_loc, S[r0:r1, c0:c1])
                if timers is not None:
                    # how long the current stream still waits for the exchange AFTER the own-column launch has finished: ~0 when the
                    # collective made progress under the scoring grid, ~ its stand-alone time when it was starved of CUs
                    ex = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                    ex[0].record()
                wait()
                if timers is not None:
                    ex[1].record()
                    timers['exchange_wait'] = ex
                    timers['exchange_bytes'] = int(words_all.numel()) * 4
                waited = True
                for key, lo, hi in (('left', 0, c0), ('right', c1, n_cap_total)):
                    if hi > lo:
                        plan = score(img, key, words_all, cap_off[lo:hi], cap_len[lo:hi], S[r0:r1, lo:hi])
        if not waited:
            wait()
        if timers is not None and timers['segments']:
            timers['scan_start'], timers['scan_end'] = timers['segments'][-1]    # (one launch: the kernel; several: the last one)
        # -- step 4
        row0 = block_range(n_img_total, comm.world, comm.rank, _IMG_ALIGN)[0]
        ranks = finalize_ranks(comm, S, row0, n_img_total, im_div)
        return S, ranks, plan


def exchange_score(comm, img, send, ranges, n_cap_total, score_fn, timers=None):
    """Steps 2 + 3 for the models whose caption embedding is ONE vector: the all-gather of the caption embeddings is started
    (`send`: this owner's block in a buffer of max-count rows), the columns of the owner's own captions are scored meanwhile,
    then -- after wait() -- the other owners' columns, one launch per run of adjacent full blocks (equal counts: one launch left
    and one right of the own range).  score_fn(img, cap_rows, out=column block of S).  -> S [n_img_local, n_cap_total]."""
    kw, kr = comm.cap_world, comm.cap_rank
    counts = [hi - lo for lo, hi in ranges]
    cap_all, maxrows, wait = comm.all_gather_rows_async(send, counts)
    n_img_local = img.shape[0]
    S = torch.empty(n_img_local, n_cap_total, device=img.device, dtype=torch.float32)
    if timers is not None:
        timers['scan_start'].record()
    if kw == 1:
        wait()
        if n_img_local and n_cap_total:
            score_fn(img, cap_all[:n_cap_total], out=S)
    else:
        c0, c1 = ranges[kr]
        if n_img_local and c1 > c0:
            score_fn(img, send[:c1 - c0], out=S[:, c0:c1])
        wait()
        runs = []                          # [first row of cap_all, row count, first column]
        for q in range(kw):
            if q == kr or counts[q] == 0:
                continue
            if runs and runs[-1][0] + runs[-1][1] == q * maxrows and runs[-1][2] + runs[-1][1] == ranges[q][0]:
                runs[-1][1] += counts[q]
            else:
                runs.append([q * maxrows, counts[q], ranges[q][0]])
        for r0, n, col in runs:
            if n_img_local:
                score_fn(img, cap_all[r0:r0 + n], out=S[:, col:col + n])
    if timers is not None:
        timers['scan_end'].record()
    return S


def score_rank_streamed(img, cap, score_fn, im_div=5, budget_bytes=64 << 20, rows_per_block=None, timers=None):
    """One process, pooled scorers: the Recall ranks of score_fn(img, cap) WITHOUT the similarity matrix in HBM (SURVEY 7 step 3, 2.3 K4
    "fuse with K9"; the reference builds the full float64 matrix and argsorts it, evaluation.py:124-153, :169, :209).

    The matrix is produced in row blocks of <= budget_bytes (64 MB: a quarter of the 256 MB Infinity Cache) into ONE reused buffer and
    every block is ranked at once by the one-pass count kernel, whose column accumulators add up over row blocks by design (the same
    call the row-sharded evaluation makes per rank): the ranker reads what the GEMM has just written from the cache, and the next
    block overwrites it there -- the 0.5 GB write and read-back of the 5k x 25k matrix never reach HBM as such.  The ground-truth
    scores the column counts compare against come from a first pass over the diagonal band (block rows x their im_div x rows captions:
    1 / 20 of the products at 5k x 25k), computed by the same GEMM kernels in the same k order: bit-identical to the band's elements
    of the full rows (tests/test_kernels_gpu.py checks the ranks against the materialised matrix, ties included).
    MEASURED (round 6, same box): slower than materialising -- SAEM 5k x 25k score + rank 1.54 ms against 0.99, CAMERA 58.9 ms against
    51.4: the band is 5 rows_per_block / Nc = 13 % more products at 640-row blocks and eight launches have eight tails, while the
    0.5 GB round trip it saves is 0.25 ms at HBM speed.  So this is an opt-in (PooledModelEval.eval(stream_scores=True),
    bench.py --stream-scores), for a process that cannot afford the 0.5 GB; the default writes the matrix once.
    -> the tuple finalize_ranks returns."""
    Ni, Nc = img.shape[0], cap.shape[0]
    dev = img.device
    rb = rows_per_block or max(128, (budget_bytes // (4 * max(Nc, 1))) // 128 * 128)
    if timers is not None:
        timers['scan_start'].record()
    buf = torch.empty(min(rb, Ni) * Nc, device=dev, dtype=torch.float32)
    s_gt = torch.full((Nc,), float('-inf'), device=dev, dtype=torch.float32)
    for r0 in range(0, Ni, rb):                      # pass 1: the band -> ground-truth score of every caption
        r1 = min(Ni, r0 + rb)
        c0, c1 = min(Nc, r0 * im_div), min(Nc, r1 * im_div)
        if c1 > c0:
            band = score_fn(img[r0:r1], cap[c0:c1], out=buf[:(r1 - r0) * (c1 - c0)].view(r1 - r0, c1 - c0))
            ops.gather_gt(band, im_div, 0, s_gt[c0:c1])
    t_rank = torch.zeros(Nc, device=dev, dtype=torch.int32)
    t_best = torch.zeros(Nc, device=dev, dtype=torch.int64)
    i_rank, i_top = [], []
    for r0 in range(0, Ni, rb):                      # pass 2: a row block of scores, ranked while it is still in the cache
        r1 = min(Ni, r0 + rb)
        blk = score_fn(img[r0:r1], cap, out=buf[:(r1 - r0) * Nc].view(r1 - r0, Nc))
        ir, it, _, _, _ = ops.rank_counts(blk, im_div, r0, s_gt, t_rank, t_best)
        i_rank.append(ir)
        i_top.append(it)
    if timers is not None:
        timers['scan_end'].record()
    both = torch.stack([torch.cat(i_rank), torch.cat(i_top)], 1).cpu().numpy() if i_rank else np.zeros((0, 2), np.int64)
    return both[:, 0].astype(np.int64), both[:, 1].astype(np.int64), t_rank.cpu().numpy().astype(np.int64), (t_best & 0xffffffff).cpu().numpy()


class PooledModelEval:
    """Sharded evaluation of the models whose caption embedding is ONE vector (VSE++, SAEM, CAMERA):
    rank p encodes its image rows and its caption slice in batches, ONE all-gather moves the caption embeddings
    (Nc x D floats: 25.6 MB SAEM, 205 MB CAMERA at coco size), the row block of the similarity matrix is one GEMM
    (cosine / pdist_cos / MultiViewMatching with its max-over-views epilogue) and the ranks are finished exactly as
    for SCAN (finalize_ranks)."""

    def __init__(self, model, comm=None, batch=4096):
        self.model, self.comm, self.batch = model, comm or Comm(), batch
        self.name = model.config['name']

    def _score(self, img, cap, out=None):
        if self.name == 'CAMERA':
            return ops.mvm_scores(img, cap, out=out)        # Fusionmodule.py:674-692
        if self.name == 'SAEM':
            return ops.pdist_cos(img, cap, out=out)         # Objectives.py:310-323
        return ops.cosine_scores(img, cap, out=out)         # Objectives.py:18-21

    def encode_images(self, images, boxes, imgs_wh):
        m, bs = self.model, self.batch
        imgs = []
        with torch.no_grad():
            for b0 in range(0, images.shape[0], bs):
                sl = slice(b0, b0 + bs)
                if self.name == 'CAMERA':
                    imgs.append(m.img_enc(images[sl], boxes[sl], imgs_wh[sl])[0])
                else:
                    imgs.append(m.img_enc(images[sl]))
        return torch.cat(imgs, 0)

    def encode_captions(self, captions, captions_mask, captions_type_ids, lengths, rows=None):
        """-> caption embeddings [n, D...]; with `rows` >= n the result is the head of a buffer of `rows` rows (the send
        buffer of the exchange: the batches are written straight into it)."""
        m, bs = self.model, self.batch
        n = captions.shape[0]
        buf = None
        with torch.no_grad():
            for b0 in range(0, n, bs):
                sl = slice(b0, b0 + bs)
                if self.name == 'CAMERA':
                    e = m.txt_enc(captions[sl], captions_mask[sl], captions_type_ids[sl])
                elif self.name == 'SAEM':
                    e = m.txt_enc(captions[sl], captions_mask[sl], captions_type_ids[sl], lengths[b0:b0 + bs])
                else:
                    e = m.txt_enc(captions[sl], lengths[b0:b0 + bs])[0]
                if buf is None:
                    buf = torch.empty((max(rows or n, n),) + tuple(e.shape[1:]), device=e.device, dtype=e.dtype)
                buf[b0:b0 + e.shape[0]] = e
        if buf is None:
            raise ValueError("PooledModelEval.encode_captions: an empty caption shard (every owner holds at least one caption)")
        return buf[:n] if rows is None else buf

    def encode(self, images, boxes, imgs_wh, captions, captions_mask, captions_type_ids, lengths):
        """Local shards -> (img_emb, cap_emb); image and caption counts are independent here (unique images)."""
        return self.encode_images(images, boxes, imgs_wh), self.encode_captions(captions, captions_mask, captions_type_ids, lengths)

    def eval(self, images, boxes, imgs_wh, captions, captions_mask, captions_type_ids, lengths, n_img_total, n_cap_total,
             im_div=5, timers=None, cap_emb=None, cap_ranges=None, stream_scores=False):
        """cap_emb: this owner's caption embeddings when the caller has run the text tower already -- its count rows, or
        a buffer of max-count rows whose head they are.  stream_scores=True (one process, opt-in): only the ranks are wanted -- the
        similarity matrix is streamed through the ranker in cache-sized row blocks and never stored (score_rank_streamed); S is None
        then.  Not the default: measured SLOWER than writing the matrix once (profiles/r06/NOTES.md)."""
        comm = self.comm
        kw, kr = comm.cap_world, comm.cap_rank
        ranges = cap_ranges or [block_range(n_cap_total, kw, q) for q in range(kw)]
        counts = [hi - lo for lo, hi in ranges]
        maxrows = max(counts)
        if cap_emb is None:
            send = self.encode_captions(captions, captions_mask, captions_type_ids, lengths, rows=maxrows)
        elif cap_emb.shape[0] == maxrows:
            send = cap_emb
        else:
            send = torch.empty((maxrows,) + tuple(cap_emb.shape[1:]), device=cap_emb.device, dtype=cap_emb.dtype)
            send[:cap_emb.shape[0]] = cap_emb
            send[cap_emb.shape[0]:].zero_()       # (the tail travels with the exchange: defined values)
        img = self.encode_images(images, boxes, imgs_wh)
        if stream_scores and not comm.on and not comm.virtual and img.shape[0] == n_img_total and 4 * n_img_total * n_cap_total > (64 << 20):
            return None, score_rank_streamed(img, send[:n_cap_total], self._score, im_div, timers=timers)
        S = exchange_score(comm, img, send, ranges, n_cap_total, self._score, timers)
        row0 = block_range(n_img_total, comm.world, comm.rank, _IMG_ALIGN)[0]
        return S, finalize_ranks(comm, S, row0, n_img_total, im_div)


# ---------------------------------------------------------------------------------------------------------
# Real data: checkpointed model + precomp files -> sharded, device-resident evaluation (the bench's path on a dataset)
_PINNED = {}
_PINNED_LOCK = __import__("threading").Lock()


def _pinned_pair(shape):
    """Two page-locked staging buffers of `shape`, taken from a process-wide pool and handed back by `_pinned_release`
    (cudaHostAlloc of tens of MB is expensive: the buffers are kept for the process).  A pair belongs to ONE _stage_rows call
    at a time -- a prefetch on the helper thread and a copy on the main thread never share staging memory."""
    key = tuple(shape)
    with _PINNED_LOCK:
        pool = _PINNED.setdefault(key, [])
        if pool:
            return pool.pop()
    return [torch.empty(shape, dtype=torch.float32).pin_memory() for _ in range(2)]


def _pinned_release(shape, pair):
    with _PINNED_LOCK:
        _PINNED.setdefault(tuple(shape), []).append(pair)


def _stage_rows(arr, r0, r1, dst, stream, chunk=128):
    """Rows [r0, r1) of a memory-mapped .npy -> dst (HBM, r1 - r0 rows) through two pinned staging buffers; the H2D copies
    run on `stream`, so reading chunk k+1 from the page cache overlaps the DMA of chunk k.  Returns the event of the last copy."""
    ev = [None, None]
    last = None
    shape = (chunk,) + tuple(arr.shape[1:])
    pair = _pinned_pair(shape)
    try:
        for k, c0 in enumerate(range(r0, r1, chunk)):
            c1 = min(r1, c0 + chunk)
            b = k & 1
            buf = pair[b]
            if ev[b] is not None:
                ev[b].synchronize()                  # the copy that last used this staging buffer is done
            np.copyto(buf[:c1 - c0].numpy(), arr[c0:c1], casting='same_kind')
            with torch.cuda.stream(stream):
                dst[c0 - r0:c1 - r0].copy_(buf[:c1 - c0], non_blocking=True)
                ev[b] = torch.cuda.Event()
                ev[b].record(stream)
                last = ev[b]
    finally:
        for e in ev:
            if e is not None:
                e.synchronize()                      # the staging buffers go back to the pool idle
        _pinned_release(shape, pair)
    return last


def _features_to_device(arr, i0, i1, dev, chunk=128):
    """Rows [i0, i1) of a memory-mapped .npy -> one HBM tensor."""
    out = torch.empty((i1 - i0,) + tuple(arr.shape[1:]), device=dev, dtype=torch.float32)
    stream = torch.cuda.Stream(device=dev)
    # `out` may be a block the caching allocator just recycled from kernels still queued on the current stream (validation
    # right after an asynchronous training step): the copies must not overtake them
    stream.wait_stream(torch.cuda.current_stream(dev))
    if i1 > i0:
        _stage_rows(arr, i0, i1, out, stream, chunk)
    torch.cuda.current_stream(dev).wait_stream(stream)
    return out


class _FeatureBlocks:
    """Rows [i0, i1) of a memory-mapped feature file as a sequence of HBM row blocks (r0, r1, tensor) for scan_eval:
    two device buffers; block k+1 is staged (page cache -> pinned -> HBM on a side stream) after the consumer has QUEUED its
    kernels for block k, so the copy runs under them -- the 1.47 GB of the MS-COCO 5k test features never sit exposed on the
    PCIe link except for the first block, whose staging `prefetch()` starts on a helper thread while the captions are
    tokenised."""

    def __init__(self, arr, i0, i1, dev, block_rows=640):
        self.arr, self.i0, self.i1, self.dev = arr, i0, i1, dev
        self.block = max(_IMG_ALIGN, block_rows // _IMG_ALIGN * _IMG_ALIGN)
        n = min(self.block, max(i1 - i0, 0))
        self.stream = torch.cuda.Stream(device=dev)
        self.stream.wait_stream(torch.cuda.current_stream(dev))
        self.bufs = [torch.empty((n,) + tuple(arr.shape[1:]), device=dev, dtype=torch.float32) for _ in range(2)]
        self.consumed = [None, None]     # event on the consumer's stream: kernels reading buffer b are queued up to here
        self._thread, self._first, self._error = None, None, None

    def _stage(self, k):
        r0 = self.i0 + k * self.block
        r1 = min(self.i1, r0 + self.block)
        b = k & 1
        if self.consumed[b] is not None:
            self.stream.wait_event(self.consumed[b])       # do not overwrite a block the GPU is still reading
        return _stage_rows(self.arr, r0, r1, self.bufs[b][:r1 - r0], self.stream)

    def prefetch(self):
        """Stage block 0 on a helper thread (numpy / torch copies release the GIL)."""
        import threading
        if self.i1 > self.i0 and self._thread is None:
            def run():
                try:
                    torch.cuda.set_device(self.dev)
                    self._first = self._stage(0)
                except BaseException as e:       # (mmap read error, pin_memory OOM, HIP error): re-raised on the consumer's thread
                    self._error = e
            self._thread = threading.Thread(target=run)
            self._thread.start()

    def __iter__(self):
        cur = torch.cuda.current_stream(self.dev)
        n_blocks = -(-(self.i1 - self.i0) // self.block) if self.i1 > self.i0 else 0
        for k in range(n_blocks):
            if k == 0 and self._thread is not None:
                self._thread.join()
                if self._error is not None:
                    raise self._error
                ev = self._first
            else:
                ev = self._stage(k)
            cur.wait_event(ev)
            r0 = k * self.block
            r1 = min(self.i1 - self.i0, r0 + self.block)
            yield r0, r1, self.bufs[k & 1][:r1 - r0]
            self.consumed[k & 1] = cur.record_event()      # the consumer has queued its work on this block


def evaluate_precomp(model, dataset, comm=None, fold=None, batch=4096, block_rows=640):
    """Recall ranks of `model` on a PrecompDataset (datamodule.data_loader), one process per GPU.

    Unlike encode_data + cal_sims (the reference-shaped path: 5 x redundant image encodes, host numpy arrays, a Python
    tile loop), every unique image is encoded once, nothing returns to the host but the rank vectors, the caption
    axis is sharded over ranks and exchanged with ONE all-gather, and the row block of the similarity matrix is scored
    by the fused kernels.  fold = (k, size) restricts to captions [k*size, (k+1)*size) (MS-COCO 1k folds,
    evaluation.py:296-300).  block_rows: images per streamed feature block of the word-level models (see _FeatureBlocks).
    Returns (i2t_rank, i2t_top1, t2i_rank, t2i_top1) as host int64 arrays."""
    comm = comm or Comm()
    if comm.virtual:
        raise ValueError("evaluate_precomp: the virtual caption split is a test hook of the resident-input path (bench.py --virtual-split)")
    cfg = model.config
    name = cfg['name']
    dev = torch.device('cuda', torch.cuda.current_device())
    im_div = dataset.im_div
    n_cap_all = len(dataset)
    cap_lo, cap_hi = (0, n_cap_all) if fold is None else (fold[0] * fold[1], min(n_cap_all, (fold[0] + 1) * fold[1]))
    n_cap = cap_hi - cap_lo
    img_lo = cap_lo // im_div
    n_img = -(-n_cap // im_div)
    if im_div != 5 or n_cap % 5:
        raise NotImplementedError("evaluate_precomp expects the 5-captions-per-image layout of the precomp test splits")
    i0, i1 = block_range(n_img, comm.world, comm.rank, _IMG_ALIGN)
    model.val_start()
    streamed = name in ('SCAN', 'SGRAF')       # word-level scorers: seconds of GPU work per row block to hide the copies under
    if streamed:
        feats = _FeatureBlocks(dataset.images, img_lo + i0, img_lo + i1, dev, block_rows)
        feats.prefetch()                       # block 0 crosses PCIe while the captions are tokenised
    elif name not in ('SAEM', 'CAMERA'):
        feats = _features_to_device(dataset.images, img_lo + i0, img_lo + i1, dev)
    with torch.no_grad():
        if name in ('SCAN', 'SGRAF', 'VSE++', 'VSE_PP', 'VSRN'):
            if name == 'VSRN':
                # VSRN: the reference's padded caption layout (every caption max_len + 1 ids, PrecompDataset.vsrn_ids): equal lengths
                ranges = caption_ranges(n_cap, comm.cap_world)
                c0, c1 = ranges[comm.cap_rank]
                ids = [dataset.vsrn_ids(cap_lo + j)[0] for j in range(c0, c1)]
                lens = np.asarray([len(x) for x in ids], np.int64)
                flat = np.concatenate([np.asarray(x, np.int64) for x in ids]) if len(ids) else np.zeros(0, np.int64)
            else:
                # every rank tokenises the split once (one regex pass, cached in the dataset) and so knows ALL lengths: the
                # caption ranges are balanced by token count without any exchange
                flat_all, lens_all = dataset.token_ids_range(cap_lo, cap_hi)
                ranges = caption_ranges(n_cap, comm.cap_world, lens_all)
                c0, c1 = ranges[comm.cap_rank]
                offs_all = np.concatenate([[0], np.cumsum(lens_all)])
                lens = lens_all[c0:c1]
                flat = flat_all[offs_all[c0]:offs_all[c1]]
            # sort this rank's captions by length (descending, stable -- pack_padded_sequence's order) and re-pack
            order = np.argsort(-lens, kind="stable")
            lens_sorted = [int(lens[i]) for i in order]
            src_off = np.concatenate([[0], np.cumsum(lens)[:-1]]) if len(lens) else np.zeros(0, np.int64)
            if len(order):
                ls = lens[order]
                dst_off = np.cumsum(ls) - ls
                packed = flat[np.repeat(src_off[order] - dst_off, ls) + np.arange(int(ls.sum()))]
            else:
                packed = np.zeros(0, np.int64)
            tok_off = np.concatenate([[0], np.cumsum(lens_sorted)[:-1]]) if len(order) else np.zeros(0, np.int64)
            toks, off = ops.h2d(packed, dev, torch.int64), ops.h2d(tok_off.astype(np.int64), dev, torch.int64)
            wi = {k: v.detach() for k, v in model.img_enc.state_dict().items()}
            if hasattr(model.img_enc, '_weight'):      # precomp_enc_type='weight_norm' stores fc.weight_g / fc.weight_v only
                wi['fc.weight'] = model.img_enc._weight().detach()
            wt = {k: v.detach() for k, v in model.txt_enc.state_dict().items()}
            # the towers' own switches (order embeddings: use_abs on both towers, ImgEncoder.py:143-145, TextEncoder.py:66-68)
            ev = GruModelEval(wi, wt, dict(cfg, bi_gru=model.txt_enc.use_bi_gru, no_txtnorm=model.txt_enc.no_txtnorm,
                                           no_imgnorm=model.img_enc.no_imgnorm, img_use_abs=getattr(model.img_enc, 'use_abs', False),
                                           txt_use_abs=getattr(model.txt_enc, 'use_abs', False)), comm)
            if name in ('VSE++', 'VSE_PP', 'VSRN'):
                if name == 'VSRN':       # GCN + region GRU tower (ImgEncoder.py:199-231), `batch` images per pass
                    img = torch.cat([model.img_enc(feats[b0:b0 + batch])[0] for b0 in range(0, feats.shape[0], batch)], 0) \
                        if feats.shape[0] else torch.zeros(0, cfg['embed_size'], device=dev)
                else:
                    img = ev.encode_images(ops.mean_mid(feats))
                cap_sorted = ev.encode_captions(toks, off, lens_sorted, gather_last=True)
                counts = [hi - lo for lo, hi in ranges]
                send = torch.empty(max(counts), cap_sorted.shape[1], device=dev, dtype=torch.float32)
                send[torch.from_numpy(np.ascontiguousarray(order)).to(dev)] = cap_sorted      # dataset order, head of the send buffer
                send[cap_sorted.shape[0]:].zero_()                                           # (the tail travels with the exchange)
                # ranked with the similarity the model was trained for: criterion.sim = cosine_sim or order_sim
                # (Objectives.py:45-50; the reference's cal_sims calls model.criterion.sim, evaluation.py:128-131)
                if cfg.get('measure', 'cosine') == 'order':
                    fn = ops.order_scores
                elif cfg.get('measure', 'cosine') == 'cosine':
                    fn = ops.cosine_scores
                else:
                    raise ValueError("unknown measure:", cfg.get('measure'))
                S = exchange_score(comm, img, send, ranges, n_cap, fn)
                return finalize_ranks(comm, S, i0, n_img, im_div)
            sw = {k: v.detach() for k, v in model.sim_enc.state_dict().items()} if name == 'SGRAF' else None
            _, ranks, _ = ev.scan_eval(feats, toks, off, lens_sorted, order, n_img, n_cap, im_div, sgraf_weights=sw, cap_ranges=ranges,
                                       all_lengths=lens_all)
            return ranks
        # ---- BERT models: one vector per caption
        if name not in ('SAEM', 'CAMERA'):
            raise NotImplementedError("evaluate_precomp: model %r" % name)
        c0, c1 = block_range(n_cap, comm.world, comm.rank)     # every caption is max_words ids: equal counts = equal tokens
        max_count = block_range(n_cap, comm.world, 0)[1]
        ids_np, mask_np, types_np = dataset.bert_features_range(cap_lo + c0, cap_lo + c1)
        ids, mask, types = (ops.h2d(a, dev, torch.long).reshape(-1, dataset.max_words) for a in (ids_np, mask_np, types_np))
        pe = PooledModelEval(model, comm, batch=batch)
        lens_b = [int(v) for v in mask_np.sum(1)]
        # text tower first: its ~second of queued GPU work (12 BERT layers) covers the host-side staging of the features
        cap = pe.encode_captions(ids, mask, types, lens_b, rows=max_count)      # (the send buffer of the exchange)
        feats = _features_to_device(dataset.images, img_lo + i0, img_lo + i1, dev)
        boxes = wh = None
        if name == 'CAMERA':
            boxes = _features_to_device(dataset.boxes, img_lo + i0, img_lo + i1, dev)
            wh = _features_to_device(dataset.img_wh, img_lo + i0, img_lo + i1, dev)
        _, ranks = pe.eval(feats, boxes, wh, None, None, None, lens_b, n_img, n_cap, im_div, cap_emb=cap)
        return ranks
