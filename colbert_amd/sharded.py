"""Doc-sharded multi-GPU rerank (no counterpart in the reference, whose rerank is single-GPU).

The index is partitioned by contiguous pid range, one shard per rank (one process per GPU).  Every rank scores
the candidates that fall in its range and takes a local top-k; ONE all_gather of ``[nq, k]`` scores + global pids
(RCCL over xGMI when the backend is ``nccl``; 12 B x nq x k per rank -- latency-bound) is followed by a per-query
``world*k -> k`` merge on every rank.  There is no other collective on this path.

Parity with the UNSHARDED reference: the reference derives its length-bucket strides from the doclens of the whole
index (colbert_ranker.py:36-40) and a doc's 0-floor depends on them (:90, :108-109), so every shard must bucket by the
GLOBAL strides, not by the percentiles of its own docs: ``global_strides`` (one all_reduce of a doclen histogram at
load time) and ``ColbertRanker(strides=...)`` / ``set_strides``; ``ShardedRanker`` does this by itself.

``score_fn`` / ``topk_fn`` are injectable so the partition/gather/merge logic can be exercised on CPU ranks
(``gloo``) in tests with the oracle as scorer; the product default is the HIP path of the local ``ColbertRanker``.
"""
import torch
import torch.distributed as dist

from . import _lib
from .ranker import strides_from_histogram

NEG_INF = float("-inf")


def shard_range(n_docs_total, rank, world):
    """Contiguous pid range [lo, hi) of a rank (same structure as the reference's per-part files, loaders.py:7-19): the
    balanced cut -- shard sizes differ by at most one doc, and no shard is empty as long as the index has at least
    ``world`` docs (with ceil(n / world)-sized shards the last ranks of a small index got nothing: 12 docs on 8 ranks)."""
    return (rank * n_docs_total) // world, ((rank + 1) * n_docs_total) // world


def localize(cand_global, lo, hi):
    """Global candidate pids -> local pids of this shard; out-of-range entries become -1 (padding slots)."""
    inr = (cand_global >= lo) & (cand_global < hi)
    return torch.where(inr, cand_global - lo, torch.full_like(cand_global, -1)), inr


def shard_candidates(cand_global, lo, hi, with_counts=False):
    """This shard's candidates moved to the front of every row, in list order, as (local pids, global pids), both
    ``[nq, ncand]`` with -1 in the tail; ``with_counts`` adds the per-row live count (int32 [nq]).  The row width is NOT
    cut and nothing is read back (no host sync): the counts stay on the device, where the rerank builds its work list
    from them (``maxsim_rerank_counted``).  Device tensors run ``maxsim_shard_candidates``; CPU tensors (the gloo tests,
    whose scorer is injected) the same stable partition in torch."""
    nq, ncand = cand_global.shape
    if cand_global.is_cuda:
        cg = cand_global.to(torch.int64).contiguous()
        loc, gp = torch.empty_like(cg), torch.empty_like(cg)
        cnt = torch.empty(nq, dtype=torch.int32, device=cg.device)
        with torch.cuda.device(cg.device):
            rc = _lib.lib.maxsim_shard_candidates(cg.data_ptr(), nq, ncand, int(lo), int(hi), loc.data_ptr(), gp.data_ptr(),
                                                  cnt.data_ptr(), torch.cuda.current_stream(cg.device).cuda_stream)
        _lib.check(rc, "maxsim_shard_candidates")
        return (loc, gp, cnt) if with_counts else (loc, gp)
    loc, inr = localize(cand_global, lo, hi)
    gp = torch.where(inr, cand_global, torch.full_like(cand_global, -1))
    order = torch.argsort((~inr).to(torch.int8), dim=1, stable=True)
    loc, gp = torch.gather(loc, 1, order), torch.gather(gp, 1, order)
    return (loc, gp, inr.sum(1).to(torch.int32)) if with_counts else (loc, gp)


def global_strides(local_doclens, group=None, device=None):
    """The reference's length-bucket strides (colbert_ranker.py:36-40: 25/50/75th percentile by ``kthvalue`` and the
    maximum) of the WHOLE index from each rank's local doclens: an all_reduce(MAX) of the largest doclen, then an
    all_reduce(SUM) of the doclen histogram; the k-th smallest value is read off the cumulative histogram -- exact.
    ``device``: where the collective's tensors live (a CUDA device for the ``nccl`` backend).  Without an initialised
    process group this is the single-index rule."""
    dl = torch.as_tensor(local_doclens, dtype=torch.int64)
    distributed = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
    mx = torch.tensor([int(dl.max().item()) if dl.numel() else 0], dtype=torch.int64)
    if distributed:
        if device is None and dist.get_backend(group) == "nccl":
            device = torch.device("cuda", torch.cuda.current_device())
        mx = mx.to(device) if device is not None else mx
        dist.all_reduce(mx, op=dist.ReduceOp.MAX, group=group)
    hist = torch.bincount(dl, minlength=int(mx.item()) + 1)
    if distributed:
        hist = hist.to(device) if device is not None else hist
        dist.all_reduce(hist, op=dist.ReduceOp.SUM, group=group)
    return strides_from_histogram(hist.cpu())


def assert_strides_agree(strides, group=None, device=None):
    """Fails loudly when the ranks of a doc-sharded index do not bucket by the same strides (a shard built with
    ``sync_strides=False`` from its own percentiles, a rank that loaded another index): the 0-floor -- and therefore
    the scores -- would silently differ from the unsharded reference (colbert_ranker.py:90, :108-109).  One small
    all_reduce(MAX) of (strides, -strides) that EVERY rank takes part in: a rank whose own list is malformed (more than
    the reference's four strides, :36-40) sends a sentinel length instead of raising on its own, so that all ranks leave
    the collective and all of them raise."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    st = sorted(int(x) for x in strides)
    too_many = len(st) > 4
    v = torch.tensor([5, 0, 0, 0, 0] if too_many else [len(st)] + st + [0] * (4 - len(st)), dtype=torch.int64)
    both = torch.cat([v, -v])
    if device is None and dist.get_backend(group) == "nccl":
        device = torch.device("cuda", torch.cuda.current_device())
    both = both.to(device) if device is not None else both
    dist.all_reduce(both, op=dist.ReduceOp.MAX, group=group)
    both = both.cpu()
    if int(both[0]) > 4:
        raise ValueError(f"at most 4 distinct strides (colbert_ranker.py:36-40); this rank has {st}")
    if not torch.equal(both[:5], -both[5:]):
        raise RuntimeError(f"doc shards disagree on the length-bucket strides: this rank has {st}, the ranks' maxima are "
                           f"{both[1:5].tolist()} and minima {(-both[6:]).tolist()} -- every shard must use the strides "
                           f"of the WHOLE index (sharded.global_strides)")


def all_gather_topk(top_s, top_p, world, group=None):
    """The path's ONE exchange step: every rank's [nq, k] scores and global pids -> [world, nq, k] on every rank.
    One collective of 12 bytes per entry: per query k int64 pids as 2k int32 words, then the k fp32 scores
    as their bit patterns."""
    nq, k = top_s.shape
    dev = top_s.device
    payload = torch.empty(nq, 3 * k, dtype=torch.int32, device=dev)
    payload[:, :2 * k] = top_p.contiguous().view(torch.int32)
    payload[:, 2 * k:] = top_s.contiguous().view(torch.int32)
    if dev.type == "cuda" and dist.get_backend(group) == "gloo":
        # rehearsal on a one-GPU box (RCCL refuses two ranks on one device): stage through the host
        payload = payload.cpu()
    g = torch.empty(world * nq, 3 * k, dtype=torch.int32, device=payload.device)
    dist.all_gather_into_tensor(g, payload, group=group)
    g = g.view(world, nq, 3 * k).to(dev)
    gp = g[..., :2 * k].contiguous().view(torch.int64)
    gs = g[..., 2 * k:].contiguous().view(torch.float32)
    return gs, gp


def merge_gathered(all_scores, all_pids, k, topk_fn):
    """[world, nq, k] gathered local top-k -> global top-k per query."""
    world, nq, kk = all_scores.shape
    s = all_scores.permute(1, 0, 2).reshape(nq, world * kk).contiguous()
    p = all_pids.permute(1, 0, 2).reshape(nq, world * kk).contiguous()
    return topk_fn(s, p, min(k, world * kk))


class ShardedRanker:
    """One rank's view of a doc-sharded index: ``local_ranker`` holds the docs with global pids [lo, hi).

    ``load_shard`` builds one straight from the reference's index files.  Built by hand (``ShardedRanker(local, lo, hi)``)
    every rank must construct it at the same point: with more than one rank the constructor runs small collectives --
    two all_reduces for the strides of the whole index (``sync_strides=True``) or none (``False``: the caller set them),
    ONE all_reduce that checks that all ranks bucket alike (always, also with ``sync_strides=False``), and one all_gather
    of (lo, docs, token rows) unless ``n_docs_total`` and ``tok_lo`` are both given.

    n_docs_total : docs of the WHOLE index (``rank_forward``'s pid range check and negative-pid wrap, colbert_ranker.py:88)
    tok_lo       : global token row of this shard's first token (``retrieve_batch`` takes GLOBAL token rows, as the ANN
                   index over the whole collection returns them, colbert_ranker.py:176-181)
    score_fn / topk_fn / pids_fn : injectable so that the partition / gather / merge logic runs on CPU ranks (``gloo``
                   tests) with the oracle as scorer; the product default is the HIP path of the local ``ColbertRanker``
    """

    def __init__(self, local_ranker, lo, hi, group=None, score_fn=None, topk_fn=None, sync_strides=True, *,
                 n_docs_total=None, tok_lo=None, pids_fn=None):
        self.local = local_ranker
        self.lo, self.hi = int(lo), int(hi)
        self.group = group
        self.score_fn = score_fn if score_fn is not None else local_ranker.score_candidates
        self.topk_fn = topk_fn if topk_fn is not None else local_ranker.topk
        self.pids_fn = pids_fn        # None: the local ranker's embedding_ids_to_pids (keep-mask and id_base applied in the kernel)
        # the product scorer / top-k take the per-row live counts (counted rows: maxsim_rerank_counted / maxsim_topk_counted);
        # injected ones (CPU tests) get the plain signature unless they say otherwise
        self.score_counted = score_fn is None
        self.topk_counted = topk_fn is None
        self.force_exchange = False   # diagnostic: run the exchange + merge even at world size 1 (bench.py --force-dist)
        self.exchange_events = None   # diagnostic: a list here collects (start, stop) HIP events of every exchange + merge
        world = self._world()
        dev = None
        if hasattr(local_ranker, "device") and local_ranker.device.type == "cuda":
            dev = local_ranker.device
        # bucket by the strides of the whole index (see the module docstring)
        if hasattr(local_ranker, "set_strides") and world > 1:
            if sync_strides:
                local_ranker.set_strides(global_strides(local_ranker.doclens, group, dev))
            # sync_strides=False (the caller set the strides itself) is still checked: ranks that bucket differently
            # would return scores that differ from the unsharded reference without any error
            assert_strides_agree(local_ranker.strides, group, dev)
        self.n_docs_total = None if n_docs_total is None else int(n_docs_total)
        self.tok_lo = None if tok_lo is None else int(tok_lo)
        n_tok_local = getattr(local_ranker, "num_embeddings", None)
        if world == 1:
            self.n_docs_total = self.hi if self.n_docs_total is None else self.n_docs_total
            self.tok_lo = 0 if (self.tok_lo is None and self.lo == 0) else self.tok_lo
        elif (self.n_docs_total is None or self.tok_lo is None) and n_tok_local is not None:
            # where the shards sit in the whole index: every rank's (lo, docs, token rows)
            mine = torch.tensor([self.lo, self.hi - self.lo, int(n_tok_local)], dtype=torch.int64)
            mine = mine.to(dev) if (dev is not None and dist.get_backend(group) == "nccl") else mine
            rows = torch.empty(world * 3, dtype=torch.int64, device=mine.device)
            dist.all_gather_into_tensor(rows, mine, group=group)
            rows = rows.view(world, 3).cpu()
            if self.n_docs_total is None:
                self.n_docs_total = int((rows[:, 0] + rows[:, 1]).max().item())
            if self.tok_lo is None:
                self.tok_lo = int(rows[rows[:, 0] < self.lo, 2].sum().item())
        self.tok_hi = None if (self.tok_lo is None or n_tok_local is None) else self.tok_lo + int(n_tok_local)

    def local_topk(self, Q, cand_global, depth, q_len=None, q_mask=None):
        """Scores this shard's share of every query's GLOBAL candidate list and returns its local top-k with global pids:
        (pids [nq,k], scores [nq,k]); slots beyond a query's local candidates are (-1, -inf)."""
        k = min(int(depth), cand_global.size(1))
        cand_local, gp, cnt = shard_candidates(cand_global, self.lo, self.hi, with_counts=True)
        kw = {}
        if q_len is not None:
            kw["q_len"] = q_len
        if q_mask is not None:
            kw["q_mask"] = q_mask
        if self.score_counted:
            kw["cand_count"] = cnt
        scores = self.score_fn(Q, cand_local, **kw)
        gp = gp.to(scores.device)
        if self.topk_counted:
            return self.topk_fn(scores, gp, k, cnt)
        return self.topk_fn(scores, gp, k)

    def _world(self):
        if not (dist.is_available() and dist.is_initialized()):
            return 1
        return dist.get_world_size(self.group)

    def rerank_batch(self, Q, cand_global, depth=10, q_len=None, q_mask=None):
        """Every rank passes the same Q [nq,Lq,h] and global candidate lists [nq,ncand]; returns the global
        top-``depth`` (pids, scores) on every rank."""
        top_p, top_s = self.local_topk(Q, cand_global, depth, q_len, q_mask)
        world = self._world()
        if world == 1:
            return top_p, top_s
        gs, gp = all_gather_topk(top_s, top_p, world, self.group)
        return merge_gathered(gs, gp, int(depth), self.topk_fn)

    def exchange_async(self, top_p, top_s, depth):
        """The exchange + merge of one batch on the ranker's side stream, so that the NEXT batch's rerank kernel (on the
        caller's stream) runs while this batch's all_gather crosses xGMI.  Returns a handle; ``handle.result()`` makes the
        caller's stream wait for the merge and returns (pids, scores).  Collectives are issued in call order on every
        rank, so handles must be created in the same order everywhere."""
        return _Exchange(self, top_p, top_s, int(depth))


    # ------------------------------------------------------------------------------------------
    def local_retrieve_topk(self, Q, keep, embedding_ids, k):
        """This shard's leg of ``retrieve_batch``: GLOBAL token rows [bs, n] (-1 = dropped) -> the rows inside this shard's
        token range as local rows -> distinct local pids (counted rows) -> counted rerank with ``keep`` as the per-token
        predicate -> counted local top-k with GLOBAL pids; (pids [bs,k], scores [bs,k]), (-1, -inf) behind a short row."""
        if self.tok_lo is None or self.tok_hi is None:
            raise ValueError("retrieve_batch needs tok_lo (the global token row of this shard's first token)")
        if self.pids_fn is None:        # colbert_ranker.py:212-229 on this shard's rows: the kernel drops foreign rows
            cand, counts = self.local.embedding_ids_to_pids(embedding_ids, trim=False, keep=keep, id_base=self.tok_lo)
        else:                           # injected (CPU ranks): local rows, -1 for foreign rows and dropped tokens' neighbours
            ids = embedding_ids.reshape(embedding_ids.size(0), keep.size(1), -1).masked_fill(~keep.unsqueeze(-1), -1)
            ids = ids.reshape(embedding_ids.size(0), -1)
            inside = (ids >= self.tok_lo) & (ids < self.tok_hi)
            cand, counts = self.pids_fn(torch.where(inside, ids - self.tok_lo, torch.full_like(ids, -1)))
        kw = {"q_mask": keep}
        if self.score_counted:
            kw["cand_count"] = counts
        scores = self.score_fn(Q, cand, **kw)
        top_p, top_s = self.topk_fn(scores, cand, k, counts) if self.topk_counted else self.topk_fn(scores, cand, k)
        return torch.where(top_p >= 0, top_p + self.lo, top_p), top_s

    def retrieve_batch(self, Q, q_active_padding, topk, embedding_ids=None, ann_search=None, faiss_depth=None):
        """The batched driver (``colbert_amd.retrieve_batch``; reference: dense_server_client.py:44-48 +
        faiss_indexers.py:224-235 + colbert_ranker.py:176-181, :212-229) on the doc-sharded index.  Every rank passes the
        same Q [bs, Lq, h], q_active_padding [bs, Lq] and the same ANN result: GLOBAL token rows of the whole collection
        (``embedding_ids`` [bs, Lq, faiss_depth], or ``ann_search``).  Each rank keeps the rows inside its token range,
        reranks its distinct docs and takes a local top-k; ONE all_gather; every rank merges and returns the reference's
        per-query ``(pids, scores)`` lists -- equal on every rank, equal to the unsharded driver's."""
        from .retriever import prepare_embedding_ids, unpack_topk_lists
        dev = self.local.device
        Q = Q.to(dev)
        keep, ids = prepare_embedding_ids(dev, Q, q_active_padding, embedding_ids, ann_search, faiss_depth, mask_ids=False)
        bs = Q.size(0)
        ids = ids.reshape(bs, -1)
        k = min(int(topk), ids.size(1))
        top_p, top_s = self.local_retrieve_topk(Q, keep, ids, k)
        world = self._world()
        if world > 1 or (self.force_exchange and dist.is_initialized()):
            gs, gp = all_gather_topk(top_s, top_p, world, self.group)
            top_p, top_s = merge_gathered(gs, gp, k, self.topk_fn)
        return unpack_topk_lists(top_p, top_s, (top_p >= 0).sum(1, dtype=torch.int32))

    def rank_forward(self, Q, pids, views=None, depth=10, output_D_embedding=False):
        """``ColbertRanker.rank_forward`` (colbert_ranker.py:75-137) on the doc-sharded index: Q [1, h, Lq] (dim-major, as
        faiss_indexers.py:232-233 hands it over), ``pids`` a list or tensor of GLOBAL pids, the same on every rank ->
        ``(pids, scores)`` lists sorted by score descending, at most ``depth`` long, on every rank."""
        assert len(pids) > 0                                                  # :76
        assert Q.size(0) in [1, len(pids)]                                    # :77
        if Q.size(0) != 1:
            raise NotImplementedError("rank_forward with one query per candidate is not exercised by the reference")
        dev = self.local.device
        n_total = self.n_docs_total
        if n_total is None:
            raise ValueError("rank_forward needs n_docs_total (the docs of the WHOLE index: the pid range check and the "
                             "negative-pid wrap of colbert_ranker.py:88)")
        pids_t = (torch.tensor(pids, dtype=torch.int64) if type(pids) is list else pids.to(torch.int64)).view(1, -1)
        lo, hi = (int(x) for x in torch.aminmax(pids_t))
        if hi >= n_total or lo < -n_total:                                     # `self.doclens[pids]`, :88
            raise IndexError(f"index {hi if hi >= n_total else lo} is out of bounds for dimension 0 with size {n_total}")
        pids_t = pids_t.to(dev)
        cand = torch.where(pids_t < 0, pids_t + n_total, pids_t) if lo < 0 else pids_t   # torch indexing wraps, :88
        Qt = Q.permute(0, 2, 1)                                               # :111 -> [1, Lq, h]
        top_p, top_s = self.rerank_batch(Qt, cand, depth=min(int(depth), cand.size(1)))
        out_p, out_s = top_p[0].tolist(), top_s[0].tolist()
        if output_D_embedding:                                                # :131-136
            D, mask = self._output_D(top_p[0], cand[0])
        if lo < 0:
            # :129 returns the caller's own values (`pids[order]`): hand every wrapped pid back as it was passed in
            # (a doc listed twice, as p and p - N, has one score; which of the two spellings comes first is a tie)
            back = {}
            for orig, c in zip(pids_t[0].tolist(), cand[0].tolist()):
                back.setdefault(c, []).append(orig)
            out_p = [back[c].pop(0) for c in out_p]
        if output_D_embedding:
            return out_p, D, mask
        return out_p, out_s

    def _reduce(self, t, op):
        """all_reduce of a small tensor over the shard group (staged through the host when CUDA tensors meet gloo: the
        one-GPU rehearsal)."""
        if self._world() == 1:
            return t
        stage = t.is_cuda and dist.get_backend(self.group) == "gloo"
        x = t.cpu() if stage else t
        dist.all_reduce(x, op=op, group=self.group)
        return x.to(t.device) if stage else x

    def _output_D(self, top_pids, all_cand):
        """colbert_ranker.py:131-136 on the doc-sharded index: D [k, S, h] fp32 and mask [k, S] of the top docs as the
        reference's strided view hands them over (:49, :105) -- slot t of a doc is token row offset + t of the CONCATENATED
        index whatever doc it belongs to: slots past a doc's end hold the next docs' tokens, which for the last docs of a shard
        live on the NEXT rank, and zeros past the end of the whole index (the reference's +512-row tail, :62).  Three small
        all_reduces: the candidates' length bucket (the reference's ``torch.cat`` works only when ALL candidates fall in one
        bucket, :132: same restriction, same error, on every rank), the top docs' (first global row, length) from their owners,
        and the rows themselves, every rank contributing the rows it holds."""
        loc = self.local
        dev = top_pids.device
        lo, hi = self.lo, self.hi
        mine = (all_cand >= lo) & (all_cand < hi)
        pad = loc.d_pad_len.to(dev)[(all_cand[mine] - lo)].to(torch.int64)
        big = 1 << 40
        mm = torch.tensor([-(int(pad.min().item()) if pad.numel() else big), int(pad.max().item()) if pad.numel() else -1],
                          dtype=torch.int64, device=dev)
        mm = self._reduce(mm, dist.ReduceOp.MAX)
        S_min, S = -int(mm[0].item()), int(mm[1].item())
        if S_min != S:
            raise RuntimeError("Sizes of tensors must match except in dimension 0 (candidates span several length buckets)")
        if self.tok_lo is None or self.tok_hi is None:
            raise ValueError("output_D_embedding needs tok_lo (the global token row of this shard's first token)")
        k = top_pids.numel()
        own = (top_pids >= lo) & (top_pids < hi)
        info = torch.zeros(k, 2, dtype=torch.int64, device=dev)               # (first global row, doclen), from the owner
        lp = top_pids[own] - lo
        info[own, 0] = self.tok_lo + loc.d_offsets.to(dev)[lp]
        info[own, 1] = loc.d_doclens.to(dev)[lp].to(torch.int64)
        info = self._reduce(info, dist.ReduceOp.SUM)
        ar = torch.arange(S, device=dev)
        rows = info[:, 0:1] + ar.unsqueeze(0)                                 # [k, S] global token rows
        held = (rows >= self.tok_lo) & (rows < self.tok_hi)
        D = torch.zeros(k, S, loc.tensor.size(-1), dtype=torch.float32, device=dev)
        D[held] = loc.tensor.to(dev)[(rows[held] - self.tok_lo)].to(torch.float32)     # :107
        D = self._reduce(D, dist.ReduceOp.SUM)                                # every row is held by exactly one rank (or none: zeros)
        mask = ar.unsqueeze(0) + 1 <= info[:, 1:2]                            # :108-109
        return D, mask


def load_shard(index_path, rank=None, world=None, device="cuda", index_dtype=torch.float16, group=None, fp32_mode="exact",
               dim=None, model=None, score_fn=None, topk_fn=None, pids_fn=None):
    """Rank ``rank``'s shard of a reference-built index, straight from its files (``{i}.pt`` + ``doclens.{i}.json``:
    loaders.py:7-32, index_manager.py:12-18, colbert_ranker.py:61-73) -> ``ShardedRanker``.

    Every rank reads ALL ``doclens.{i}.json`` (small) and derives, with no collective: its contiguous pid range
    (``shard_range``), the global token row of its first token, and the length-bucket strides of the WHOLE index
    (colbert_ranker.py:36-40; the 0-floor of :90, :108-109 depends on them).  Only the part files that overlap the
    shard's token range are opened, memory-mapped, and only the overlapping rows are copied to HBM -- a part that
    straddles a shard boundary is sliced.  ``rank`` / ``world`` default to the process group's; given explicitly they
    also work without one (one process loading shard after shard).  The constructor's stride cross-check (one
    all_reduce) still runs when a process group with more than one rank exists."""
    from .ranker import ColbertRanker, reference_strides
    from . import index_io
    if rank is None or world is None:
        if not (dist.is_available() and dist.is_initialized()):
            raise ValueError("load_shard: give rank and world, or initialise torch.distributed first")
        rank = dist.get_rank(group) if rank is None else rank
        world = dist.get_world_size(group) if world is None else world
    _, parts_paths, _ = index_io.get_parts(index_path)                        # colbert_ranker.py:18
    parts_doclens = index_io.load_doclens(index_path, flatten=False)          # :22
    doclens = [int(x) for y in parts_doclens for x in y]
    n_total = len(doclens)
    # every rank knows n_total and world: an index with fewer docs than ranks is refused by ALL ranks here, before any
    # collective (a rank that raised alone would leave the others waiting in the constructor's all_reduce)
    if n_total < world:
        raise ValueError(f"the index has {n_total} docs, fewer than the {world} ranks: at least one doc per shard is needed")
    lo, hi = shard_range(n_total, rank, world)
    strides = reference_strides(torch.tensor(doclens, dtype=torch.int64))    # of the WHOLE index
    tok_lo = sum(doclens[:lo])
    # walk the parts: docs [d0, d1) and token rows [t0, t1) of each; keep the overlap with [lo, hi)
    local_doclens, slices = [], []
    d0 = t0 = 0
    for path, dl in zip(parts_paths, parts_doclens):
        d1, t1 = d0 + len(dl), t0 + sum(int(x) for x in dl)
        a, b = max(lo, d0), min(hi, d1)
        if a < b:
            sub = [int(x) for x in dl[a - d0:b - d0]]
            r0 = sum(int(x) for x in dl[:a - d0])
            local_doclens.append(sub)
            slices.append((path, r0, r0 + sum(sub), t1 - t0))
        d0, t0 = d1, t1

    def rows():
        for path, r0, r1, n_rows in slices:
            part = index_io.load_index_part(path, mmap=True)
            assert part.size(0) == n_rows, (path, part.size(0), n_rows)       # colbert_ranker.py:69-70's size agreement
            yield part[r0:r1]
            del part
    local = ColbertRanker(parts=rows(), parts_doclens=local_doclens, dim=dim, model=model, device=device,
                          index_dtype=index_dtype, fp32_mode=fp32_mode, strides=strides)
    return ShardedRanker(local, lo, hi, group=group, score_fn=score_fn, topk_fn=topk_fn, pids_fn=pids_fn,
                         sync_strides=False, n_docs_total=n_total, tok_lo=tok_lo)


class _Exchange:
    def __init__(self, sr, top_p, top_s, depth):
        self.out = (top_p, top_s)
        self.done = None
        world = sr._world()
        if world == 1 and not (sr.force_exchange and dist.is_initialized()):
            return
        if top_s.device.type != "cuda":        # CPU ranks (gloo tests): synchronous
            gs, gp = all_gather_topk(top_s, top_p, world, sr.group)
            self.out = merge_gathered(gs, gp, depth, sr.topk_fn)
            return
        dev = top_s.device
        if getattr(sr, "_side", None) is None:
            sr._side = torch.cuda.Stream(device=dev)
        main = torch.cuda.current_stream(dev)
        ready = torch.cuda.Event()
        ready.record(main)
        self._inputs = (top_p, top_s)          # kept alive until result(): they were allocated on the caller's stream
        timing = getattr(sr, "exchange_events", None)     # bench.py: a list that receives (start, stop) event pairs
        with torch.cuda.stream(sr._side):
            sr._side.wait_event(ready)
            if timing is not None:
                t0 = torch.cuda.Event(enable_timing=True)
                t0.record(sr._side)
            gs, gp = all_gather_topk(top_s, top_p, world, sr.group)
            self.out = merge_gathered(gs, gp, depth, sr.topk_fn)
            self.done = torch.cuda.Event(enable_timing=timing is not None)
            self.done.record(sr._side)
            if timing is not None:
                timing.append((t0, self.done))

    def result(self):
        if self.done is not None:
            main = torch.cuda.current_stream(self.out[0].device)
            main.wait_event(self.done)
            for t in self.out:
                t.record_stream(main)          # allocated on the side stream, consumed on the caller's
            self.done = None
            self._inputs = None
        return self.out
