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
    """Contiguous pid range [lo, hi) of a rank (same structure as the reference's per-part files, loaders.py:7-19)."""
    per = (n_docs_total + world - 1) // world
    lo = min(rank * per, n_docs_total)
    return lo, min(lo + per, n_docs_total)


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
    all_reduce(MAX) of (strides, -strides)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    st = sorted(int(x) for x in strides)
    if len(st) > 4:
        raise ValueError(f"at most 4 distinct strides (colbert_ranker.py:36-40), got {st}")
    v = torch.tensor([len(st)] + st + [0] * (4 - len(st)), dtype=torch.int64)
    both = torch.cat([v, -v])
    if device is None and dist.get_backend(group) == "nccl":
        device = torch.device("cuda", torch.cuda.current_device())
    both = both.to(device) if device is not None else both
    dist.all_reduce(both, op=dist.ReduceOp.MAX, group=group)
    both = both.cpu()
    if not torch.equal(both[:5], -both[5:]):
        raise RuntimeError(f"doc shards disagree on the length-bucket strides: this rank has {st}, the ranks' maxima are "
                           f"{both[1:5].tolist()} and minima {(-both[6:]).tolist()} -- every shard must use the strides "
                           f"of the WHOLE index (sharded.global_strides)")


def all_gather_topk(top_s, top_p, world, group=None):
    """The path's ONE exchange step: every rank's [nq, k] scores and global pids -> [world, nq, k] on every rank.
    Scores travel as their bit patterns next to the pids in one int64 payload, so it is a single collective."""
    nq, k = top_s.shape
    dev = top_s.device
    payload = torch.empty(nq, 2 * k, dtype=torch.int64, device=dev)
    payload[:, :k] = top_p
    payload[:, k:] = top_s.contiguous().view(torch.int32)
    if dev.type == "cuda" and dist.get_backend(group) == "gloo":
        # rehearsal on a one-GPU box (RCCL refuses two ranks on one device): stage through the host
        payload = payload.cpu()
    g = torch.empty(world * nq, 2 * k, dtype=torch.int64, device=payload.device)
    dist.all_gather_into_tensor(g, payload, group=group)
    g = g.view(world, nq, 2 * k).to(dev)
    gs = g[..., k:].to(torch.int32).view(torch.float32)
    return gs, g[..., :k]


def merge_gathered(all_scores, all_pids, k, topk_fn):
    """[world, nq, k] gathered local top-k -> global top-k per query."""
    world, nq, kk = all_scores.shape
    s = all_scores.permute(1, 0, 2).reshape(nq, world * kk).contiguous()
    p = all_pids.permute(1, 0, 2).reshape(nq, world * kk).contiguous()
    return topk_fn(s, p, min(k, world * kk))


class ShardedRanker:
    def __init__(self, local_ranker, lo, hi, group=None, score_fn=None, topk_fn=None, sync_strides=True):
        self.local = local_ranker
        self.lo, self.hi = int(lo), int(hi)
        self.group = group
        self.score_fn = score_fn if score_fn is not None else local_ranker.score_candidates
        self.topk_fn = topk_fn if topk_fn is not None else local_ranker.topk
        # the product scorer / top-k take the per-row live counts (counted rows: maxsim_rerank_counted / maxsim_topk_counted);
        # injected ones (CPU tests) get the plain signature unless they say otherwise
        self.score_counted = score_fn is None
        self.topk_counted = topk_fn is None
        self.force_exchange = False   # diagnostic: run the exchange + merge even at world size 1 (bench.py --force-dist)
        self.exchange_events = None   # diagnostic: a list here collects (start, stop) HIP events of every exchange + merge
        # bucket by the strides of the whole index (see the module docstring); every rank must construct its
        # ShardedRanker at the same point (two small all_reduces)
        if hasattr(local_ranker, "set_strides") and self._world() > 1:
            dev = local_ranker.device if local_ranker.device.type == "cuda" else None
            if sync_strides:
                local_ranker.set_strides(global_strides(local_ranker.doclens, group, dev))
            # sync_strides=False (the caller set the strides itself) is still checked: ranks that bucket differently
            # would return scores that differ from the unsharded reference without any error
            assert_strides_agree(local_ranker.strides, group, dev)

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
