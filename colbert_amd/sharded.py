"""Doc-sharded multi-GPU rerank (no counterpart in the reference, whose rerank is single-GPU).

The index is partitioned by contiguous pid range, one shard per rank (one process per GPU).  Every rank scores
the candidates that fall in its range and takes a local top-k; ONE all_gather of ``[nq, k]`` scores + global pids
(RCCL over xGMI when the backend is ``nccl``; 12 B x nq x k per rank -- latency-bound) is followed by a per-query
``world*k -> k`` merge on every rank.  There is no other collective on this path.

``score_fn`` / ``topk_fn`` are injectable so the partition/gather/merge logic can be exercised on CPU ranks
(``gloo``) in tests with the oracle as scorer; the product default is the HIP path of the local ``ColbertRanker``.
"""
import torch
import torch.distributed as dist

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
    def __init__(self, local_ranker, lo, hi, group=None, score_fn=None, topk_fn=None):
        self.local = local_ranker
        self.lo, self.hi = int(lo), int(hi)
        self.group = group
        self.score_fn = score_fn if score_fn is not None else local_ranker.score_candidates
        self.topk_fn = topk_fn if topk_fn is not None else local_ranker.topk
        self.force_exchange = False   # diagnostic: run the exchange + merge even at world size 1 (bench.py --force-dist)

    def local_topk(self, Q, cand_global, depth, q_len=None, compact=True):
        cand_local, inr = localize(cand_global, self.lo, self.hi)
        gp = torch.where(inr, cand_global, torch.full_like(cand_global, -1))
        k = min(int(depth), cand_global.size(1))
        if compact and cand_global.size(1) > k:
            # a shard owns ~1/world of each list: move its candidates to the front and cut the width to the longest
            # local list (never below k, so the [nq, k] gather shape is the same on every rank) -- otherwise most
            # descriptor lanes of the rerank kernel would hold padding slots
            order = torch.argsort((~inr).to(torch.int8), dim=1, stable=True)
            width = max(int(inr.sum(1).max().item()), k)
            order = order[:, :width]
            cand_local = torch.gather(cand_local, 1, order)
            gp = torch.gather(gp, 1, order)
        scores = self.score_fn(Q, cand_local, q_len) if q_len is not None else self.score_fn(Q, cand_local)
        gp = gp.to(scores.device)
        return self.topk_fn(scores, gp, k)          # (pids [nq,k] global, scores [nq,k]); padding slots = (-1, -inf)

    def _world(self):
        if not (dist.is_available() and dist.is_initialized()):
            return 1
        return dist.get_world_size(self.group)

    def rerank_batch(self, Q, cand_global, depth=10, q_len=None):
        """Every rank passes the same Q [nq,Lq,h] and global candidate lists [nq,ncand]; returns the global
        top-``depth`` (pids, scores) on every rank."""
        top_p, top_s = self.local_topk(Q, cand_global, depth, q_len)
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
        with torch.cuda.stream(sr._side):
            sr._side.wait_event(ready)
            gs, gp = all_gather_topk(top_s, top_p, world, sr.group)
            self.out = merge_gathered(gs, gp, depth, sr.topk_fn)
            self.done = torch.cuda.Event()
            self.done.record(sr._side)

    def result(self):
        if self.done is not None:
            main = torch.cuda.current_stream(self.out[0].device)
            main.wait_event(self.done)
            for t in self.out:
                t.record_stream(main)          # allocated on the side stream, consumed on the caller's
            self.done = None
            self._inputs = None
        return self.out
