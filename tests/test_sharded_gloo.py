"""CPU, world_size 2, gloo: the doc-sharded path (partition -> local top-k -> all_gather -> merge) with the
oracle injected as the scorer.  This exercises colbert_amd/sharded.py's host logic exactly as the N>1 bench runs
it, minus the HIP kernel (covered by the gpu tests)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn.functional as F

from oracle.maxsim_oracle import RefRanker

NEG_INF = float("-inf")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def cpu_topk(scores, pids, k):
    es, ei = torch.sort(scores, dim=1, descending=True, stable=True)
    p = ei if pids is None else torch.gather(pids, 1, ei)
    return p[:, :k].contiguous(), es[:, :k].contiguous()


def make_scorer(ref):
    def scorer(Q, cand_local, q_len=None):
        out = torch.full(cand_local.shape, NEG_INF)
        for qi in range(Q.size(0)):
            ok = (cand_local[qi] >= 0).nonzero().flatten()
            if len(ok):
                q = Q[qi] if q_len is None else Q[qi, : int(q_len[qi])]
                out[qi, ok] = ref.all_scores(q.unsqueeze(0).permute(0, 2, 1), cand_local[qi, ok].tolist())
        return out
    return scorer


def build_world(seed=0, ndocs=40, h=16):
    gen = torch.Generator().manual_seed(seed)
    doclens = torch.randint(1, 20, (ndocs,), generator=gen).tolist()
    emb = F.normalize(torch.randn(sum(doclens), h, generator=gen), dim=-1).half()
    Q = F.normalize(torch.randn(3, 6, h, generator=gen), dim=-1)
    cand = torch.stack([torch.randperm(ndocs, generator=gen)[:17] for _ in range(3)])
    return doclens, emb, Q, cand


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from colbert_amd.sharded import ShardedRanker, shard_range
        doclens, emb, Q, cand = build_world()
        ndocs = len(doclens)
        lo, hi = shard_range(ndocs, rank, world)
        offs = [0]
        for d in doclens:
            offs.append(offs[-1] + d)
        # NOTE: each shard computes its own length-bucket strides (as a reference-built per-shard index would)
        local = RefRanker([emb[offs[lo]:offs[hi]]], [doclens[lo:hi]], dim=emb.size(1))
        sh = ShardedRanker(object(), lo, hi, score_fn=make_scorer(local), topk_fn=cpu_topk)
        top_p, top_s = sh.rerank_batch(Q, cand, depth=5)
        # expected: per query, union of per-shard scores
        exp = torch.full(cand.shape, NEG_INF)
        for r in range(world):
            l2, h2 = shard_range(ndocs, r, world)
            ref_r = RefRanker([emb[offs[l2]:offs[h2]]], [doclens[l2:h2]], dim=emb.size(1))
            sc = make_scorer(ref_r)(Q, torch.where((cand >= l2) & (cand < h2), cand - l2, torch.full_like(cand, -1)))
            exp = torch.maximum(exp, sc)
        ep, es = cpu_topk(exp, cand, 5)
        ok = torch.equal(top_s, es) and torch.equal(top_p, ep)
        # the pipelined form bench.py uses: local top-k, then the exchange as a handle (synchronous on CPU ranks);
        # two batches in flight, resolved in issue order
        h1 = sh.exchange_async(*sh.local_topk(Q, cand, 5), 5)
        h2 = sh.exchange_async(*sh.local_topk(Q, cand.flip(1), 5), 5)
        p1, s1 = h1.result()
        p2, s2 = h2.result()
        ok = ok and torch.equal(s1, es) and torch.equal(p1, ep) and torch.equal(s2, es)
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_doc_sharded_rerank_world2():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    assert dict(ret) == {0: True, 1: True}


def test_shard_range_and_localize():
    from colbert_amd.sharded import localize, merge_gathered, shard_range
    assert [shard_range(10, r, 4) for r in range(4)] == [(0, 3), (3, 6), (6, 9), (9, 10)]
    assert [shard_range(8, r, 8) for r in range(8)] == [(i, i + 1) for i in range(8)]
    c = torch.tensor([[0, 5, 9, 3], [4, 4, 2, 8]])
    loc, inr = localize(c, 3, 6)
    assert loc.tolist() == [[-1, 2, -1, 0], [1, 1, -1, -1]] and inr.tolist() == [[False, True, False, True], [True, True, False, False]]
    gs = torch.tensor([[[3.0, 1.0]], [[2.0, NEG_INF]]])        # [world=2, nq=1, k=2]
    gp = torch.tensor([[[7, 5]], [[9, -1]]])
    p, s = merge_gathered(gs, gp, 3, cpu_topk)
    assert s.tolist() == [[3.0, 2.0, 1.0]] and p.tolist() == [[7, 9, 5]]
