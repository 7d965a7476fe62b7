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


def same_ranking(got_p, got_s, exp_scores, cand, k, atol=1e-5):
    """Top-k equality modulo the order of tied scores (the reference's sort is unstable, colbert_ranker.py:128): the score
    lists are equal, and every returned pid carries the score the whole-index oracle gives THAT pid."""
    _, es = cpu_topk(exp_scores, cand, k)
    if not torch.allclose(got_s, es, atol=atol, rtol=0):
        return False
    for qi in range(cand.size(0)):
        if len(set(got_p[qi].tolist())) != got_p.size(1):
            return False
        for p, sc in zip(got_p[qi].tolist(), got_s[qi].tolist()):
            pos = (cand[qi] == p).nonzero()
            if len(pos) == 0 or abs(float(exp_scores[qi, pos[0, 0]]) - sc) > atol:
                return False
    return True


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


def build_world(seed=0, ndocs=48, h=16):
    """A ragged index whose two halves have DIFFERENT length distributions (short docs first, long docs last), so the
    percentile strides of a shard differ from those of the whole index, and token embeddings that all lean towards
    +e0, so the query -e0 has a negative similarity with every token: its score is decided by the reference's
    zero-padding floor alone (colbert_ranker.py:90,108-109) -- i.e. by which strides the docs were bucketed with."""
    gen = torch.Generator().manual_seed(seed)
    half = ndocs // 2
    doclens = torch.cat([torch.randint(1, 9, (half,), generator=gen), torch.randint(6, 20, (ndocs - half,), generator=gen)]).tolist()
    emb = torch.randn(sum(doclens), h, generator=gen) * 0.3
    emb[:, 0] += 1.0
    emb = F.normalize(emb, dim=-1).half()
    Q = F.normalize(torch.randn(3, 6, h, generator=gen), dim=-1)
    Q[2] = 0.0
    Q[2, :, 0] = -1.0                      # the "negative" query
    cand = torch.stack([torch.randperm(ndocs, generator=gen)[:29] for _ in range(3)])
    return doclens, emb, Q, cand


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from colbert_amd.sharded import ShardedRanker, global_strides, shard_range
        doclens, emb, Q, cand = build_world()
        ndocs = len(doclens)
        lo, hi = shard_range(ndocs, rank, world)
        offs = [0]
        for d in doclens:
            offs.append(offs[-1] + d)
        # expected: ONE reference ranker over the WHOLE index (what an unsharded deployment returns)
        whole = RefRanker([emb], [doclens], dim=emb.size(1))
        exp = torch.stack([whole.all_scores(Q[qi].unsqueeze(0).permute(0, 2, 1), cand[qi].tolist()) for qi in range(Q.size(0))])
        K = cand.size(1)                    # the whole list: every candidate's score is checked, not only the best five
        # the shard: an oracle ranker over the local docs, bucketed by the GLOBAL strides
        local = RefRanker([emb[offs[lo]:offs[hi]]], [doclens[lo:hi]], dim=emb.size(1))
        own_strides = list(local.strides)
        gs = global_strides(doclens[lo:hi])
        checks = {'global_strides': gs == whole.strides, 'not_vacuous': own_strides != whole.strides}
        # the SHIPPED constructor under a real process group: ShardedRanker.__init__ derives the strides of the whole
        # index (two all_reduces, sharded.py) and re-buckets the shard's ColbertRanker through set_strides; the oracle
        # scorer then buckets by whatever strides that left behind
        import colbert_amd
        cr = colbert_amd.ColbertRanker(parts=[emb[offs[lo]:offs[hi]]], parts_doclens=[doclens[lo:hi]], dim=emb.size(1), device="cpu")
        checks['ctor_own_strides_first'] = cr.strides == own_strides
        sh = ShardedRanker(cr, lo, hi, score_fn=make_scorer(local), topk_fn=cpu_topk)
        checks['ctor_global_strides'] = cr.strides == whole.strides
        checks['ctor_pad_len'] = torch.equal(cr.d_pad_len.long(), whole.bucket_strides(list(range(lo, hi))))
        local.strides = list(cr.strides)
        local.views = local._create_views(local.tensor)
        # a rank that keeps its own percentiles (sync_strides=False) is an error, not a silently different score
        cr_own = colbert_amd.ColbertRanker(parts=[emb[offs[lo]:offs[hi]]], parts_doclens=[doclens[lo:hi]], dim=emb.size(1), device="cpu")
        try:
            ShardedRanker(cr_own, lo, hi, score_fn=make_scorer(local), topk_fn=cpu_topk, sync_strides=False)
            checks['disagreeing_strides_raise'] = False
        except RuntimeError as e:
            checks['disagreeing_strides_raise'] = "disagree" in str(e)
        top_p, top_s = sh.rerank_batch(Q, cand, depth=K)
        checks['sharded_eq_whole'] = same_ranking(top_p, top_s, exp, cand, K)
        top_p, top_s = sh.rerank_batch(Q, cand, depth=5)
        checks['sharded_eq_whole_top5'] = same_ranking(top_p, top_s, exp, cand, 5)
        # ... and bucketing each shard by its OWN percentiles (round 1's behaviour) gives different scores here
        local_own = RefRanker([emb[offs[lo]:offs[hi]]], [doclens[lo:hi]], dim=emb.size(1))
        sh_own = ShardedRanker(object(), lo, hi, score_fn=make_scorer(local_own), topk_fn=cpu_topk)
        own_p, own_s = sh_own.rerank_batch(Q, cand, depth=K)
        checks['own_strides_differ'] = not same_ranking(own_p, own_s, exp, cand, K, atol=1e-3)
        # q_mask / q_len travel through the sharded path
        qm = torch.ones(Q.shape[:2], dtype=torch.long)
        qm[:, 1] = 0
        qm[0, 4] = 0
        def masked_scorer(Qb, cl, q_mask=None, q_len=None):
            out = torch.full(cl.shape, NEG_INF)
            for qi in range(Qb.size(0)):
                okc = (cl[qi] >= 0).nonzero().flatten()
                if len(okc):
                    q = Qb[qi][q_mask[qi].bool()]
                    out[qi, okc] = local.all_scores(q.unsqueeze(0).permute(0, 2, 1), cl[qi, okc].tolist())
            return out
        shm = ShardedRanker(object(), lo, hi, score_fn=masked_scorer, topk_fn=cpu_topk)
        mp_, ms_ = shm.rerank_batch(Q, cand, depth=K, q_mask=qm)
        expm = torch.stack([whole.all_scores(Q[qi][qm[qi].bool()].unsqueeze(0).permute(0, 2, 1), cand[qi].tolist())
                            for qi in range(Q.size(0))])
        checks['q_mask'] = same_ranking(mp_, ms_, expm, cand, K)
        # the pipelined form bench.py uses: local top-k, then the exchange as a handle (synchronous on CPU ranks);
        # two batches in flight, resolved in issue order
        h1 = sh.exchange_async(*sh.local_topk(Q, cand, 5), 5)
        h2 = sh.exchange_async(*sh.local_topk(Q, cand.flip(1), 5), 5)
        p1, s1 = h1.result()
        p2, s2 = h2.result()
        checks['pipelined'] = same_ranking(p1, s1, exp, cand, 5) and same_ranking(p2, s2, exp, cand, 5)
        ret[rank] = sorted(k for k, v in checks.items() if not v)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_doc_sharded_rerank_world2():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    assert dict(ret) == {0: [], 1: []}          # per rank: the names of the failed checks


def test_strides_from_histogram_equal_kthvalue():
    from colbert_amd.ranker import reference_strides, strides_from_histogram
    g = torch.Generator().manual_seed(5)
    for n, hi in [(4, 3), (5, 9), (97, 180), (1000, 12), (1001, 300), (64, 1)]:
        dl = torch.randint(0 if hi > 1 else 1, hi + 1, (n,), generator=g)
        if int(dl.max()) == 0:
            dl[0] = 1
        assert strides_from_histogram(torch.bincount(dl)) == reference_strides(dl), (n, hi)
    with pytest.raises(RuntimeError):
        strides_from_histogram(torch.bincount(torch.tensor([1, 2, 3])))      # N < 4: kthvalue(0), as the reference


def test_shard_candidates_cpu_partition_is_stable():
    from colbert_amd.sharded import shard_candidates
    c = torch.tensor([[0, 5, 9, 3, 4], [4, 4, 2, 8, 5]])
    loc, gp = shard_candidates(c, 3, 6)
    assert loc.tolist() == [[2, 0, 1, -1, -1], [1, 1, 2, -1, -1]]
    assert gp.tolist() == [[5, 3, 4, -1, -1], [4, 4, 5, -1, -1]]


def test_shard_candidates_cpu_counts_and_single_process_checks():
    """The per-row live counts the counted rerank is scheduled from (CPU form of maxsim_shard_candidates' out_count), and
    the stride agreement check as a no-op without a process group."""
    from colbert_amd.sharded import assert_strides_agree, shard_candidates
    c = torch.tensor([[0, 5, 9, 3, 4], [4, 4, 2, 8, 5], [9, 9, 9, 9, 9]])
    loc, gp, cnt = shard_candidates(c, 3, 6, with_counts=True)
    assert cnt.dtype == torch.int32 and cnt.tolist() == [3, 3, 0]
    assert loc.tolist() == [[2, 0, 1, -1, -1], [1, 1, 2, -1, -1], [-1] * 5]
    for q in range(3):                                   # the counted-rows precondition: live first, negative behind
        assert bool((loc[q, :cnt[q]] >= 0).all()) and bool((loc[q, cnt[q]:] < 0).all())
    assert_strides_agree([8, 20, 180])                   # no process group: nothing to compare (world 2: the gloo test above)


def test_shard_range_and_localize():
    from colbert_amd.sharded import localize, merge_gathered, shard_range
    assert [shard_range(10, r, 4) for r in range(4)] == [(0, 2), (2, 5), (5, 7), (7, 10)]      # balanced: sizes differ by <= 1
    assert [shard_range(12, r, 8) for r in range(8)] == [(0, 1), (1, 3), (3, 4), (4, 6), (6, 7), (7, 9), (9, 10), (10, 12)]   # none empty
    for n, w in ((1, 1), (7, 3), (64, 5), (1000, 8), (8, 8)):
        rs = [shard_range(n, r, w) for r in range(w)]
        assert rs[0][0] == 0 and rs[-1][1] == n and all(a[1] == b[0] for a, b in zip(rs, rs[1:])) and all(hi > lo for lo, hi in rs)
    assert [shard_range(8, r, 8) for r in range(8)] == [(i, i + 1) for i in range(8)]
    c = torch.tensor([[0, 5, 9, 3], [4, 4, 2, 8]])
    loc, inr = localize(c, 3, 6)
    assert loc.tolist() == [[-1, 2, -1, 0], [1, 1, -1, -1]] and inr.tolist() == [[False, True, False, True], [True, True, False, False]]
    gs = torch.tensor([[[3.0, 1.0]], [[2.0, NEG_INF]]])        # [world=2, nq=1, k=2]
    gp = torch.tensor([[[7, 5]], [[9, -1]]])
    p, s = merge_gathered(gs, gp, 3, cpu_topk)
    assert s.tolist() == [[3.0, 2.0, 1.0]] and p.tolist() == [[7, 9, 5]]
