"""The doc-sharded path from the reference's index FILES to python lists:
``sharded.load_shard`` ({i}.pt + doclens.{i}.json -> one rank's pid range, global strides derived locally),
``ShardedRanker.rank_forward`` and ``ShardedRanker.retrieve_batch``.

CPU (gloo, world 2): both ranks are built from a directory written by ``save_index`` out of the golden fixtures
(generated with the imported reference, tests/golden/make_golden.py) and must return the goldens' results; the scorer,
top-k and ids->pids steps are the oracle / torch (injected), everything else is the shipped host logic.
GPU: N = 2, 4, 8 shards loaded one after another from the same files on the HIP scorer, merged with
``merge_gathered``: equal to the unsharded HIP result bit for bit."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle.maxsim_oracle import RefRanker

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
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


def write_index(tmp, name):
    from colbert_amd.index_io import save_index
    z = np.load(os.path.join(GOLD, name))
    parts = [torch.from_numpy(z["part0"]), torch.from_numpy(z["part1"])]
    dls = [z["doclens0"].tolist(), z["doclens1"].tolist()]
    save_index(tmp, parts, dls)
    return z, parts, dls


def oracle_fns(sh, strides):
    """Scorer / ids->pids for a CPU shard: the oracle over the shard's OWN rows bucketed by the strides the shipped
    loader derived (``sh.local.strides``), and the reference's emb2pid + set() (colbert_ranker.py:163-174, :212-229)."""
    loc = sh.local
    ref = RefRanker([loc.tensor], [loc.doclens.tolist()], dim=loc.dim, strides=strides)   # (a 1-doc shard has no percentiles)
    emb2pid = torch.repeat_interleave(torch.arange(loc.n_docs), loc.doclens)

    def scorer(Q, cand_local, q_mask=None, q_len=None):
        out = torch.full(cand_local.shape, NEG_INF)
        for qi in range(Q.size(0)):
            ok = (cand_local[qi] >= 0).nonzero().flatten()
            if len(ok):
                q = Q[qi] if q_mask is None else Q[qi][q_mask[qi].bool()]
                out[qi, ok] = ref.all_scores(q.unsqueeze(0).permute(0, 2, 1), cand_local[qi, ok].tolist())
        return out

    def pids_fn(local_ids):
        bs, n = local_ids.shape
        out = torch.full((bs, n), -1, dtype=torch.int64)
        cnt = torch.zeros(bs, dtype=torch.int32)
        for qi in range(bs):
            live = local_ids[qi][local_ids[qi] >= 0]
            u = sorted(set(emb2pid[live].tolist()))
            out[qi, :len(u)] = torch.tensor(u, dtype=torch.int64)
            cnt[qi] = len(u)
        return out, cnt
    return scorer, pids_fn


def load_cpu_shard(path, rank, world):
    from colbert_amd.sharded import load_shard
    sh = load_shard(path, rank, world, device="cpu", score_fn=lambda *a, **k: None, topk_fn=cpu_topk, pids_fn=lambda x: None)
    sh.score_fn, sh.pids_fn = oracle_fns(sh, sh.local.strides)
    return sh


def cover_ids(doclens, Lq, depth, nq, seed):
    """ANN-like token rows [nq, Lq, depth] that touch EVERY doc at least once (so the distinct pids are all docs), with
    duplicates and -1 padding, in global token rows."""
    g = torch.Generator().manual_seed(seed)
    offs = np.concatenate([[0], np.cumsum(doclens)])
    ids = torch.full((nq, Lq * depth), -1, dtype=torch.int64)
    for q in range(nq):
        rows = [int(offs[d] + torch.randint(0, int(doclens[d]), (1,), generator=g)) for d in range(len(doclens))]
        extra = torch.randint(0, int(offs[-1]), (Lq * depth - len(rows) - 7,), generator=g).tolist()
        row = torch.tensor(rows + extra + [-1] * 7)
        ids[q] = row[torch.randperm(row.numel(), generator=g)]
    return ids.view(nq, Lq, depth)


def _worker(rank, world, port, tmp_r, tmp_m, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        checks = {}
        # --- ragged_rerank_64: rank_forward from files
        z = np.load(os.path.join(GOLD, "ragged_rerank_64.npz"))
        from colbert_amd.sharded import shard_range
        sh = load_cpu_shard(tmp_r, None if rank else 0, None if rank else world)   # ranks > 0 take rank/world from the group
        checks["range"] = (sh.lo, sh.hi) == shard_range(64, rank, world)
        checks["global_strides"] = sh.local.strides == z["strides"].tolist()
        checks["pad_len"] = sh.local.d_pad_len.tolist() == z["pad_len"][sh.lo:sh.hi].tolist()
        dl_all = z["doclens0"].tolist() + z["doclens1"].tolist()
        checks["tok_lo"] = sh.tok_lo == sum(dl_all[:sh.lo]) and sh.tok_hi == sum(dl_all[:sh.hi])
        checks["n_docs_total"] = sh.n_docs_total == 64
        Q = torch.from_numpy(z["Q"])
        pids = z["pids"].tolist()
        got_p, got_s = sh.rank_forward(Q, pids, depth=10)
        checks["top10_pids"] = got_p == z["top10_pids"].tolist()
        checks["top10_scores"] = np.allclose(np.array(got_s), z["top10_scores"], rtol=0, atol=1e-5)
        # the whole score vector, through the tensor form of pids and the full depth
        got_p, got_s = sh.rank_forward(Q, torch.tensor(pids), depth=64)
        by_pid = dict(zip(got_p, got_s))
        checks["all_scores"] = len(by_pid) == 64 and np.allclose([by_pid[p] for p in pids], z["expected_scores"], rtol=0, atol=1e-5)
        # the query whose score is decided by the 0-floor alone (all similarities negative)
        got_p, got_s = sh.rank_forward(torch.from_numpy(z["Q_neg"]), pids, depth=64)
        by_pid = dict(zip(got_p, got_s))
        checks["zero_floor"] = np.allclose([by_pid[p] for p in pids], z["expected_scores_neg"], rtol=0, atol=1e-5)
        # reference contract: negative pids wrap and come back as passed in; out-of-range raises IndexError (:88)
        neg = [p - 64 if i % 3 == 0 else p for i, p in enumerate(pids)]
        got_p, got_s = sh.rank_forward(Q, neg, depth=10)
        checks["negative_wrap"] = ([p % 64 for p in got_p] == z["top10_pids"].tolist() and set(got_p) <= set(neg)
                                   and np.allclose(np.array(got_s), z["top10_scores"], rtol=0, atol=1e-5))
        try:
            sh.rank_forward(Q, pids[:5] + [64], depth=10)
            checks["index_error"] = False
        except IndexError:
            checks["index_error"] = True
        try:
            sh.rank_forward(Q, [], depth=10)
            checks["empty_asserts"] = False
        except AssertionError:
            checks["empty_asserts"] = True
        # a list that lives on ONE shard only (the last rank's docs): every other rank's share of it is empty -- its local
        # top-k is all (-1, -inf) padding, which must cross all_gather_topk + merge_gathered without winning a slot; and
        # depth (10) larger than what any rank holds of a 4-pid list spread over the shards
        score_of = dict(zip(pids, z["expected_scores"].tolist()))
        last_lo, last_hi = shard_range(64, world - 1, world)
        own = [p for p in pids if last_lo <= p < last_hi]
        got_p, got_s = sh.rank_forward(Q, own, depth=10)
        want = sorted(own, key=lambda p: -score_of[p])[:10]
        checks["one_shard_list"] = (len(got_p) == min(10, len(own)) and np.allclose(got_s, [score_of[p] for p in want], rtol=0, atol=1e-5)
                                    and all(abs(score_of[p] - sc) <= 1e-5 for p, sc in zip(got_p, got_s)))
        few = [pids[0], pids[21], pids[42], pids[63]]
        got_p, got_s = sh.rank_forward(Q, few, depth=10)
        want = sorted(few, key=lambda p: -score_of[p])
        checks["k_above_live_count"] = got_p == want and np.allclose(got_s, [score_of[p] for p in want], rtol=0, atol=1e-5)
        # rerank_batch on a 2-query batch whose second row is all padding on every rank but one
        from colbert_amd.sharded import all_gather_topk, merge_gathered
        Qt = Q.permute(0, 2, 1).contiguous()
        cand2 = torch.full((2, 64), -1, dtype=torch.int64)
        cand2[0] = torch.tensor(pids)
        cand2[1, :len(own)] = torch.tensor(own)
        top_p, top_s = sh.rerank_batch(torch.cat([Qt, Qt]), cand2, depth=12)
        n1 = min(12, len(own))
        checks["batch_with_padding_rows"] = (top_p[0, :10].tolist() == z["top10_pids"].tolist()
                                             and sorted(top_p[1, :n1].tolist()) == sorted(sorted(own, key=lambda p: -score_of[p])[:n1])
                                             and bool((top_p[1, n1:] == -1).all()) and bool(torch.isinf(top_s[1, n1:]).all()))

        # output_D_embedding (colbert_ranker.py:131-136) across shards: the top docs' padded rows as the reference's strided
        # view hands them over -- incl. slots past a doc's end that belong to the next doc / the next SHARD / the zero tail --
        # against the restated ranker over the whole index; candidates of ONE length bucket (the reference's torch.cat, :132)
        whole_ref = RefRanker([torch.from_numpy(z["part0"]), torch.from_numpy(z["part1"])],
                              [z["doclens0"].tolist(), z["doclens1"].tolist()], dim=128)
        pad_all = z["pad_len"].tolist()
        for bucket in sorted(set(pad_all)):
            one = [p for p in pids if pad_all[p] == bucket]                # (pids holds all 64 docs, the index's last one too: its
                                                                           #  padding slots are the zero tail behind the index)
            got_p, got_D, got_m = sh.rank_forward(Q, one, depth=7, output_D_embedding=True)
            exp_p, exp_D, exp_m = whole_ref.rank_forward(Q, one, depth=7, output_D_embedding=True)
            checks[f"output_D_{bucket}"] = (got_p == exp_p and torch.equal(got_D, exp_D.float()) and torch.equal(got_m, exp_m)
                                            and got_D.dtype == torch.float32 and tuple(got_D.shape) == (min(7, len(one)), bucket, 128))
        try:
            sh.rank_forward(Q, pids, depth=5, output_D_embedding=True)     # all 64 docs: several buckets
            checks["output_D_mixed_buckets_raise"] = False
        except RuntimeError as e:
            checks["output_D_mixed_buckets_raise"] = "Sizes of tensors must match" in str(e)

        # --- masked_query_rerank: the batched driver from files, global token rows in
        m = np.load(os.path.join(GOLD, "masked_query_rerank.npz"))
        shm = load_cpu_shard(tmp_m, rank, world)
        dl = m["doclens0"].tolist() + m["doclens1"].tolist()
        checks["m_range"] = (shm.lo, shm.hi) == shard_range(12, rank, world)          # part0 has 5 docs: parts are sliced
        Qm, keep = torch.from_numpy(m["Q"]), torch.from_numpy(m["q_word_mask"])
        ids = cover_ids(dl, Qm.size(1), 4, Qm.size(0), seed=11)
        from colbert_amd import retrieve_batch
        out = retrieve_batch(shm, Qm, keep, topk=12, embedding_ids=ids)
        ok = len(out) == Qm.size(0)
        m_pids = m["pids"].tolist()                 # expected_scores[q, i] is the score of pid m_pids[i]
        exp_by_pid = torch.from_numpy(m["expected_scores"])[:, torch.argsort(torch.tensor(m_pids))]
        emb2pid = torch.repeat_interleave(torch.arange(12), torch.tensor(dl))
        n_hit = []
        for qi, (p, s) in enumerate(out):
            exp = exp_by_pid[qi]
            live = ids[qi][keep[qi].bool()].flatten()               # rows of dropped tokens do not count (keep_nonzero)
            hit = sorted(set(emb2pid[live[live >= 0]].tolist()))     # colbert_ranker.py:212-229
            n_hit.append(len(hit))
            order = torch.sort(exp[hit], descending=True, stable=True)
            ok = ok and sorted(p) == hit
            ok = ok and np.allclose(np.array(s), order.values.numpy(), rtol=0, atol=1e-5)
            ok = ok and all(abs(float(exp[pid]) - sc) <= 1e-5 for pid, sc in zip(p, s))
        ok = ok and n_hit[0] == 12 and n_hit[2] < 12                 # all docs for query 0; the one-live-token query sees few
        checks["retrieve_batch"] = ok
        # fewer distinct candidates than topk: only the docs that were hit come back, rows of dropped tokens are ignored
        ids2 = torch.full_like(ids, -1)
        offs = np.concatenate([[0], np.cumsum(dl)])
        live_tok = int(keep[0].nonzero()[0])
        dead_tok = int((keep[0] == 0).nonzero()[0])
        ids2[0, live_tok, :3] = torch.tensor([int(offs[2]), int(offs[9]) + 1, int(offs[2]) + 1])
        ids2[0, dead_tok, :2] = torch.tensor([int(offs[4]), int(offs[11])])       # a dropped token's neighbours
        out2 = shm.retrieve_batch(Qm, keep, 12, embedding_ids=ids2)
        exp0 = exp_by_pid[0].numpy()
        want = sorted([2, 9], key=lambda d: -exp0[d])
        checks["short_rows"] = (out2[0][0] == want and np.allclose(out2[0][1], [exp0[d] for d in want], atol=1e-5)
                                and out2[1] == ([], []) and out2[2] == ([], []))
        # ann_search callable form: called once with the live tokens only
        calls = []

        def ann(q_live, depth):
            calls.append(tuple(q_live.shape))
            return ids[keep.bool()]
        out3 = shm.retrieve_batch(Qm, keep, 12, ann_search=ann, faiss_depth=4)
        checks["ann_search"] = calls == [(int(keep.sum()), Qm.size(2))] and [o[0] for o in out3] == [o[0] for o in out]
        ret[rank] = sorted(k for k, v in checks.items() if not v)
    finally:
        dist.destroy_process_group()


def _spawn(world, tmp_path):
    tmp_r, tmp_m = str(tmp_path / "ragged"), str(tmp_path / "masked")
    write_index(tmp_r, "ragged_rerank_64.npz")
    write_index(tmp_m, "masked_query_rerank.npz")
    port = _free_port()
    ret = mp.Manager().dict()
    mp.spawn(_worker, args=(world, port, tmp_r, tmp_m, ret), nprocs=world, join=True)
    assert dict(ret) == {r: [] for r in range(world)}          # per rank: the names of the failed checks


@pytest.mark.timeout(300)
def test_sharded_from_files_world2(tmp_path):
    _spawn(2, tmp_path)


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world", [3, 8])
def test_sharded_from_files_uneven_worlds(tmp_path, world):
    """The same checks THROUGH THE COLLECTIVES at world 3 and 8 (gloo): shards of unequal size (64 docs: 21 / 21 / 22; 12
    docs on 8 ranks: one or two each), ranks whose share of a list is empty, depth above a rank's live count, negative pids,
    the batched driver on global token rows -- every rank must return the goldens' results."""
    _spawn(world, tmp_path)


def _tiny_worker(rank, world, port, tmp, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import datetime
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60))
    try:
        try:
            load_cpu_shard(tmp, rank, world)
            ret[rank] = "no error"
        except ValueError as e:
            ret[rank] = str(e)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_index_smaller_than_world_is_refused_by_every_rank(tmp_path):
    """A 1-doc index on 2 ranks: BOTH ranks raise before any collective (a rank raising alone would leave the other in the
    constructor's all_reduce until the process-group timeout)."""
    from colbert_amd.index_io import save_index
    g = torch.Generator().manual_seed(0)
    save_index(str(tmp_path), [torch.nn.functional.normalize(torch.randn(5, 128, generator=g), dim=-1).half()], [[5]])
    ret = mp.Manager().dict()
    mp.spawn(_tiny_worker, args=(2, _free_port(), str(tmp_path), ret), nprocs=2, join=True)
    assert all("fewer than the 2 ranks" in ret[r] for r in range(2)), dict(ret)


@pytest.mark.parametrize("world", [1, 2, 3, 5, 8])
def test_load_shard_slices_parts(tmp_path, world):
    """Shards cut the part files at arbitrary docs: the concatenation of all shards' rows is the index, every shard
    carries the strides of the whole index, and tok_lo / lo line up with the doclens prefix sums.  No process group."""
    from colbert_amd.ranker import reference_strides
    from colbert_amd.sharded import load_shard, shard_range
    z, parts, dls = write_index(str(tmp_path), "ragged_rerank_64.npz")
    whole = torch.cat(parts)
    dl = dls[0] + dls[1]
    offs = np.concatenate([[0], np.cumsum(dl)])
    rows, docs = [], []
    for r in range(world):
        sh = load_shard(str(tmp_path), r, world, device="cpu", score_fn=lambda *a: None, topk_fn=cpu_topk, pids_fn=lambda x: None)
        assert (sh.lo, sh.hi) == shard_range(64, r, world)
        assert sh.tok_lo == offs[sh.lo] and sh.tok_hi == offs[sh.hi] and sh.n_docs_total == 64
        assert sh.local.strides == reference_strides(torch.tensor(dl)) == z["strides"].tolist()
        assert sh.local.doclens.tolist() == dl[sh.lo:sh.hi]
        assert sh.local.d_pad_len.tolist() == z["pad_len"][sh.lo:sh.hi].tolist()
        rows.append(sh.local.tensor[:sh.local.num_embeddings])
        docs += sh.local.doclens.tolist()
    assert docs == dl and torch.equal(torch.cat(rows), whole)


def test_load_shard_errors(tmp_path):
    from colbert_amd.sharded import load_shard
    write_index(str(tmp_path), "masked_query_rerank.npz")          # 12 docs
    with pytest.raises(ValueError, match="fewer than the 13 ranks"):
        load_shard(str(tmp_path), 7, 13, device="cpu", score_fn=lambda *a: None, topk_fn=cpu_topk)
    sh = load_shard(str(tmp_path), 7, 8, device="cpu", score_fn=lambda *a: None, topk_fn=cpu_topk)  # 12 docs on 8 ranks: nobody is empty
    assert (sh.lo, sh.hi) == (10, 12)
    with pytest.raises(ValueError, match="rank and world"):
        load_shard(str(tmp_path), device="cpu")                     # no process group to take them from


def test_exchange_payload_is_12_bytes_per_entry():
    """[nq, k] fp32 scores + [nq, k] int64 pids = 12 B per entry in ONE collective (world 1, gloo)."""
    from colbert_amd import sharded
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(_free_port())
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        seen = []
        orig = dist.all_gather_into_tensor

        def spy(out, inp, group=None):
            seen.append(inp.numel() * inp.element_size())
            return orig(out, inp, group=group)
        dist.all_gather_into_tensor = spy
        try:
            s = torch.tensor([[3.5, NEG_INF, -0.0], [1e-30, 2.0, 7.0]])
            p = torch.tensor([[2 ** 40 + 5, -1, 7], [0, 2 ** 31, 9]])
            gs, gp = sharded.all_gather_topk(s, p, 1)
        finally:
            dist.all_gather_into_tensor = orig
        assert seen == [2 * 3 * 12]
        assert torch.equal(gs[0].view(torch.int32), s.view(torch.int32)) and torch.equal(gp[0], p)
    finally:
        dist.destroy_process_group()


# ---------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize("index_dtype", [torch.float16, torch.float32])
def test_shards_from_files_equal_unsharded_on_gpu(tmp_path, index_dtype):
    """N = 2, 4, 8 shards loaded one after another from the same files, each on the HIP scorer; their local top-k merged
    with ``merge_gathered`` equal the unsharded HIP ranker's result bit for bit -- for rank_forward's candidate lists
    (ragged_rerank_64: also against the golden) and for the batched driver on global token rows (masked query tokens)."""
    import colbert_amd
    from colbert_amd.sharded import load_shard, merge_gathered
    z, parts, dls = write_index(str(tmp_path), "ragged_rerank_64.npz")
    dl = dls[0] + dls[1]
    whole = colbert_amd.ColbertRanker(index_path=str(tmp_path), device="cuda:0", index_dtype=index_dtype)
    Q = torch.from_numpy(z["Q"]).cuda()
    pids = z["pids"].tolist()
    exp_p, exp_s = whole.rank_forward(Q, pids, depth=10)
    assert exp_p == z["top10_pids"].tolist() and np.allclose(exp_s, z["top10_scores"], rtol=0, atol=1e-4)
    g = torch.Generator().manual_seed(3)
    nq, Lq = 5, 32
    Qb = torch.nn.functional.normalize(torch.randn(nq, Lq, 128, generator=g), dim=-1).cuda()
    keep = (torch.rand(nq, Lq, generator=g) > 0.3).long().cuda()
    ids = cover_ids(dl, Lq, 8, nq, seed=5)
    ids[:, :, 6:] = torch.randint(0, sum(dl), (nq, Lq, 2), generator=g)
    ids = ids.cuda()
    exp_lists = colbert_amd.retrieve_batch(whole, Qb, keep, topk=20, embedding_ids=ids)
    cand = torch.tensor(pids, device="cuda").view(1, -1)
    Qt = Q.permute(0, 2, 1).contiguous()
    for N in (2, 4, 8):
        shards = [load_shard(str(tmp_path), r, N, device="cuda:0", index_dtype=index_dtype) for r in range(N)]
        assert all(s.local.strides == whole.strides for s in shards)
        # rank_forward's leg: every shard's local top-10 of the same global list
        tops = [s.local_topk(Qt, cand, 10) for s in shards]
        gs, gp = torch.stack([t[1] for t in tops]), torch.stack([t[0] for t in tops])
        mp_, ms_ = merge_gathered(gs, gp, 10, whole.topk)
        assert mp_[0].tolist() == exp_p and ms_[0].tolist() == exp_s, N
        # a world-1 ShardedRanker call end to end (no process group: no exchange), on the first shard's own docs
        own = [p for p in pids if shards[0].lo <= p < shards[0].hi]
        op, os_ = shards[0].rank_forward(Q, own, depth=64)
        ep, es = whole.rank_forward(Q, own, depth=64)
        assert (op, os_) == (ep, es), N
        # the batched driver's leg on GLOBAL token rows
        from colbert_amd.retriever import prepare_embedding_ids
        keep_b, ids_m = prepare_embedding_ids(Qb.device, Qb, keep, ids)
        tops = [s.local_retrieve_topk(Qb, keep_b, ids_m.reshape(nq, -1), 20) for s in shards]
        gs, gp = torch.stack([t[1] for t in tops]), torch.stack([t[0] for t in tops])
        mp_, ms_ = merge_gathered(gs, gp, 20, whole.topk)
        for qi, (ep, es) in enumerate(exp_lists):
            n = len(ep)
            assert ms_[qi, :n].tolist() == es and sorted(mp_[qi, :n].tolist()) == sorted(ep), (N, qi)
            by = dict(zip(ep, es))
            assert all(by[p] == s for p, s in zip(mp_[qi, :n].tolist(), ms_[qi, :n].tolist())), (N, qi)
            assert bool((mp_[qi, n:] == -1).all())
