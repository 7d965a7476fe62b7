"""GPU parity tests added in round 3: counted candidate rows (the device-built dense work list behind the doc-sharded
step and ANN pid lists), reference-faithful failure modes of rank_forward, the 768-dim ragged fp16 deployment shape.
Tolerances as in test_gpu_parity.py: fp32 |d| <= 1e-4, 16-bit inputs |d| <= 1e-3."""
import ctypes

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

ATOL32 = 1e-4
ATOL16 = 1e-3


@pytest.fixture(scope="module")
def ca():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    import colbert_amd
    return colbert_amd


def nrm(gen, *shape):
    return F.normalize(torch.randn(*shape, generator=gen), dim=-1)


def _counted_rows(gen, nq, ncand, ndocs, counts):
    """Rows as maxsim_shard_candidates / maxsim_embedding_ids_to_pids write them: counts[q] live pids first, then -1."""
    cand = torch.full((nq, ncand), -1, dtype=torch.int64)
    for q in range(nq):
        c = int(counts[q])
        cand[q, :c] = torch.randint(0, ndocs, (c,), generator=gen)
    return cand


# ------------------------------------------------------------------------------------------------------
# counted rows: maxsim_rerank_counted / maxsim_topk_counted == maxsim_rerank_ex / maxsim_topk, bit for bit
# ------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("cfg", [
    dict(dtype=torch.float32, mode="exact", Lq=32, lo=30, hi=180, nq=37, ncand=300),
    dict(dtype=torch.float32, mode="exact", Lq=12, lo=30, hi=180, nq=16, ncand=200),     # one 16-column block
    dict(dtype=torch.float32, mode="exact", Lq=70, lo=40, hi=120, nq=9, ncand=90),       # three query slices (accumulating passes)
    dict(dtype=torch.float32, mode="bf16x3", Lq=32, lo=30, hi=180, nq=12, ncand=150),
    dict(dtype=torch.float32, mode="fast", Lq=32, lo=30, hi=180, nq=12, ncand=150),
    dict(dtype=torch.float16, mode="exact", Lq=32, lo=25, hi=180, nq=20, ncand=260),     # the reference's storage dtype
    dict(dtype=torch.bfloat16, mode="exact", Lq=32, lo=25, hi=180, nq=10, ncand=128),
    dict(dtype=torch.float32, mode="exact", Lq=32, lo=1, hi=12, nq=8, ncand=100),        # ragged short docs: static grid fallback
    dict(dtype=torch.float32, mode="exact", Lq=8, lo=8, hi=8, nq=40, ncand=500),         # uniform 8-token docs: the fixed-length list kernel
    dict(dtype=torch.float32, mode="exact", Lq=40, lo=16, hi=16, nq=9, ncand=130),       # uniform 16, two blocks, query slices
    dict(dtype=torch.float32, mode="exact", Lq=20, lo=4, hi=4, nq=12, ncand=300),        # uniform 4
    dict(dtype=torch.float16, mode="exact", Lq=32, lo=10, hi=60, nq=6, ncand=70, h=768), # wide rows: static grid fallback
], ids=lambda c: f"{str(c['dtype']).split('.')[-1]}-{c['mode']}-Lq{c['Lq']}-h{c.get('h', 128)}-{c['lo']}_{c['hi']}")
def test_counted_rows_equal_full_width_rows(ca, cfg):
    gen = torch.Generator().manual_seed(31)
    h, ndocs, nq, ncand = cfg.get("h", 128), 500, cfg["nq"], cfg["ncand"]
    doclens = torch.randint(cfg["lo"], cfg["hi"] + 1, (ndocs,), generator=gen).tolist()
    if cfg["lo"] != cfg["hi"]:
        doclens[7] = 0                                                                   # an empty doc scores 0 (not in a uniform index)
    emb = nrm(gen, sum(doclens), h).to(cfg["dtype"])
    r = ca.ColbertRanker(parts=[emb], parts_doclens=[doclens], dim=h, index_dtype=cfg["dtype"], fp32_mode=cfg["mode"])
    counts = torch.randint(0, ncand + 1, (nq,), generator=gen)
    counts[0], counts[1], counts[2] = 0, ncand, 1                                        # empty row, full row, one doc
    cand = _counted_rows(gen, nq, ncand, ndocs, counts)
    cand[1, 3] = 7                                                                       # the empty doc
    cand[1, 5] = ndocs + 9                                                               # an out-of-range pid INSIDE the live part: -inf
    Q = nrm(gen, nq, cfg["Lq"], h)
    qm = (torch.rand(nq, cfg["Lq"], generator=gen) > 0.2).long()
    qm[:, 0] = 1
    for kw in (dict(), dict(q_mask=qm)):
        full = r.score_candidates(Q, cand.cuda(), **kw)
        cnt = r.score_candidates(Q, cand.cuda(), cand_count=counts.cuda(), **kw)
        assert torch.equal(full.cpu(), cnt.cpu())                                        # incl. the -inf tails
        assert bool(torch.isinf(cnt[0]).all()) and float(cnt[1, 5]) == float("-inf")
        assert cfg["lo"] == cfg["hi"] or float(cnt[1, 3]) == 0.0
    for k in (1, 10, ncand):
        p0, s0 = r.topk(full, cand.cuda(), k)
        p1, s1 = r.topk(cnt, cand.cuda(), k, counts.cuda())
        assert torch.equal(s0.cpu(), s1.cpu()) and torch.equal(p0.cpu(), p1.cpu())


@pytest.mark.parametrize("dtype,qdtype,h,Lq", [(torch.float16, torch.float32, 768, 32), (torch.float16, torch.float16, 768, 32),
                                               (torch.bfloat16, torch.float32, 256, 20), (torch.float32, torch.float32, 384, 32),
                                               (torch.float16, torch.float32, 768, 45), (torch.float16, torch.float32, 200, 32)])
def test_counted_rows_on_wide_embeddings(ca, dtype, qdtype, h, Lq):
    """Counted rows on rows wider than 128 dims (the reference's default deployment is dim 768): the LDS-query kernel walks a
    work list of WORKGROUP items; bit-identical to the static grid.  (h = 200 is not a multiple of 128: static fallback.)"""
    gen = torch.Generator().manual_seed(77)
    ndocs, nq, ncand = 300, 23, 180
    doclens = (torch.randn(ndocs, generator=gen) * 60 + 150).round().clamp(1, 384).long().tolist()
    emb = nrm(gen, sum(doclens), h).to(dtype)
    r = ca.ColbertRanker(parts=[emb], parts_doclens=[doclens], dim=h, index_dtype=dtype)
    counts = torch.randint(0, ncand + 1, (nq,), generator=gen)
    counts[0], counts[1], counts[2] = 0, ncand, 1
    cand = _counted_rows(gen, nq, ncand, ndocs, counts)
    cand[1, 7] = ndocs + 1
    Q = nrm(gen, nq, Lq, h).to(qdtype)
    qm = (torch.rand(nq, Lq, generator=gen) > 0.2).long()
    qm[:, 0] = 1
    for kw in (dict(), dict(q_mask=qm)):
        full = r.score_candidates(Q, cand.cuda(), **kw)
        cnt = r.score_candidates(Q, cand.cuda(), cand_count=counts.cuda(), **kw)
        assert torch.equal(full.cpu(), cnt.cpu())
    # a big, few-candidates-per-row launch (one rank's share of an 8-way step) and a tiny one
    for nq2, ncand2, mean in ((200, 1000, 125), (2, 40, 9)):
        c2 = torch.randint(max(0, mean - mean // 4), mean + mean // 4 + 1, (nq2,), generator=gen)
        cd = _counted_rows(gen, nq2, ncand2, ndocs, c2)
        Q2 = nrm(gen, nq2, Lq, h).to(qdtype)
        assert torch.equal(r.score_candidates(Q2, cd.cuda()).cpu(), r.score_candidates(Q2, cd.cuda(), cand_count=c2.cuda()).cpu())


def test_counted_topk_on_long_rows(ca):
    """Rows longer than 2048 slots (ANN pid lists: up to 16384 = the reference's BSIZE) with a live count: the sort kernel
    works on the next power of two >= the count; same lists as the uncounted top-k, incl. ties, k > count and count = 0."""
    gen = torch.Generator().manual_seed(8)
    nq, n = 9, 5000
    counts = torch.tensor([0, 1, 2, 3, 1000, 2047, 2048, 2049, 5000], dtype=torch.int32)
    scores = torch.full((nq, n), float("-inf"))
    pids = torch.full((nq, n), -1, dtype=torch.int64)
    for q in range(nq):
        c = int(counts[q])
        scores[q, :c] = torch.randint(0, 50, (c,), generator=gen).float() / 7        # heavy ties
        pids[q, :c] = torch.randperm(100000, generator=gen)[:c]
    lib = ca._lib.lib
    st = torch.cuda.current_stream().cuda_stream
    sc, pc, cc = scores.cuda(), pids.cuda(), counts.cuda()
    for k in (1, 100, 3000):
        o = [torch.empty(nq, k, device="cuda"), torch.empty(nq, k, dtype=torch.int64, device="cuda"),
             torch.empty(nq, k, device="cuda"), torch.empty(nq, k, dtype=torch.int64, device="cuda")]
        assert lib.maxsim_topk(sc.data_ptr(), pc.data_ptr(), nq, n, k, o[0].data_ptr(), o[1].data_ptr(), st) == 0
        assert lib.maxsim_topk_counted(sc.data_ptr(), pc.data_ptr(), cc.data_ptr(), nq, n, k, o[2].data_ptr(), o[3].data_ptr(), st) == 0
        assert torch.equal(o[0].cpu(), o[2].cpu()) and torch.equal(o[1].cpu(), o[3].cpu()), k


def test_counted_rows_small_and_large_launches(ca):
    """The builder picks the docs per wave item ON THE DEVICE from the number of live candidates: a launch with few of
    them gets shorter items (more waves), one with many the ~1.4 k-token streams.  Both ends against the static grid."""
    gen = torch.Generator().manual_seed(5)
    ndocs, h = 3000, 128
    doclens = torch.randint(100, 181, (ndocs,), generator=gen).tolist()
    emb = nrm(gen, sum(doclens), h)
    r = ca.ColbertRanker(parts=[emb], parts_doclens=[doclens], dim=h, index_dtype=torch.float32)
    for nq, ncand, mean in ((1, 1000, 1000), (3, 64, 5), (300, 400, 50), (64, 1000, 125)):
        counts = torch.randint(max(0, mean - mean // 4), min(ncand, mean + mean // 4) + 1, (nq,), generator=gen)
        cand = _counted_rows(gen, nq, ncand, ndocs, counts)
        Q = nrm(gen, nq, 32, h)
        full = r.score_candidates(Q, cand.cuda())
        cnt = r.score_candidates(Q, cand.cuda(), cand_count=counts.cuda())
        assert torch.equal(full.cpu(), cnt.cpu()), (nq, ncand, mean)


def test_worklist_layout(ca):
    """The work list itself: items cover every live slot exactly once, in row order, with at most D docs each, and the
    header agrees (maxsim_worklist.h).  Read back through the C ABI's scratch buffer."""
    gen = torch.Generator().manual_seed(9)
    ndocs, h, nq, ncand = 2000, 128, 50, 333
    doclens = [150] * ndocs
    emb = nrm(gen, sum(doclens), h)
    r = ca.ColbertRanker(parts=[emb], parts_doclens=[doclens], dim=h, index_dtype=torch.float32)
    counts = torch.randint(0, ncand + 1, (nq,), generator=gen).to(torch.int32)
    cand = _counted_rows(gen, nq, ncand, ndocs, counts).cuda()
    Q = nrm(gen, nq, 32, h).cuda()
    lib = ca._lib.lib
    nbytes = int(lib.maxsim_worklist_bytes(nq, ncand))
    wl = torch.zeros(nbytes // 4, dtype=torch.int32, device="cuda")
    scores = torch.empty(nq, ncand, device="cuda")
    cc = counts.cuda()
    rc = lib.maxsim_rerank_counted(ctypes.byref(r._iv), Q.data_ptr(), 0, None, None, cand.data_ptr(), cc.data_ptr(), nq, ncand,
                                   32, scores.data_ptr(), wl.data_ptr(), nbytes, torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    w = wl.cpu()
    total, D, live = int(w[0]), int(w[1]), int(w[2])
    assert live == int(counts.sum()) and 1 <= D <= 64
    starts = w[16:16 + nq + 1].tolist()
    off = (16 + nq + 1 + 3) & ~3
    items = w[off:off + 2 * total].view(total, 2)
    assert starts[0] == 0 and starts[-1] == total
    for q in range(nq):
        c = int(counts[q])
        mine = items[starts[q]:starts[q + 1]]
        assert len(mine) == (c + D - 1) // D
        pos = 0
        for qq, packed in mine.tolist():
            b, n = packed & 0xfffff, packed >> 20
            assert qq == q and b == pos and 1 <= n <= D
            pos += n
        assert pos == c


def test_sharded_local_topk_uses_counted_rows(ca):
    """ShardedRanker.local_topk (shard filter -> counted rerank -> counted top-k) == the same steps on full-width rows."""
    from colbert_amd.sharded import ShardedRanker, shard_candidates
    gen = torch.Generator().manual_seed(13)
    ndocs, h = 1500, 128
    doclens = torch.randint(60, 181, (ndocs,), generator=gen).tolist()
    emb = nrm(gen, sum(doclens), h).half()
    r = ca.ColbertRanker(parts=[emb], parts_doclens=[doclens], dim=h, index_dtype=torch.float16)
    lo, hi = 3 * ndocs, 4 * ndocs                                        # rank 3 of 8
    cand = torch.randint(0, 8 * ndocs, (40, 1000), generator=gen).cuda()
    Q = nrm(gen, 40, 32, h)
    sh = ShardedRanker(r, lo, hi)
    p1, s1 = sh.local_topk(Q, cand, 100)
    loc, gp = shard_candidates(cand, lo, hi)
    p0, s0 = r.topk(r.score_candidates(Q, loc), gp, 100)
    assert torch.equal(p0.cpu(), p1.cpu()) and torch.equal(s0.cpu(), s1.cpu())
    assert int((p1 >= 0).sum()) > 0 and bool(((p1 < 0) | ((p1 >= lo) & (p1 < hi))).all())


# ------------------------------------------------------------------------------------------------------
# rank_forward's failure modes: `self.doclens[pids]`, colbert_ranker.py:88
# ------------------------------------------------------------------------------------------------------
def test_rank_forward_bad_pids_raise_like_the_reference(ca):
    """A pid >= n_docs or < -n_docs raises IndexError where the reference's `self.doclens[pids]` does (the oracle's
    restatement raises the same), for every input form and on both host paths (CPython glue / ctypes); nothing is
    returned with a -inf score.  A negative pid in [-n_docs, -1] indexes from the end as torch indexing does, and the
    caller's own value comes back (colbert_ranker.py:129)."""
    from colbert_amd import ranker as rk
    from oracle.maxsim_oracle import RefRanker
    gen = torch.Generator().manual_seed(23)
    ndocs = 200
    doclens = torch.randint(1, 120, (ndocs,), generator=gen).tolist()
    emb = nrm(gen, sum(doclens), 128).half()
    ref = RefRanker([emb], [doclens], dim=128)
    r = ca.ColbertRanker(parts=[emb], parts_doclens=[doclens], dim=128)
    Q = nrm(gen, 32, 128).unsqueeze(0).permute(0, 2, 1)
    good = torch.randperm(ndocs, generator=gen)[:50].tolist()
    for bad in (ndocs, ndocs + 12345, -ndocs - 1, 2 ** 40):
        pids = good[:20] + [bad] + good[20:]
        with pytest.raises(IndexError):
            ref.rank_forward(Q, pids, depth=10)
        forms = [pids, torch.tensor(pids), torch.tensor(pids).cuda(), [np.int64(p) for p in pids]]
        for form in forms:
            with pytest.raises(IndexError, match="out of bounds for dimension 0 with size 200"):
                r.rank_forward(Q, form, depth=10)
        saved, rk._fastrank = rk._fastrank, None
        try:
            with pytest.raises(IndexError):
                r.rank_forward(Q, pids, depth=10)
        finally:
            rk._fastrank = saved
        with pytest.raises(IndexError):
            r.rank_forward(Q, pids, depth=10, output_D_embedding=True)
    # negative pids wrap: doc ndocs + p is scored, p itself is returned
    neg = good[:10] + [-1, -ndocs, -7] + good[10:30]
    pos = [p if p >= 0 else p + ndocs for p in neg]
    ep, es = ref.rank_forward(Q, pos, depth=33)
    for form in (neg, torch.tensor(neg), torch.tensor(neg).cuda()):
        gp, gs = r.rank_forward(Q, form, depth=33)
        np.testing.assert_allclose(np.array(gs), np.array(es), rtol=0, atol=ATOL32)
        assert [p if p >= 0 else p + ndocs for p in gp] == ep and set(gp) == set(neg)
    # the good list still works after the failures (no state was left behind in the per-thread workspace)
    assert r.rank_forward(Q, good, depth=10)[0] == ref.rank_forward(Q, good, depth=10)[0]


# ------------------------------------------------------------------------------------------------------
# the reference's default deployment shape: dim 768 (proj_conf/dense.yaml:6-8), fp16 index (encoder.py:175), ragged docs
# ------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", [torch.float16, torch.float32])
def test_ragged_rerank_768_golden(ca, golden, dtype):
    g = golden("ragged_rerank_768")
    parts = [g["part0"], g["part1"]]
    pdl = [g["doclens0"].tolist(), g["doclens1"].tolist()]
    r = ca.ColbertRanker(parts=parts, parts_doclens=pdl, dim=768, index_dtype=dtype)
    assert r.strides == g["strides"].tolist()
    assert torch.equal(r.d_pad_len.cpu().long(), g["pad_len"])
    pids = g["pids"]
    for key, exp in (("Q", "expected_scores"), ("Q_neg", "expected_scores_neg")):
        sc = r.score_candidates(g[key].permute(0, 2, 1), pids.view(1, -1))
        torch.testing.assert_close(sc.cpu()[0], g[exp], rtol=0, atol=ATOL32)        # fp32 query on an fp16-exact index: fp32 tolerance
    tp, ts = r.rank_forward(g["Q"], pids.tolist(), depth=10)
    assert tp == g["top10_pids"].tolist()
    np.testing.assert_allclose(ts, g["top10_scores"].numpy(), rtol=0, atol=ATOL32)


def test_ragged_768_fp16_batch_vs_oracle(ca):
    """A bigger ragged 768-dim fp16 index (doclens 1..384, the reference's doc_maxlen) scored in batches against the
    oracle's float64 closed form: every candidate of every query, incl. counted rows (static-grid fallback for h != 128)."""
    from oracle.maxsim_oracle import ragged_scores_f64
    gen = torch.Generator().manual_seed(41)
    ndocs, h = 120, 768
    doclens = (torch.randn(ndocs, generator=gen) * 80 + 200).round().clamp(1, 384).long().tolist()
    doclens[:4] = [384, 1, 383, 33]
    emb = nrm(gen, sum(doclens), h).half()
    r = ca.ColbertRanker(parts=[emb], parts_doclens=[doclens], dim=h, index_dtype=torch.float16)
    nq, ncand = 5, 60
    Q = nrm(gen, nq, 32, h)
    cand = torch.stack([torch.randperm(ndocs, generator=gen)[:ncand] for _ in range(nq)])
    sc = r.score_candidates(Q, cand.cuda()).cpu()
    for qi in range(nq):
        exp = ragged_scores_f64(emb, r.doclens, r.doclens_pfxsum, r.d_pad_len.cpu(), Q[qi], cand[qi].tolist())
        np.testing.assert_allclose(sc[qi].numpy(), exp, rtol=0, atol=ATOL32)
    # 16-bit query handed over in the index's own type: the 16-bit-input tolerance
    sc16 = r.score_candidates(Q.half(), cand.cuda()).cpu()
    for qi in range(nq):
        exp = ragged_scores_f64(emb, r.doclens, r.doclens_pfxsum, r.d_pad_len.cpu(), Q[qi].half().float(), cand[qi].tolist())
        np.testing.assert_allclose(sc16[qi].numpy(), exp, rtol=0, atol=ATOL16)


def test_masked_query_rerank_golden(ca, golden):
    """The batched driver's q_mask / q_len predicates against the fixture the imported reference wrote by compacting the
    query first (keep_nonzero, training_utils.py:48-53) and scoring what is left: one launch for the three queries with
    their masks, the compacted queries one by one, and retrieve-style counted rows -- fp16 index as the reference stores it."""
    g = golden("masked_query_rerank")
    r = ca.ColbertRanker(parts=[g["part0"], g["part1"]], parts_doclens=[g["doclens0"].tolist(), g["doclens1"].tolist()],
                         dim=128, index_dtype=torch.float16)
    pids = g["pids"].cuda()
    cand = pids[None, :].repeat(3, 1)
    exp = g["expected_scores"]
    sc = r.score_candidates(g["Q"], cand, q_mask=g["q_word_mask"]).cpu()
    torch.testing.assert_close(sc, exp, rtol=0, atol=ATOL32)
    cnt = torch.full((3,), cand.size(1), dtype=torch.int32, device="cuda")
    sc2 = r.score_candidates(g["Q"], cand, q_mask=g["q_word_mask"], cand_count=cnt).cpu()
    assert torch.equal(sc2, sc)
    for q in range(3):                                   # the reference's own order of operations: compact, then score
        live = g["q_word_mask"][q].bool()
        one = r.score_candidates(g["Q"][q][live][None], cand[q:q + 1]).cpu()
        torch.testing.assert_close(one[0], exp[q], rtol=0, atol=ATOL32)
        tp, ts = r.rank_forward(g["Q"][q][live][None].permute(0, 2, 1), g["pids"].tolist(), depth=5)
        order = torch.argsort(exp[q], descending=True)[:5]
        np.testing.assert_allclose(ts, exp[q][order].numpy(), rtol=0, atol=ATOL32)
    # query 1's mask is a prefix mask plus holes: q_len alone (the prefix form) must NOT equal it, q_len + q_mask must
    ql = torch.tensor([32, 20, 32], dtype=torch.int32)
    sc3 = r.score_candidates(g["Q"], cand, q_len=ql, q_mask=g["q_word_mask"]).cpu()
    torch.testing.assert_close(sc3, exp, rtol=0, atol=ATOL32)


# ------------------------------------------------------------------------------------------------------
# the online call (maxsim_rank_forward: rerank + counting top-k, polled completion) against the batched entry points
# ------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype,Lq", [(torch.float32, 32), (torch.float32, 9), (torch.float16, 32), (torch.bfloat16, 32)])
def test_rank_forward_equals_batched_entry_points(ca, dtype, Lq):
    """rank_forward (one C call: the small-launch forms of the rerank kernel -- docs of a 16-bit index split over waves,
    the query staged through LDS -- then the counting top-k whose last workgroup stores the polled completion word) returns
    exactly what the batched entry points return for the same list, for list lengths around the ranking-group size (16), the
    reference's ~1000, the counting kernel's limit (2048) and beyond (the sort kernel), call after call on one workspace (its
    counter returns to zero), with depth below, at and above the list length."""
    gen = torch.Generator().manual_seed(3)
    ndocs, h = 4000, 128
    doclens = torch.randint(100, 301, (ndocs,), generator=gen).tolist()            # long enough for the split form (fp16 / bf16)
    emb = nrm(gen, sum(doclens), h).to(dtype)
    r = ca.ColbertRanker(parts=[emb], parts_doclens=[doclens], dim=h, index_dtype=dtype)
    assert r.device.index is not None                    # "cuda" is spelled out, so the caller's cuda:N tensors compare equal
    q = nrm(gen, Lq, h)
    Q = q.unsqueeze(0).permute(0, 2, 1)
    # the reference-shaped call (list of pids, permuted view of a contiguous fp32 q on the index's device) goes straight to
    # the C glue; every other form through the general preamble: same lists
    pl = torch.randint(0, ndocs, (500,), generator=gen).tolist()
    fast = r.rank_forward(Q.cuda(), pl, depth=50)
    r._fast_ok = False
    try:
        assert r.rank_forward(Q.cuda(), pl, depth=50) == fast
    finally:
        r._fast_ok = True
    assert r.rank_forward(Q, pl, depth=50) == fast and r.rank_forward(Q.cuda().contiguous(), pl, depth=50) == fast
    for n in (1, 2, 15, 16, 17, 100, 333, 1000, 1000, 2047, 2048, 2049, 3000, 7):
        pids = torch.randint(0, ndocs, (n,), generator=gen).tolist()                # duplicates allowed: equal scores, tie order by position
        for depth in (1, 10, n, n + 5):
            k = min(depth, n)
            gp, gs = r.rank_forward(Q, pids, depth=depth)
            cand = torch.tensor(pids).view(1, -1).cuda()
            tp, ts = r.topk(r.score_candidates(q.unsqueeze(0), cand), cand, k)
            assert gp == tp[0].tolist() and gs == ts[0].tolist(), (n, depth)


def test_rank_forward_concurrent_threads_and_streams(ca):
    """Three host threads, each with its own workspace (counter, pinned buffers, completion word) and its own stream, issue
    calls at the same time: one call's completion must not be confused with another's."""
    import threading
    gen = torch.Generator().manual_seed(4)
    ndocs, h = 3000, 128
    doclens = torch.randint(60, 181, (ndocs,), generator=gen).tolist()
    emb = nrm(gen, sum(doclens), h)
    r = ca.ColbertRanker(parts=[emb], parts_doclens=[doclens], dim=h, index_dtype=torch.float32)
    q = nrm(gen, 32, h)
    Q = q.unsqueeze(0).permute(0, 2, 1).cuda()
    lists = [torch.randint(0, ndocs, (1000,), generator=gen).tolist() for _ in range(8)]
    exp = []
    for pl in lists:
        cand = torch.tensor(pl).view(1, -1).cuda()
        tp, ts = r.topk(r.score_candidates(q.unsqueeze(0), cand), cand, 100)
        exp.append((tp[0].tolist(), ts[0].tolist()))
    torch.cuda.synchronize()
    errors = []

    def worker(tid):
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            for it in range(60):
                i = (it + tid) % len(lists)
                if r.rank_forward(Q, lists[i], depth=100) != exp[i]:
                    errors.append((tid, it))
    ths = [threading.Thread(target=worker, args=(t,)) for t in range(3)]
    [t.start() for t in ths]
    [t.join() for t in ths]
    assert not errors, errors[:5]


# ------------------------------------------------------------------------------------------------------
# uniform short docs (the multi-view configuration, BASELINE configs[3]): the fixed-length kernel
# ------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("L", [4, 8, 16])
@pytest.mark.parametrize("Lq", [8, 16, 27, 40])
def test_uniform_short_docs_kernel_is_bit_identical(ca, L, Lq):
    """An index whose every doc has exactly L tokens runs k_maxsim_stream_uni (doc length compiled in); its scores equal,
    bit for bit, those of the general half-tile kernel on the same index (the same ranker with the promise withdrawn:
    uniform_len = 0) and agree with the oracle's closed form; padding slots, out-of-range pids, row widths that leave the
    last tile / wave / workgroup partly filled, q_mask, queries longer than 32 tokens (accumulating passes)."""
    import copy
    from oracle.maxsim_oracle import ragged_scores_f64
    gen = torch.Generator().manual_seed(100 + L)
    ndocs, h = 5000, 128
    doclens = [L] * ndocs
    emb = nrm(gen, sum(doclens), h)
    r = ca.ColbertRanker(parts=[emb], parts_doclens=[doclens], dim=h, index_dtype=torch.float32)
    assert r._iv.uniform_len == L
    g = copy.copy(r)                                     # the same index without the promise -> general kernels
    g._iv = r._index_view()
    g._iv.uniform_len = 0
    g._iv_ref = ctypes.byref(g._iv)
    g._iv_addr = ctypes.addressof(g._iv)
    for nq, ncand in ((1, 1), (3, 7), (2, 65), (5, 1000), (300, 130), (2, 2049)):
        Q = nrm(gen, nq, Lq, h)
        cand = torch.randint(0, ndocs, (nq, ncand), generator=gen)
        if ncand > 3:
            cand[0, 1] = -1                              # padding slot
            cand[-1, ncand - 2] = ndocs + 3              # out of range
        qm = (torch.rand(nq, Lq, generator=gen) > 0.25).long()
        qm[:, 0] = 1
        for kw in (dict(), dict(q_mask=qm)):
            a = r.score_candidates(Q, cand.cuda(), **kw).cpu()
            b = g.score_candidates(Q, cand.cuda(), **kw).cpu()
            assert torch.equal(a, b), (L, Lq, nq, ncand, list(kw))
        if ncand > 3:
            assert float(a[0, 1]) == float("-inf") and float(a[-1, ncand - 2]) == float("-inf")
        if nq <= 5 and ncand <= 1000:
            for qi in range(nq):
                ok = [(c, p) for c, p in enumerate(cand[qi].tolist()) if 0 <= p < ndocs][:40]
                exp = ragged_scores_f64(emb, r.doclens, r.doclens_pfxsum, r.d_pad_len.cpu(), Q[qi][qm[qi].bool()], [p for _, p in ok])
                np.testing.assert_allclose(a[qi, [c for c, _ in ok]].numpy(), exp, rtol=0, atol=ATOL32)


def test_uniform_promise_only_for_uniform_unpadded_indexes(ca):
    """ColbertRanker sets uniform_len only when every doc has the same length AND that length is the only length bucket
    (no doc padded): a shard re-bucketed by the strides of a ragged whole index does not qualify."""
    gen = torch.Generator().manual_seed(2)
    emb = nrm(gen, 8 * 50, 128)
    r = ca.ColbertRanker(parts=[emb], parts_doclens=[[8] * 50], dim=128, index_dtype=torch.float32)
    assert r._iv.uniform_len == 8
    r.set_strides([8, 20])                               # global strides of a whole index with longer docs elsewhere
    assert r._iv.uniform_len == 8                        # still unpadded: 8 is a bucket of its own
    r.set_strides([12, 20])                              # every doc now sits in the 12-bucket: padded, 0-floor applies
    assert r._iv.uniform_len == 0
    rag = ca.ColbertRanker(parts=[emb], parts_doclens=[[8] * 49 + [4, 4]], dim=128, index_dtype=torch.float32)
    assert rag._iv.uniform_len == 0


# ------------------------------------------------------------------------------------------------------
# the stride collectives of the sharded path on RCCL (backend "nccl"), as far as one GPU allows
# ------------------------------------------------------------------------------------------------------
def test_stride_sync_collectives_run_on_rccl(ca, monkeypatch):
    """`global_strides` / `assert_strides_agree` issue their all_reduces on CUDA tensors when the backend is nccl (= RCCL).
    A one-GPU box cannot hold two RCCL ranks, so this runs them on a world-1 RCCL group with the world size reported as
    2 (the `> 1` guards would skip them otherwise): placement, dtypes and the results' way back to the host are exercised;
    the multi-rank semantics are covered by the world-2 gloo test."""
    import socket
    import torch.distributed as dist
    from colbert_amd import sharded
    from colbert_amd.ranker import reference_strides
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=torch.device("cuda", torch.cuda.current_device()))
    try:
        monkeypatch.setattr(sharded.dist, "get_world_size", lambda group=None: 2)
        gen = torch.Generator().manual_seed(6)
        doclens = torch.randint(1, 181, (500,), generator=gen)
        gs = sharded.global_strides(doclens)                                  # all_reduce(MAX) + all_reduce(SUM) on cuda tensors
        assert gs == reference_strides(doclens)
        sharded.assert_strides_agree(gs)                                      # all_reduce(MAX) of (strides, -strides)
        emb = nrm(gen, int(doclens.sum()), 128).half()
        r = ca.ColbertRanker(parts=[emb], parts_doclens=[doclens.tolist()], dim=128)
        # the shipped constructor: both, then set_strides (placement given: its all_gather needs the real world size)
        sr = sharded.ShardedRanker(r, 0, 500, n_docs_total=500, tok_lo=0)
        assert r.strides == gs and sr._world() == 2
    finally:
        monkeypatch.undo()
        dist.destroy_process_group()


# ------------------------------------------------------------------------------------------------------
# small launches on wide embeddings: docs split over waves (the online call on the default deployment, dim 768)
# ------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype,qdtype,h", [(torch.float16, torch.float32, 768), (torch.float16, torch.float16, 768),
                                            (torch.bfloat16, torch.float32, 256), (torch.float32, torch.float32, 384)])
def test_split_small_launch_on_wide_embeddings_is_bit_identical(ca, dtype, qdtype, h):
    """One query x up to ~1000 ragged docs of up to 384 tokens: each doc streamed by 2 or 4 waves (whole-tile slices, parked
    per-token maxima, the unsplit wave's floor + sum tree).  Scores equal, bit for bit, those of the batched (unsplit) kernel
    -- the same candidates scored inside a 64-query batch -- for list lengths that leave teams / workgroups partly filled,
    docs shorter than a tile (empty slices), empty docs, padding slots, the 0-floor ("negative" query) and q_mask."""
    gen = torch.Generator().manual_seed(55)
    ndocs = 700
    doclens = (torch.randn(ndocs, generator=gen) * 90 + 200).round().clamp(1, 384).long().tolist()
    doclens[:6] = [384, 1, 31, 32, 33, 0]
    emb = torch.randn(sum(doclens), h, generator=gen) * 0.1
    emb[:, 0] += 1.0
    emb = F.normalize(emb, dim=-1).to(dtype)
    r = ca.ColbertRanker(parts=[emb], parts_doclens=[doclens], dim=h, index_dtype=dtype)
    Qall = nrm(gen, 64, 32, h)
    Qall[1] = 0.0
    Qall[1, :, 0] = -1.0                                     # every similarity negative: the 0-floor decides
    Qall = Qall.to(qdtype)
    qm = (torch.rand(64, 32, generator=gen) > 0.2).long()
    qm[:, 0] = 1
    for n in (1, 2, 3, 4, 5, 9, 333, 1000):
        cand = torch.randint(0, ndocs, (64, n), generator=gen)
        cand[:, : min(n, 6)] = torch.arange(min(n, 6))       # the special docs first
        if n > 8:
            cand[:, 7] = -1
        big = r.score_candidates(Qall, cand.cuda()).cpu()                         # 64 queries: the unsplit kernel
        bigm = r.score_candidates(Qall, cand.cuda(), q_mask=qm).cpu()
        for qi in (0, 1, 5):
            one = r.score_candidates(Qall[qi:qi + 1], cand[qi:qi + 1].cuda()).cpu()   # 1 query: the split form
            assert torch.equal(one[0], big[qi]), (n, qi)
            onem = r.score_candidates(Qall[qi:qi + 1], cand[qi:qi + 1].cuda(), q_mask=qm[qi:qi + 1]).cpu()
            assert torch.equal(onem[0], bigm[qi]), (n, qi, "mask")
    # and through the online call
    pids = torch.randint(0, ndocs, (500,), generator=gen).tolist()
    gp, gs = r.rank_forward(Qall[0:1].permute(0, 2, 1), pids, depth=50)
    c = torch.tensor(pids).view(1, -1).cuda()
    tp, ts = r.topk(r.score_candidates(Qall[:8], c.expand(8, -1).contiguous())[:1], c, 50)
    assert gp == tp[0].tolist() and gs == ts[0].tolist()
