"""GPU parity tests added in round 2: the batched driver's query-token mask (keep_nonzero), the retrieve driver, the
packed doc table, the fused rank_forward entry point, doc-shard candidate filtering with GLOBAL strides, and the
NaN / Inf contract.  Tolerances as in test_gpu_parity.py: fp32 |d| <= 1e-4, 16-bit inputs |d| <= 1e-3."""
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


def _random_index(gen, ndocs, h, lo, hi, dtype=torch.float16):
    doclens = torch.randint(lo, hi + 1, (ndocs,), generator=gen).tolist()
    half = ndocs // 2
    pdl = [doclens[:half], doclens[half:]]
    parts = [nrm(gen, sum(d), h).to(dtype) for d in pdl]
    return parts, pdl


def _holey_mask(gen, nq, Lq):
    """q_active_padding as tokenize_seqs emits it (tokenizers.py:36): zeros for punctuation / [SEP] in the MIDDLE of the
    sequence and for the padding tail; one query keeps everything, one keeps a single token."""
    m = (torch.rand(nq, Lq, generator=gen) > 0.3).long()
    m[:, 0] = 1
    for qi in range(nq):
        tail = int(torch.randint(0, Lq // 2 + 1, (1,), generator=gen))
        if tail:
            m[qi, Lq - tail:] = 0
    m[0] = 1
    if nq > 1:
        m[1] = 0
        m[1, Lq // 2] = 1
    return m


# ------------------------------------------------------------------------------------------------------
# q_mask: dense_server_client.py:45 + training_utils.py:48-53 for a batch, without compacting Q
# ------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("cfg", [
    dict(ndocs=300, h=128, lo=1, hi=180, nq=6, ncand=97, Lq=32, dtype=torch.float32),      # f32 MFMA, two column blocks
    dict(ndocs=120, h=128, lo=1, hi=60, nq=4, ncand=64, Lq=12, dtype=torch.float32),       # one column block
    dict(ndocs=64, h=128, lo=8, hi=8, nq=4, ncand=64, Lq=8, dtype=torch.float32),          # half-tile kernel
    dict(ndocs=64, h=128, lo=8, hi=12, nq=4, ncand=64, Lq=32, dtype=torch.float32),        # half-tile kernel, two blocks
    dict(ndocs=200, h=128, lo=1, hi=90, nq=4, ncand=50, Lq=32, dtype=torch.float16),       # reference storage dtype
    dict(ndocs=200, h=128, lo=1, hi=90, nq=4, ncand=50, Lq=32, dtype=torch.bfloat16),
    dict(ndocs=14, h=768, lo=100, hi=256, nq=3, ncand=14, Lq=32, dtype=torch.bfloat16, qdtype=torch.bfloat16),  # LDS-query kernel
    dict(ndocs=30, h=256, lo=1, hi=70, nq=3, ncand=30, Lq=32, dtype=torch.float32),
    dict(ndocs=40, h=24, lo=1, hi=40, nq=3, ncand=30, Lq=12, dtype=torch.float32),         # generic kernel
    dict(ndocs=60, h=128, lo=1, hi=90, nq=3, ncand=50, Lq=40, dtype=torch.float32),        # Lq > 32: two query slices
])
def test_q_mask_equals_keep_nonzero(ca, cfg):
    from oracle.maxsim_oracle import RefRanker, keep_nonzero
    gen = torch.Generator().manual_seed(1000 + cfg["ndocs"] + cfg["h"] + cfg["Lq"])
    parts, pdl = _random_index(gen, cfg["ndocs"], cfg["h"], cfg["lo"], cfg["hi"], cfg["dtype"])
    ref = RefRanker(parts, pdl, dim=cfg["h"], index_dtype=cfg["dtype"])
    r = ca.ColbertRanker(parts=parts, parts_doclens=pdl, dim=cfg["h"], index_dtype=cfg["dtype"])
    Q = nrm(gen, cfg["nq"], cfg["Lq"], cfg["h"])
    Qdev = Q.to(cfg["qdtype"]) if "qdtype" in cfg else Q
    Q = Qdev.float()
    mask = _holey_mask(gen, cfg["nq"], cfg["Lq"])
    cand = torch.stack([torch.randperm(cfg["ndocs"], generator=gen)[:cfg["ncand"]] for _ in range(cfg["nq"])])
    atol = ATOL32
    sc = r.score_candidates(Qdev, cand, q_mask=mask).cpu()
    tp, ts = r.rerank_batch(Qdev, cand, depth=10, q_mask=mask)
    for qi in range(cfg["nq"]):
        q_live, _ = keep_nonzero(Q[qi], mask[qi])                               # what the reference hands to search()
        exp = ref.all_scores(q_live.unsqueeze(0).permute(0, 2, 1), cand[qi].tolist())
        torch.testing.assert_close(sc[qi], exp, rtol=0, atol=atol)
        ep, es = ref.rank_forward(q_live.unsqueeze(0).permute(0, 2, 1), cand[qi].tolist(), depth=10)
        np.testing.assert_allclose(ts[qi].cpu().numpy(), np.array(es), rtol=0, atol=atol)
    # q_len and q_mask combine (token scored iff below q_len AND kept), and a float / bool mask means the same
    q_len = torch.tensor([max(1, cfg["Lq"] - 3)] * cfg["nq"], dtype=torch.int32)
    sc2 = r.score_candidates(Qdev, cand, q_len=q_len, q_mask=mask.bool()).cpu()
    m2 = mask.clone()
    m2[:, cfg["Lq"] - 3:] = 0
    sc3 = r.score_candidates(Qdev, cand, q_mask=m2.float()).cpu()
    assert torch.equal(sc2, sc3)
    # an all-ones mask is bit-identical to no mask
    assert torch.equal(r.score_candidates(Qdev, cand, q_mask=torch.ones_like(mask)).cpu(), r.score_candidates(Qdev, cand).cpu())


def test_retrieve_batch_equals_reference_loop(ca):
    """colbert_amd.retrieve_batch against the reference's per-query loop (dense_server_client.py:44-48 ->
    faiss_indexers.py:224-235 -> colbert_ranker.py:176-229, 75-137) on the oracle, with a toy exact-search ANN."""
    from oracle.maxsim_oracle import RefRanker, keep_nonzero
    gen = torch.Generator().manual_seed(321)
    doclens = torch.randint(1, 60, (300,), generator=gen).tolist()
    parts = [nrm(gen, sum(doclens), 128).half()]
    ref = RefRanker(parts, [doclens], dim=128)
    r = ca.ColbertRanker(parts=parts, parts_doclens=[doclens], dim=128)
    bs, Lq, depth = 5, 32, 8
    Q = nrm(gen, bs, Lq, 128)
    mask = _holey_mask(gen, bs, Lq)
    index_f = parts[0].float()
    emb2pid = torch.repeat_interleave(torch.arange(len(doclens)), torch.tensor(doclens))     # colbert_ranker.py:163-174

    def ann(q_live, faiss_depth):                              # stands in for faiss_index.search (third-party)
        return (q_live.float().cpu() @ index_f.T).topk(faiss_depth, dim=-1).indices

    out = ca.retrieve_batch(r, Q, mask, topk=10, ann_search=ann, faiss_depth=depth)
    # the other calling form: ids for every token (dead rows hold garbage that must be ignored)
    all_ids = (Q.reshape(-1, 128) @ index_f.T).topk(depth, dim=-1).indices.view(bs, Lq, depth)
    out2 = ca.retrieve_batch(r, Q, mask, topk=10, embedding_ids=all_ids)
    assert len(out) == bs
    for qi in range(bs):
        q_live, _ = keep_nonzero(Q[qi], mask[qi])                                            # dense_server_client.py:45
        ids = ann(q_live, depth)                                                              # colbert_ranker.py:183-210
        pids = sorted(set(emb2pid[ids.reshape(-1)].tolist()))                                 # :212-229 (set -> any order)
        ep, es = ref.rank_forward(q_live.unsqueeze(0).permute(0, 2, 1), pids, depth=10)      # faiss_indexers.py:232-234
        for got in (out[qi], out2[qi]):
            assert len(got[0]) == len(ep) == len(got[1])
            np.testing.assert_allclose(np.array(got[1]), np.array(es), rtol=0, atol=ATOL32)
            assert got[0] == ep


# ------------------------------------------------------------------------------------------------------
# packed doc table, the fused rank_forward entry, C-ABI level
# ------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("h,dtype", [(128, torch.float32), (128, torch.float16), (768, torch.bfloat16), (24, torch.float32)])
def test_doc_table_is_bit_identical_to_the_three_arrays(ca, h, dtype):
    gen = torch.Generator().manual_seed(7 + h)
    parts, pdl = _random_index(gen, 150, h, 0, 120, dtype)
    r = ca.ColbertRanker(parts=parts, parts_doclens=pdl, dim=h, index_dtype=dtype)
    Q = nrm(gen, 4, 32, h).cuda()
    cand = torch.stack([torch.randperm(150, generator=gen)[:90] for _ in range(4)]).cuda()
    cand[0, 3], cand[1, 7] = -1, 10 ** 9                                 # padding slots
    with_table = r.score_candidates(Q, cand)
    assert r._iv.doc_table is not None
    tbl = r.d_doc_table.view(torch.int64).view(-1, 2).cpu()
    assert torch.equal(tbl[:, 0], r.d_offsets.cpu())
    assert torch.equal(tbl[:, 1] & 0xffffffff, r.d_doclens.cpu().long()) and torch.equal(tbl[:, 1] >> 32, r.d_pad_len.cpu().long())
    L = ca._lib.lib
    plain = torch.empty_like(with_table)
    rc = L.maxsim_rerank(r.tensor.data_ptr(), r._iv.index_dtype, r.num_embeddings, r.d_offsets.data_ptr(),
                         r.d_doclens.data_ptr(), r.d_pad_len.data_ptr(), r.n_docs, Q.data_ptr(), 0, None, cand.data_ptr(),
                         4, 90, 32, h, plain.data_ptr(), torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    torch.cuda.synchronize()
    assert torch.equal(plain, with_table)


def test_rank_forward_input_forms_and_threads(ca, golden):
    """rank_forward through the one-call entry (pinned in/out buffers): python list, CPU tensor and device tensor pids
    give the same lists; concurrent host threads have their own workspaces."""
    import threading
    from oracle.maxsim_oracle import RefRanker
    gen = torch.Generator().manual_seed(17)
    parts, pdl = _random_index(gen, 500, 128, 1, 180, torch.float16)
    ref = RefRanker(parts, pdl, dim=128)
    r = ca.ColbertRanker(parts=parts, parts_doclens=pdl, dim=128)
    q = nrm(gen, 32, 128)
    Q = q.unsqueeze(0).permute(0, 2, 1)                      # [1, h, Lq], a permuted view as faiss_indexers.py:232-233
    pids = torch.randperm(500, generator=gen)[:333].tolist()
    ep, es = ref.rank_forward(Q, pids, depth=100)
    for form in (pids, torch.tensor(pids), torch.tensor(pids).cuda(), np.array(pids).tolist()):
        for Qin in (Q, Q.cuda(), Q.contiguous().cuda().half()):
            gp, gs = r.rank_forward(Qin, form, depth=100)
            atol = ATOL32 if Qin.dtype == torch.float32 else 5e-3
            np.testing.assert_allclose(np.array(gs), np.array(es), rtol=0, atol=atol)
            if Qin.dtype == torch.float32:
                assert gp == ep
    assert r.rank_forward(Q, pids[:3], depth=10)[0] == ref.rank_forward(Q, pids[:3], depth=10)[0]     # depth > n
    with pytest.raises(AssertionError):
        r.rank_forward(Q, [], depth=10)
    errors = []

    def worker(tid):
        g2 = torch.Generator().manual_seed(100 + tid)
        for it in range(20):
            pp = torch.randperm(500, generator=g2)[:200].tolist()
            a = r.rank_forward(Q, pp, depth=20)
            b = ref.rank_forward(Q, pp, depth=20)
            if a[0] != b[0] or max(abs(x - y) for x, y in zip(a[1], b[1])) > ATOL32:
                errors.append((tid, it))
    ths = [threading.Thread(target=worker, args=(t,)) for t in range(4)]
    [t.start() for t in ths]
    [t.join() for t in ths]
    assert not errors, errors
    # the CPython glue (csrc/fastrank.c) and the ctypes path make the same library call: identical lists; a list the
    # glue refuses (numpy integers) takes the general path (out-of-range pids: test_rank_forward_bad_pids_raise_like_the_reference)
    from colbert_amd import ranker as rk
    assert rk._fastrank is not None
    fast = r.rank_forward(Q, pids, depth=100)
    weird = r.rank_forward(Q, [np.int64(p) for p in pids], depth=100)
    saved, rk._fastrank = rk._fastrank, None
    try:
        slow = r.rank_forward(Q, pids, depth=100)
    finally:
        rk._fastrank = saved
    assert fast == slow == weird
    assert type(fast[0]) is list and type(fast[0][0]) is int and type(fast[1][0]) is float


# ------------------------------------------------------------------------------------------------------
# doc-sharded path on the GPU: shard filter kernel, global strides
# ------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("nq,ncand", [(1, 1), (3, 64), (5, 1000), (2, 257), (4, 16384)])
def test_shard_candidates_kernel_equals_stable_partition(ca, nq, ncand):
    from colbert_amd.sharded import shard_candidates
    gen = torch.Generator().manual_seed(nq * 7 + ncand)
    cand = torch.randint(0, 4000, (nq, ncand), generator=gen)
    cand[0, 0] = -1
    for lo, hi in ((0, 4000), (1000, 2000), (3999, 4000), (5000, 6000), (100, 100)):
        loc_c, gp_c = shard_candidates(cand, lo, hi)                     # torch reference (CPU ranks)
        loc_g, gp_g = shard_candidates(cand.cuda(), lo, hi)              # maxsim_shard_candidates
        assert torch.equal(loc_g.cpu(), loc_c) and torch.equal(gp_g.cpu(), gp_c)
    # in place + counts through the C ABI
    c = cand.cuda().clone()
    cnt = torch.empty(nq, dtype=torch.int32, device="cuda")
    rc = ca._lib.lib.maxsim_shard_candidates(c.data_ptr(), nq, ncand, 1000, 2000, c.data_ptr(), None, cnt.data_ptr(),
                                             torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    exp_loc, _ = shard_candidates(cand, 1000, 2000)
    assert torch.equal(c.cpu(), exp_loc) and torch.equal(cnt.cpu().long(), (exp_loc >= 0).sum(1))


def test_sharded_shard_scores_equal_the_unsharded_reference(ca):
    """Each shard of a ragged index, bucketed by the GLOBAL strides, scores its docs exactly as ONE reference ranker over
    the whole index does -- including the 'negative' query whose score is the zero-padding floor alone
    (colbert_ranker.py:90,108-109); bucketed by its own percentiles it does not."""
    from colbert_amd.sharded import ShardedRanker, merge_gathered, shard_range
    from colbert_amd.ranker import reference_strides
    from oracle.maxsim_oracle import RefRanker
    gen = torch.Generator().manual_seed(77)
    ndocs, h, world = 400, 128, 2
    doclens = torch.cat([torch.randint(1, 40, (ndocs // 2,), generator=gen), torch.randint(30, 181, (ndocs // 2,), generator=gen)]).tolist()
    emb = torch.randn(sum(doclens), h, generator=gen) * 0.05
    emb[:, 0] += 1.0
    emb = F.normalize(emb, dim=-1).half()
    whole = RefRanker([emb], [doclens], dim=h)
    Q = nrm(gen, 3, 32, h)
    Q[2] = 0.0
    Q[2, :, 0] = -1.0
    cand = torch.stack([torch.randperm(ndocs, generator=gen)[:150] for _ in range(3)])
    # make sure the negative query meets docs whose length EQUALS a stride (no padding, hence no floor) under the global
    # strides and under the second shard's own percentiles
    own1 = reference_strides(torch.tensor(doclens[ndocs // 2:]))
    special = [p for p in range(ndocs) if doclens[p] in whole.strides or (p >= ndocs // 2 and doclens[p] in own1)][:40]
    rest = [p for p in cand[2].tolist() if p not in special]
    cand[2] = torch.tensor((special + rest)[:150])
    exp = torch.stack([whole.all_scores(Q[qi:qi + 1].permute(0, 2, 1), cand[qi].tolist()) for qi in range(3)])
    offs = [0]
    for d in doclens:
        offs.append(offs[-1] + d)
    gstr = reference_strides(torch.tensor(doclens))
    assert gstr == whole.strides
    tops, tops_own = [], []
    for rank in range(world):
        lo, hi = shard_range(ndocs, rank, world)
        kw = dict(parts=[emb[offs[lo]:offs[hi]]], parts_doclens=[doclens[lo:hi]], dim=h)
        r = ca.ColbertRanker(strides=gstr, **kw)
        r_own = ca.ColbertRanker(**kw)
        assert r.strides == whole.strides and r_own.strides != whole.strides
        assert torch.equal(r.d_pad_len.cpu().long(), whole.bucket_strides(list(range(lo, hi))))
        tops.append(ShardedRanker(r, lo, hi).local_topk(Q, cand.cuda(), 150))
        tops_own.append(ShardedRanker(r_own, lo, hi).local_topk(Q, cand.cuda(), 150))
    def merged(t):
        gs = torch.stack([x[1] for x in t])
        gp = torch.stack([x[0] for x in t])
        return merge_gathered(gs, gp, 150, r.topk)
    mp, ms = merged(tops)
    es, ei = torch.sort(exp, dim=1, descending=True, stable=True)
    np.testing.assert_allclose(ms.cpu().numpy(), es.numpy(), rtol=0, atol=ATOL32)
    lookup = [dict(zip(cand[qi].tolist(), exp[qi].tolist())) for qi in range(3)]
    for qi in range(3):
        assert sorted(mp[qi].tolist()) == sorted(cand[qi].tolist())
        for p, v in zip(mp[qi].tolist(), ms[qi].tolist()):
            assert abs(lookup[qi][p] - v) <= ATOL32
    assert bool((exp[2] == 0).any()) and bool((exp[2] < -1).any())      # the floor decides: both kinds are present
    mp_own, ms_own = merged(tops_own)
    # per-shard percentiles would change scores: a different set of docs of the negative query gets the 0-floor
    assert max(abs(lookup[2][p] - v) for p, v in zip(mp_own[2].tolist(), ms_own[2].tolist())) > 1.0


# ------------------------------------------------------------------------------------------------------
# NaN / Inf contract (DESIGN.md "Non-finite inputs")
# ------------------------------------------------------------------------------------------------------
def test_non_finite_inputs_contract(ca):
    """The reference's torch.max / sum propagate NaN (BaseModel.py:44-45).  The encoder emits L2-normalised, finite rows
    (BaseModel.py:26), so finite inputs are the contract; what the kernels do beyond it is pinned here:
      * +-Inf similarities behave as in torch wherever torch's result is not NaN (max picks +inf, the sum is +-inf);
      * a NaN similarity is IGNORED by the running max (v_max_f32 = IEEE maxNum), where torch returns NaN: a doc whose
        every other token is finite gets the max over those tokens; a query token whose similarities are all NaN
        contributes -inf, 0 with the padding floor."""
    from oracle.maxsim_oracle import ref_score
    gen = torch.Generator().manual_seed(3)
    doclens = [20, 20, 20, 20, 20, 20]
    emb = nrm(gen, sum(doclens), 128)
    emb[5, 3] = float("inf")            # doc 0: one +inf component
    emb[25, 7] = float("-inf")          # doc 1: one -inf component
    emb[45, 9] = float("nan")           # doc 2: one NaN component in one token
    Q = nrm(gen, 1, 32, 128).abs() + 0.01         # strictly positive components: inf * q is +-inf, never NaN
    r = ca.ColbertRanker(parts=[emb], parts_doclens=[doclens], dim=128, index_dtype=torch.float32)
    sc = r.score_candidates(Q, torch.arange(6).view(1, 6)).cpu()[0]
    D = emb.view(6, 20, 128)
    exp = ref_score(Q, D, torch.ones(1, 32), torch.ones(6, 20))[0]
    assert sc[0] == float("inf") and exp[0] == float("inf")
    # -inf token: torch's max ignores it too (other tokens are larger): finite, equal
    assert abs(sc[1] - exp[1]) <= ATOL32
    assert torch.isnan(exp[2])                                   # torch propagates
    alt = emb.clone()
    alt[45] = alt[44]                                            # the kernel's answer: the NaN token never wins a max
    exp2 = ref_score(Q, alt.view(6, 20, 128)[2:3], torch.ones(1, 32), torch.ones(1, 20))[0, 0]
    assert torch.isfinite(sc[2]) and abs(sc[2] - exp2) <= ATOL32
    np.testing.assert_allclose(sc[3:].numpy(), exp[3:].numpy(), rtol=0, atol=ATOL32)
    # the dense operator seam behaves the same way
    out = ca.score(Q.cuda(), D.cuda(), torch.ones(1, 32).cuda(), torch.ones(6, 20).cuda()).cpu()[0]
    assert out[0] == float("inf") and torch.isfinite(out[2]) and abs(out[1] - exp[1]) <= ATOL32


# ------------------------------------------------------------------------------------------------------
# small launches: docs split over several waves, top-k fused into the rerank launch
# ------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16])
@pytest.mark.parametrize("lo,hi", [(180, 180), (1, 180), (129, 400), (300, 700)])
def test_split_launch_is_bit_identical_to_the_batched_kernel(ca, dtype, lo, hi):
    """A doc streamed by 2 or 4 waves (one query x few candidates) scores bit-for-bit what the one-wave-per-doc kernel
    gives the same (query, doc) pair inside a big batch: max is exact and the sum over query tokens uses the same tree."""
    gen = torch.Generator().manual_seed(lo * 7 + hi)
    nd = 700
    doclens = torch.randint(lo, hi + 1, (nd,), generator=gen).tolist()
    doclens[5] = 0
    parts = [nrm(gen, sum(doclens), 128).to(dtype)]
    r = ca.ColbertRanker(parts=parts, parts_doclens=[doclens], dim=128, index_dtype=dtype)
    Q = nrm(gen, 40, 32, 128).cuda()
    cand = torch.stack([torch.randperm(nd, generator=gen)[:600] for _ in range(40)]).cuda()
    cand[0, 11], cand[0, 12] = -1, 5
    big = r.score_candidates(Q, cand)                                  # 24000 docs: the regular kernel
    for n in (1, 2, 3, 37, 200, 600):                                  # one query x n docs: the split forms
        small = r.score_candidates(Q[:1], cand[:1, :n])
        assert torch.equal(small, big[:1, :n]), n
    two = r.score_candidates(Q[:2], cand[:2, :100])
    assert torch.equal(two, big[:2, :100])


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
def test_fused_rank_forward_repeated_calls(ca, dtype):
    """The one-launch rank_forward (top-k in the last workgroup): many calls in a row on one workspace (its counters must
    come back to zero), list lengths around the sort's sizes, every call equal to rerank + top-k done separately."""
    gen = torch.Generator().manual_seed(5)
    nd = 3000
    doclens = torch.randint(100, 181, (nd,), generator=gen).tolist()
    parts = [nrm(gen, sum(doclens), 128).to(dtype)]
    r = ca.ColbertRanker(parts=parts, parts_doclens=[doclens], dim=128, index_dtype=dtype)
    for it, n in enumerate([1000, 1, 2, 255, 256, 257, 1000, 1023, 1024, 1025, 2047, 2048, 2049, 1000, 3000, 1000, 1000]):
        q = nrm(gen, 32, 128)
        Q = q.unsqueeze(0).permute(0, 2, 1)
        pids = torch.randperm(nd, generator=gen)[:n].tolist()
        depth = [100, 10, 1, 3000][it % 4]
        gp, gs = r.rank_forward(Q, pids, depth=depth)
        sc = r.score_candidates(q.unsqueeze(0), torch.tensor([pids]))
        es, ei = torch.sort(sc[0].cpu(), descending=True, stable=True)
        k = min(depth, n)
        assert gs == es[:k].tolist(), (it, n)
        assert gp == [pids[i] for i in ei[:k].tolist()], (it, n)


@pytest.mark.parametrize("ncand,k", [(1000, 100), (1000, 10), (1000, 1), (1000, 256), (1000, 257), (1000, 1000), (2048, 100),
                                      (2047, 256), (300, 100), (256, 256), (100, 100), (3, 2), (1, 1), (1500, 300)])
def test_topk_selection_with_heavy_ties(ca, ncand, k):
    """The selection top-k (radix descent + counting, k <= 256) and the sort it falls back to, on score rows made of a
    handful of distinct values (ties across the k-th place are the rule), +-inf, and signed zeros: equal to a stable
    descending sort -- score desc, then lower list position (any tie order conforms to the reference's unstable sort,
    colbert_ranker.py:128; this pins OURS)."""
    gen = torch.Generator().manual_seed(ncand * 3 + k)
    nq = 6
    s = torch.randn(nq, ncand, generator=gen)
    s[0] = torch.randint(0, 3, (ncand,), generator=gen).float()             # three distinct values
    s[1] = 7.25                                                             # all equal
    s[2] = torch.randint(0, 2, (ncand,), generator=gen).float() * float("inf")
    s[2][s[2] != s[2]] = float("-inf")                                     # 0 * inf = nan -> -inf: rows of +inf / -inf
    s[3, ::2] = -s[3, ::2].abs()
    s[3, : ncand // 3] = 0.0
    s[3, 1: ncand // 3: 2] = -0.0                                           # -0.0 sorts below +0.0 in key order
    s[4] = (s[4] * 4).round() / 4                                           # quantised: many boundary ties
    pids = torch.randint(0, 10 ** 12, (nq, ncand), generator=gen)
    r = ca.ColbertRanker(parts=[torch.zeros(4, 8)], parts_doclens=[[1, 1, 1, 1]], dim=8)
    tp, ts = r.topk(s.cuda(), pids.cuda(), k)
    # expected: total order on (orderable(score) desc, position asc) -- for floats this is a stable descending sort,
    # except that -0.0 ranks below +0.0 (bit pattern order), which torch.sort treats as equal: compare through the bits
    keys = s.clone().view(torch.int32).long()
    keys = torch.where(keys < 0, -(keys & 0x7fffffff) - 1, keys)            # monotone map of the float order, -0 < +0
    es_k, ei = torch.sort(keys, dim=1, descending=True, stable=True)
    assert torch.equal(tp.cpu(), torch.gather(pids, 1, ei[:, :k]))
    assert torch.equal(ts.cpu().view(torch.int32), torch.gather(s, 1, ei[:, :k]).view(torch.int32))


# ------------------------------------------------------------------------------------------------------
# the GEMM-blocked all-pairs kernel (training-form forward) against the streaming kernel and the oracle
# ------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("cfg", [
    dict(nq=50, nd=70, Lq=32, Ld=384, h=768, dtype=torch.bfloat16),     # the reference's training shape, R = 3
    dict(nq=23, nd=45, Lq=32, Ld=256, h=128, dtype=torch.float16),      # R = 2 exactly full
    dict(nq=16, nd=90, Lq=9, Ld=129, h=128, dtype=torch.bfloat16),      # R = 2, one row past 128; two K slices (the least)
    dict(nq=31, nd=33, Lq=32, Ld=128, h=192, dtype=torch.float16),      # R = 1, three K slices
    dict(nq=8, nd=200, Lq=32, Ld=1, h=256, dtype=torch.bfloat16),       # one-token docs
])
def test_allpairs_kernel_matches_streaming_kernel_and_oracle(ca, cfg):
    """maxsim_score_dense_fwd through the C ABI: float32 masks take the GEMM-blocked kernel (maxsim_allpairs.h), int64 masks
    the streaming kernel (same values: 0/1).  Scores agree with each other and with the oracle on the rounded inputs
    (1e-3: 16-bit inputs); arg-max indices are torch.max's (first maximal token) wherever the top two similarities of a
    (query token, doc) pair are not within rounding of each other."""
    from oracle.maxsim_oracle import ref_score
    from colbert_amd.scoring import _DT, _MDT
    L = ca._lib.lib
    gen = torch.Generator().manual_seed(cfg["nq"] * 31 + cfg["Ld"])
    nq, nd, Lq, Ld, h, dt = cfg["nq"], cfg["nd"], cfg["Lq"], cfg["Ld"], cfg["h"], cfg["dtype"]
    Q = nrm(gen, nq, Lq, h).to(dt)
    D = nrm(gen, nd, Ld, h).to(dt)
    qm = (torch.rand(nq, Lq, generator=gen) > 0.15).long()
    dm = (torch.rand(nd, Ld, generator=gen) > 0.25).long()
    dm[:, 0] = 1
    Qd, Dd = Q.cuda(), D.cuda()
    assert L.maxsim_score_dense_kernel(nq, nd, Lq, Ld, h, _DT[dt], _MDT[torch.float32]) == 1     # the GEMM-blocked kernel ...
    assert L.maxsim_score_dense_kernel(nq, nd, Lq, Ld, h, _DT[dt], _MDT[torch.int64]) == 0       # ... and the streaming one
    res = {}
    for name, mt in (("gemm", torch.float32), ("stream", torch.int64)):
        qmd, dmd = qm.to(mt).cuda(), dm.to(mt).cuda()
        out = torch.empty(nq, nd, device="cuda")
        arg = torch.full((nq, nd, Lq), -7, dtype=torch.int32, device="cuda")
        rc = L.maxsim_score_dense_fwd(Qd.data_ptr(), Dd.data_ptr(), qmd.data_ptr(), dmd.data_ptr(), nq, nd, Lq, Ld, h,
                                      _DT[dt], _MDT[mt], out.data_ptr(), arg.data_ptr(), torch.cuda.current_stream().cuda_stream)
        assert rc == 0
        torch.cuda.synchronize()
        res[name] = (out.cpu(), arg.cpu())
    exp = ref_score(Q.float(), D.float(), qm, dm)
    torch.testing.assert_close(res["gemm"][0], exp, rtol=0, atol=ATOL16)
    torch.testing.assert_close(res["gemm"][0], res["stream"][0], rtol=0, atol=ATOL16)
    # arg-max against torch.max on the float64 similarities, skipping near-ties
    sim = torch.einsum("qmh,dnh->qdmn", (Q.float() * qm[..., None]).double(), (D.float() * dm[..., None]).double())
    top2 = sim.topk(min(2, Ld), dim=-1).values
    clear = (top2[..., 0] - top2[..., -1] > 1e-4) if Ld > 1 else torch.ones(nq, nd, Lq, dtype=torch.bool)
    ref_idx = sim.argmax(-1).to(torch.int32)
    for name in ("gemm", "stream"):
        got = res[name][1]
        assert bool((got >= 0).all()) and bool((got < Ld).all()), name
        assert bool((got[clear] == ref_idx[clear]).all()), name
    assert float(clear.float().mean()) > 0.5
    # a query token of weight 0 has every similarity 0: torch.max returns the first token
    dropped = (qm == 0)[:, None, :].expand(nq, nd, Lq)
    for name in ("gemm", "stream"):
        assert bool((res[name][1][dropped] == 0).all()), name


@pytest.mark.gpu
@pytest.mark.parametrize("Ld,dt", [(128, torch.float16), (250, torch.bfloat16), (384, torch.bfloat16)])
def test_allpairs_kernel_prefix_masks(ca, Ld, dt):
    """The masks the reference actually builds (tokenizers.py:57: ones up to the doc's length, zeros after): the kernel's
    epilogue treats 32-row blocks of all ones / all zeros / tile padding without a multiplication.  Lengths on and
    around the block boundaries, length 1, full length and an all-zero row; against the streaming kernel (int64 masks)
    and the oracle, arg-max included (a doc's zero-weight rows have similarity 0: they win over negative similarities,
    and the first of them is the arg-max)."""
    from oracle.maxsim_oracle import ref_score
    from colbert_amd.scoring import _DT, _MDT
    L = ca._lib.lib
    gen = torch.Generator().manual_seed(5 + Ld)
    nq, nd, Lq, h = 20, 72, 32, 128
    Q = nrm(gen, nq, Lq, h).to(dt)
    D = nrm(gen, nd, Ld, h).to(dt)
    lens = torch.randint(1, Ld + 1, (nd,), generator=gen)
    edge = [0, 1, 31, 32, 33, 63, 64, 65, 96, 127, 128, Ld - 1, Ld]
    lens[:len(edge)] = torch.tensor([min(e, Ld) for e in edge])
    dm = (torch.arange(Ld)[None, :] < lens[:, None]).long()
    qlen = torch.randint(1, Lq + 1, (nq,), generator=gen)
    qm = (torch.arange(Lq)[None, :] < qlen[:, None]).long()
    Qd, Dd = Q.cuda(), D.cuda()
    assert L.maxsim_score_dense_kernel(nq, nd, Lq, Ld, h, _DT[dt], _MDT[torch.float32]) == 1
    res = {}
    for name, mt in (("gemm", torch.float32), ("stream", torch.int64)):
        qmd, dmd = qm.to(mt).cuda(), dm.to(mt).cuda()
        out = torch.empty(nq, nd, device="cuda")
        arg = torch.full((nq, nd, Lq), -7, dtype=torch.int32, device="cuda")
        rc = L.maxsim_score_dense_fwd(Qd.data_ptr(), Dd.data_ptr(), qmd.data_ptr(), dmd.data_ptr(), nq, nd, Lq, Ld, h,
                                      _DT[dt], _MDT[mt], out.data_ptr(), arg.data_ptr(), torch.cuda.current_stream().cuda_stream)
        assert rc == 0
        torch.cuda.synchronize()
        res[name] = (out.cpu(), arg.cpu())
    exp = ref_score(Q.float(), D.float(), qm, dm)
    torch.testing.assert_close(res["gemm"][0], exp, rtol=0, atol=ATOL16)
    torch.testing.assert_close(res["gemm"][0], res["stream"][0], rtol=0, atol=ATOL16)
    sim = torch.einsum("qmh,dnh->qdmn", (Q.float() * qm[..., None]).double(), (D.float() * dm[..., None]).double())
    top2 = sim.topk(2, dim=-1).values
    live = (qm == 1)[:, None, :].expand(nq, nd, Lq)
    clear = (top2[..., 0] - top2[..., 1] > 1e-4) & live
    # the maximum is a zero-weight row's exact 0 (every live similarity negative): the first such row
    zero_wins = live & (top2[..., 0] == 0) & (lens < Ld)[None, :, None]
    ref_idx = sim.argmax(-1).to(torch.int32)
    first_zero = lens.to(torch.int32)[None, :, None].expand(nq, nd, Lq)
    for name in ("gemm", "stream"):
        got = res[name][1]
        assert bool((got >= 0).all()) and bool((got < Ld).all()), name
        assert bool((got[clear] == ref_idx[clear]).all()), name
        assert bool((got[zero_wins] == first_zero[zero_wins]).all()), name
        assert bool((got[~live] == 0).all()), name
    assert float(clear.float().mean()) > 0.3


@pytest.mark.gpu
def test_allpairs_kernel_random_shapes_against_streaming_kernel(ca):
    """Thirty random shapes the GEMM-blocked kernel serves (every R, partial query blocks, nd not a multiple of the 8 XCDs,
    Lq < 32, 2..16 K slices, both 16-bit types, prefix / random / no masks): scores equal the streaming kernel's within the
    16-bit tolerance, arg-max equal wherever the winner is clear."""
    from colbert_amd.scoring import _DT, _MDT
    L = ca._lib.lib
    rng = np.random.default_rng(2024)
    served = 0
    for case in range(30):
        dt = [torch.bfloat16, torch.float16][case % 2]
        Lq = int(rng.choice([1, 7, 16, 31, 32]))
        Ld = int(rng.integers(1, 385))
        h = 64 * int(rng.integers(2, 17))
        nq = int(rng.integers(1, 60))
        nd = int(rng.integers(1, 80))
        while nd * ((nq + 7) // 8) < 128:
            nd += 9
        gen = torch.Generator().manual_seed(case)
        Q = nrm(gen, nq, Lq, h).to(dt).cuda()
        D = nrm(gen, nd, Ld, h).to(dt).cuda()
        mode = case % 3
        if mode == 0:
            qm = dm = None
        elif mode == 1:
            qm = (torch.arange(Lq)[None, :] < torch.randint(1, Lq + 1, (nq, 1), generator=gen)).long()
            dm = (torch.arange(Ld)[None, :] < torch.randint(1, Ld + 1, (nd, 1), generator=gen)).long()
        else:
            qm = (torch.rand(nq, Lq, generator=gen) > 0.2).long()
            dm = (torch.rand(nd, Ld, generator=gen) > 0.3).long()
        assert L.maxsim_score_dense_kernel(nq, nd, Lq, Ld, h, _DT[dt], 0 if qm is None else _MDT[torch.float32]) == 1
        served += 1
        res = []
        for mt in (torch.float32, torch.int64):
            if qm is None and mt == torch.int64:
                # no masks: the streaming kernel is reached through all-ones int64 masks
                qa, da = torch.ones(nq, Lq, dtype=mt).cuda(), torch.ones(nd, Ld, dtype=mt).cuda()
            elif qm is None:
                qa = da = None
            else:
                qa, da = qm.to(mt).cuda(), dm.to(mt).cuda()
            out = torch.empty(nq, nd, device="cuda")
            arg = torch.full((nq, nd, Lq), -7, dtype=torch.int32, device="cuda")
            rc = L.maxsim_score_dense_fwd(Q.data_ptr(), D.data_ptr(), None if qa is None else qa.data_ptr(),
                                          None if da is None else da.data_ptr(), nq, nd, Lq, Ld, h, _DT[dt],
                                          0 if qa is None else _MDT[mt], out.data_ptr(), arg.data_ptr(),
                                          torch.cuda.current_stream().cuda_stream)
            assert rc == 0, (case, rc)
            res.append((out, arg))
        # the same kernel without arg-max tracking (maxsim_score_dense: the no-grad operator): the same scores, bit for bit
        qa, da = (None, None) if qm is None else (qm.float().cuda(), dm.float().cuda())
        o3 = torch.empty(nq, nd, device="cuda")
        rc = L.maxsim_score_dense(Q.data_ptr(), D.data_ptr(), None if qa is None else qa.data_ptr(), None if da is None else da.data_ptr(),
                                  nq, nd, Lq, Ld, h, _DT[dt], 0 if qa is None else _MDT[torch.float32], o3.data_ptr(),
                                  torch.cuda.current_stream().cuda_stream)
        assert rc == 0, (case, rc)
        torch.cuda.synchronize()
        (o1, a1), (o2, a2) = res
        tag = (case, nq, nd, Lq, Ld, h, str(dt), mode)
        assert torch.equal(o1, o3), tag
        assert bool(torch.isfinite(o1).all()), tag
        torch.testing.assert_close(o1, o2, rtol=0, atol=ATOL16, msg=str(tag))
        assert bool((a1 >= 0).all()) and bool((a1 < Ld).all()), tag
        agree = float((a1 == a2).float().mean())
        assert agree > 0.97 or Ld == 1, (tag, agree)       # (the kernels may break near-ties of rounded sums differently)
    assert served == 30


@pytest.mark.gpu
def test_allpairs_kernel_addresses_past_2_gib(ca):
    """The GEMM-blocked kernel addresses D with 32-bit byte offsets from the tensor base (a buffer descriptor): a D of
    2.2 GiB puts the last docs past 2^31.  First, middle and last docs against float32 torch on the same rounded inputs;
    the last doc also exercises the reads past the end of the tensor (its tile's padding rows)."""
    from colbert_amd.scoring import _DT, _MDT
    L = ca._lib.lib
    nq, nd, Lq, Ld, h, dt = 8, 3900, 32, 380, 768, torch.bfloat16          # 3900 x 380 x 1536 B = 2.12 GiB
    assert nd * Ld * h * 2 > 2 ** 31
    g = torch.Generator(device="cuda").manual_seed(11)
    Q = torch.nn.functional.normalize(torch.randn(nq, Lq, h, generator=g, device="cuda"), dim=-1).to(dt)
    D = torch.empty(nd, Ld, h, dtype=dt, device="cuda")
    for i in range(0, nd, 500):                                              # (float32 staging in pieces: 0.6 GB at a time)
        D[i:i + 500] = torch.nn.functional.normalize(torch.randn(min(500, nd - i), Ld, h, generator=g, device="cuda"), dim=-1).to(dt)
    assert L.maxsim_score_dense_kernel(nq, nd, Lq, Ld, h, _DT[dt], 0) == 1
    out = torch.empty(nq, nd, device="cuda")
    arg = torch.full((nq, nd, Lq), -7, dtype=torch.int32, device="cuda")
    rc = L.maxsim_score_dense_fwd(Q.data_ptr(), D.data_ptr(), None, None, nq, nd, Lq, Ld, h, _DT[dt], 0,
                                  out.data_ptr(), arg.data_ptr(), torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    torch.cuda.synchronize()
    docs = torch.tensor([0, 1, 7, 8, 1949, 2000, 3678, 3679, 3680, 3681, 3891, 3892, 3898, 3899], device="cuda")  # 3680: the first past 2^31
    sim = torch.einsum("qmh,dnh->qdmn", Q.float(), D[docs].float())
    exp = sim.max(-1).values.sum(-1)
    torch.testing.assert_close(out[:, docs], exp, rtol=0, atol=ATOL16)
    top2 = sim.topk(2, dim=-1).values
    clear = top2[..., 0] - top2[..., 1] > 1e-4
    got = arg[:, docs]
    assert bool((got >= 0).all()) and bool((got < Ld).all())
    assert bool((got[clear] == sim.argmax(-1).to(torch.int32)[clear]).all())
    assert bool(torch.isfinite(out).all())


@pytest.mark.gpu
@pytest.mark.parametrize("Ld", [100, 200, 300])
def test_allpairs_kernel_general_float_masks(ca, Ld):
    """Masks the reference never builds but its interface allows: fractional, negative and zero float32 weights on both
    sides (BaseModel.py:41-43 multiplies whatever it is given).  The GEMM-blocked kernel applies a non-negative q_mask
    after the max and takes the multiplication back into every similarity when a weight is negative; both paths against
    float64 on the rounded inputs, arg-max wherever the top two are clearly apart."""
    from colbert_amd.scoring import _DT, _MDT
    L = ca._lib.lib
    gen = torch.Generator().manual_seed(77)
    nq, nd, Lq, h, dt = 24, 64, 32, 128, torch.bfloat16     # 64 x 3 tiles: enough for the GEMM-blocked kernel
    Q = nrm(gen, nq, Lq, h).to(dt)
    D = nrm(gen, nd, Ld, h).to(dt)
    qm = torch.rand(nq, Lq, generator=gen) * 1.5
    qm[torch.rand(nq, Lq, generator=gen) < 0.2] = 0.0
    qm[::3] *= torch.where(torch.rand(nq // 3, Lq, generator=gen) < 0.3, -1.0, 1.0)     # every third query: some negative
    dm = torch.rand(nd, Ld, generator=gen) * 2 - 0.5
    dm[torch.rand(nd, Ld, generator=gen) < 0.2] = 0.0
    out = torch.empty(nq, nd, device="cuda")
    arg = torch.full((nq, nd, Lq), -7, dtype=torch.int32, device="cuda")
    Qd, Dd, qmd, dmd = Q.cuda(), D.cuda(), qm.cuda(), dm.cuda()
    assert L.maxsim_score_dense_kernel(nq, nd, Lq, Ld, h, _DT[dt], _MDT[torch.float32]) == 1
    rc = L.maxsim_score_dense_fwd(Qd.data_ptr(), Dd.data_ptr(), qmd.data_ptr(), dmd.data_ptr(), nq, nd, Lq, Ld, h,
                                  _DT[dt], _MDT[torch.float32], out.data_ptr(), arg.data_ptr(), torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    torch.cuda.synchronize()
    sim = torch.einsum("qmh,dnh->qdmn", Q.double(), D.double()) * qm.double()[:, None, :, None] * dm.double()[None, :, None, :]
    exp = sim.max(-1).values.sum(-1)
    torch.testing.assert_close(out.cpu().double(), exp, rtol=0, atol=2e-3)
    top2 = sim.topk(2, dim=-1).values
    clear = top2[..., 0] - top2[..., 1] > 1e-4
    got = arg.cpu()
    assert bool((got >= 0).all()) and bool((got < Ld).all())
    assert bool((got[clear] == sim.argmax(-1).to(torch.int32)[clear]).all())
    assert float(clear.float().mean()) > 0.5
