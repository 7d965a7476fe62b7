"""GPU parity tests added in round 5: the reference's multi-view deployment shapes in its own storage dtype
(proj_conf/dense.yaml:8,29-32: dim 768, q_view = d_view = 16; BaseModel.py:21-24 keeps the first `view` tokens;
colbert_ranker.py:62 stores fp16) -- the fixed-length 16-bit kernel k_maxsim_stream_uni16 for dim 128 and the LDS-query
kernel for dim 768.  Tolerances as in test_gpu_parity.py: 16-bit inputs |d| <= 1e-3 against the oracle on identically
rounded inputs; kernel forms against each other: bit for bit."""
import copy
import ctypes

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

ATOL16 = 1e-3


@pytest.fixture(scope="module")
def ca():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    import colbert_amd
    return colbert_amd


def nrm(gen, *shape):
    return F.normalize(torch.randn(*shape, generator=gen), dim=-1)


def without_promise(r):
    """The same index with the fixed-length promise withdrawn (uniform_len = 0): the library takes its general kernels."""
    g = copy.copy(r)
    g._iv = r._index_view()
    g._iv.uniform_len = 0
    g._iv_ref = ctypes.byref(g._iv)
    g._iv_addr = ctypes.addressof(g._iv)
    return g


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16], ids=["fp16", "bf16"])
@pytest.mark.parametrize("L,Lq", [(8, 8), (8, 32), (4, 7), (4, 20), (16, 16), (16, 40), (8, 1)])
def test_uniform_short_docs_16bit_kernel_is_bit_identical(ca, dtype, L, Lq):
    """A 16-bit index whose every doc has exactly L tokens runs k_maxsim_stream_uni16 (doc length compiled in); its scores
    equal, bit for bit, those of the general packed-tile kernel on the same index (the promise withdrawn) and agree with the
    oracle's closed form on the identically rounded values: padding slots, out-of-range pids, row widths that leave the last
    tile / wave / workgroup partly filled, q_mask, q_len, fp32 and 16-bit queries, queries longer than 32 tokens."""
    from oracle.maxsim_oracle import ragged_scores_f64
    gen = torch.Generator().manual_seed(500 + L + Lq)
    ndocs, h = 5000, 128
    doclens = [L] * ndocs
    emb = nrm(gen, sum(doclens), h).to(dtype)
    r = ca.ColbertRanker(parts=[emb], parts_doclens=[doclens], dim=h, index_dtype=dtype)
    assert r._iv.uniform_len == L
    g = without_promise(r)
    for nq, ncand in ((1, 1), (3, 7), (2, 65), (5, 1000), (300, 130), (2, 2049), (70, 513)):
        Q = nrm(gen, nq, Lq, h)
        cand = torch.randint(0, ndocs, (nq, ncand), generator=gen)
        if ncand > 3:
            cand[0, 1] = -1                              # padding slot
            cand[-1, ncand - 2] = ndocs + 3              # out of range
        qm = (torch.rand(nq, Lq, generator=gen) > 0.25).long()
        qm[:, 0] = 1
        ql = torch.randint(1, Lq + 1, (nq,), generator=gen)
        for kw in (dict(), dict(q_mask=qm), dict(q_len=ql)):
            for Qx in (Q, Q.to(dtype)):
                a = r.score_candidates(Qx, cand.cuda(), **kw).cpu()
                b = g.score_candidates(Qx, cand.cuda(), **kw).cpu()
                assert torch.equal(a, b), (L, Lq, nq, ncand, list(kw), Qx.dtype)
        a = r.score_candidates(Q, cand.cuda(), q_mask=qm).cpu()
        if ncand > 3:
            assert float(a[0, 1]) == float("-inf") and float(a[-1, ncand - 2]) == float("-inf")
        if nq <= 5 and ncand <= 1000:
            for qi in range(nq):
                ok = [(c, p) for c, p in enumerate(cand[qi].tolist()) if 0 <= p < ndocs][:40]
                exp = ragged_scores_f64(emb.float(), r.doclens, r.doclens_pfxsum, r.d_pad_len.cpu(), Q[qi][qm[qi].bool()], [p for _, p in ok])
                np.testing.assert_allclose(a[qi, [c for c, _ in ok]].numpy(), exp, rtol=0, atol=ATOL16)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16], ids=["fp16", "bf16"])
@pytest.mark.parametrize("L,Lq,nq,ncand", [(8, 8, 40, 500), (16, 40, 9, 130), (4, 20, 12, 300), (8, 32, 3, 2100)])
def test_uniform_16bit_counted_rows_equal_full_width_rows(ca, dtype, L, Lq, nq, ncand):
    """Counted rows (a doc shard's share / ANN pid lists) of a uniform 16-bit index walk the device-built work list with the
    LIST form of k_maxsim_stream_uni16: same bits as the static grid on the same rows, and as the general kernels."""
    gen = torch.Generator().manual_seed(77 + L)
    ndocs, h = 3000, 128
    emb = nrm(gen, ndocs * L, h).to(dtype)
    r = ca.ColbertRanker(parts=[emb], parts_doclens=[[L] * ndocs], dim=h, index_dtype=dtype)
    g = without_promise(r)
    counts = torch.randint(0, ncand + 1, (nq,), generator=gen)
    counts[0], counts[-1] = 0, ncand
    cand = torch.full((nq, ncand), -1, dtype=torch.int64)
    for q in range(nq):
        cand[q, :int(counts[q])] = torch.randint(0, ndocs, (int(counts[q]),), generator=gen)
    Q = nrm(gen, nq, Lq, h)
    cc = counts.to(torch.int32).cuda()
    full = r.score_candidates(Q, cand.cuda()).cpu()
    counted = r.score_candidates(Q, cand.cuda(), cand_count=cc).cpu()
    general = g.score_candidates(Q, cand.cuda(), cand_count=cc).cpu()
    assert torch.equal(full, counted) and torch.equal(full, general)
    live = cand >= 0
    assert torch.isinf(full[~live]).all() and torch.isfinite(full[live]).all()
    # counted top-k over the same rows
    k = 10
    p1, s1 = r.topk(full.cuda(), cand.cuda(), k, cc)
    exp_s = torch.sort(full, dim=1, descending=True, stable=True).values[:, :k]
    assert torch.equal(s1.cpu(), exp_s)


def test_multiview_goldens_from_the_imported_reference(ca, golden):
    """`mv768_fp16.npz` was written by tests/golden/make_golden.py from the imported BaseModel.score on the reference's
    default multi-view shape (Q 2 x 16 x 768, D 32 x 16 x 768, fp16-rounded values scored in fp32; dense.yaml:8,29-32),
    `mv128_fp16.npz` on BASELINE configs[3]'s shape in the reference's storage dtype (Q 2 x 8 x 128, D 64 x 8 x 128): the
    fused rerank on an fp16 index of those docs must reproduce the [q, d] matrix (|d| <= 1e-3), through the static grid,
    counted rows and the operator seam."""
    for name, L, h in (("mv768_fp16", 16, 768), ("mv128_fp16", 8, 128)):
        gd = golden(name)
        assert gd["D"].dtype == torch.float16                            # stored in the index's own dtype
        Q, D, exp = gd["Q"], gd["D"].float(), gd["expected"]
        nq, nd = Q.size(0), D.size(0)
        assert tuple(D.shape[1:]) == (L, h) and Q.size(1) == L
        idx = gd["D"].reshape(nd * L, h)
        r = ca.ColbertRanker(parts=[idx], parts_doclens=[[L] * nd], dim=h, index_dtype=torch.float16)
        assert r._iv.uniform_len == L
        cand = torch.arange(nd).repeat(nq, 1)
        got = r.score_candidates(Q, cand.cuda()).cpu()
        np.testing.assert_allclose(got.numpy(), exp.numpy(), rtol=0, atol=ATOL16)
        cc = torch.full((nq,), nd, dtype=torch.int32).cuda()
        assert torch.equal(r.score_candidates(Q, cand.cuda(), cand_count=cc).cpu(), got)
        assert torch.equal(without_promise(r).score_candidates(Q, cand.cuda()).cpu(), got)
        # the operator seam on the same tensors (fp32 operands holding fp16 values, ones masks as colbert_ranker.py:108-112)
        ones_q, ones_d = torch.ones(nq, L, dtype=torch.long), torch.ones(nd, L, dtype=torch.long)
        dense = ca.score(Q.cuda(), D.cuda(), ones_q.cuda(), ones_d.cuda()).cpu()
        np.testing.assert_allclose(dense.numpy(), exp.numpy(), rtol=0, atol=1e-4)
        # rank_forward (the reference's call: Q [1, h, Lq], python list of pids): the oracle's order on the golden's scores
        for qi in range(nq):
            pids, scores = r.rank_forward(Q[qi:qi + 1].permute(0, 2, 1), list(range(nd)), depth=10)
            order = torch.sort(exp[qi], descending=True, stable=True).indices[:10].tolist()
            assert pids == order or np.allclose(scores, exp[qi][order].numpy(), atol=ATOL16)
            np.testing.assert_allclose(scores, exp[qi][order].numpy(), rtol=0, atol=ATOL16)


@pytest.mark.parametrize("name,L,lq,h,ndocs", [("mv128", 8, 8, 128, 1_000_000), ("mv768", 16, 16, 768, 200_000)])
def test_multiview_full_batch_properties(ca, name, L, lq, h, ndocs):
    """BASELINE-sized launches (256 queries x 1000 candidates) of the two multi-view shapes on an fp16 index, through
    size-independent properties: permuting a query's candidate list permutes its scores (bitwise); a candidate's score does not
    depend on its neighbours in the list nor on the batch (one query alone == the same row of the batch, bitwise); doubling
    an fp16 query doubles the score (exact in floating point); a sample agrees with fp32 torch on the stored values."""
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev).manual_seed(9)
    idx = torch.empty(ndocs * L, h, dtype=torch.float16, device=dev)
    step = 1 << 19
    for a in range(0, idx.size(0), step):
        b = min(a + step, idx.size(0))
        idx[a:b] = F.normalize(torch.randn(b - a, h, generator=gen, device=dev), dim=-1).half()
    r = ca.ColbertRanker.from_device_tensor(idx, [L] * ndocs)
    assert r._iv.uniform_len == L
    nq, ncand = 256, 1000
    Q = F.normalize(torch.randn(nq, lq, h, generator=gen, device=dev), dim=-1)
    cand = torch.stack([torch.randperm(ndocs, generator=gen, device=dev)[:ncand] for _ in range(nq)])
    s = r.score_candidates(Q, cand)
    assert torch.isfinite(s).all()
    perm = torch.randperm(ncand, generator=gen, device=dev)
    assert torch.equal(r.score_candidates(Q, cand[:, perm]), s[:, perm])
    assert torch.equal(r.score_candidates(Q[17:18], cand[17:18, :333]), s[17:18, :333])
    Qh = Q.half()                                         # a 16-bit query has no low piece: doubling it is exact end to end
    assert torch.equal(r.score_candidates(2.0 * Qh, cand), 2.0 * r.score_candidates(Qh, cand))
    qs, cs = Q[:3], cand[:3, :50]
    rows = (cs.unsqueeze(-1) * L + torch.arange(L, device=dev)).view(3, -1)
    D = idx[rows].float().view(3, 50, L, h)
    exp = torch.einsum("qmh,qdnh->qdmn", qs, D).max(-1).values.sum(-1)
    np.testing.assert_allclose(s[:3, :50].cpu().numpy(), exp.cpu().numpy(), rtol=0, atol=ATOL16)
    # top-100 of the full batch == stable sort of the score matrix
    top_p, top_s = r.topk(s, cand, 100)
    exp_s, exp_i = torch.sort(s, dim=1, descending=True, stable=True)
    assert torch.equal(top_s, exp_s[:, :100]) and torch.equal(top_p, cand.gather(1, exp_i[:, :100]))


# ------------------------------------------------------------------------------------------------------
# ids -> distinct pids in every regime of the row-block table (written for round 5's block-level de-duplication experiment,
# docs/experiments.md "ids -> distinct pids: block set"; the regimes stay as coverage of k_unique_pids)
# ------------------------------------------------------------------------------------------------------
def _emb2pid(doclens):
    return torch.repeat_interleave(torch.arange(len(doclens)), torch.tensor(doclens))     # colbert_ranker.py:163-174


def _expect(e, e2p, keep=None, base=0):
    """sorted(set(emb2pid[ids])) per query (colbert_ranker.py:212-229, :234) over the live ids."""
    nq = e.size(0)
    flat = e.reshape(nq, -1)
    out = []
    for q in range(nq):
        ids = flat[q]
        if keep is not None:
            ids = e[q][keep[q].bool()].flatten()
        ids = ids - base
        ids = ids[(ids >= 0) & (ids < e2p.numel())]
        out.append(sorted(set(e2p[ids].tolist())))
    return out


def _check(cand, cnt, exp):
    cand, cnt = cand.cpu(), cnt.cpu()
    for q, want in enumerate(exp):
        assert int(cnt[q]) == len(want), (q, int(cnt[q]), len(want))
        assert cand[q, :len(want)].tolist() == want, q
        assert bool((cand[q, len(want):] == -1).all()), q


@pytest.mark.parametrize("regime", ["long_docs", "two_doc_boundaries", "short_docs_with_empties", "one_token_docs", "huge_docs"])
@pytest.mark.parametrize("n", [16384, 5000, 300])
def test_ids_to_pids_row_block_regimes(ca, regime, n):
    """Bit-exact against sorted(set(emb2pid[ids])) in every block regime of the row-block table: blocks inside one doc, blocks
    with one doc boundary (ids on both sides of it, on one side only, exactly at it), blocks with many docs and empty docs in
    between (a search between two table entries), one-token docs (64 docs per block); ids clustered on a few hot docs (many
    ids per block and per doc), ids spread over up to n distinct blocks, ids on both sides of doc boundaries, FAISS's -1
    padding, rows outside the index, a shard's id_base and the keep-mask."""
    g = torch.Generator().manual_seed(len(regime) * 1000 + n)
    if regime == "long_docs":
        doclens = torch.randint(100, 400, (6000,), generator=g).tolist()
    elif regime == "two_doc_boundaries":
        doclens = torch.randint(60, 70, (20000,), generator=g).tolist()                  # nearly every block holds one boundary
    elif regime == "short_docs_with_empties":
        doclens = torch.randint(0, 12, (150000,), generator=g).tolist()
        doclens[1000:1400] = [0] * 400                                                   # a long run of empty docs
    elif regime == "one_token_docs":
        doclens = [1] * 700000
    else:
        doclens = [5000, 1, 70000, 0, 0, 3, 64, 64, 128, 1000000]                        # blocks of one doc for thousands of blocks
    doclens[0] = max(doclens[0], 1)
    emb = torch.zeros(sum(doclens), 16, dtype=torch.float16)                             # (the rows' values do not matter here)
    r = ca.ColbertRanker(parts=[emb], parts_doclens=[doclens], dim=16, index_dtype=torch.float16)
    e2p = _emb2pid(doclens)
    ntok = e2p.numel()
    offs = torch.tensor([0] + doclens).cumsum(0)
    nq = 5
    e = torch.empty(nq, n, dtype=torch.int64)
    e[0] = torch.randint(0, ntok, (n,), generator=g)                                     # spread: up to n distinct blocks
    hot = torch.randint(0, len(doclens), (40,), generator=g)                             # clustered on 40 docs
    hot = hot[torch.tensor(doclens)[hot] > 0]
    pick = hot[torch.randint(0, hot.numel(), (n,), generator=g)]
    e[1] = offs[pick] + (torch.rand(n, generator=g) * torch.tensor(doclens)[pick]).long()
    starts = offs[:-1][torch.tensor(doclens) > 0]                                        # first / last rows of docs: the boundaries
    sel = starts[torch.randint(0, starts.numel(), (n,), generator=g)]
    e[2] = sel - (torch.arange(n) % 2)                                                   # a doc's first row, or the row before it
    e[2].clamp_(0, ntok - 1)
    e[3] = torch.arange(n) * max(1, ntok // n) % ntok                                    # an arithmetic walk: distinct blocks when ntok is large
    e[4] = e[0]
    e[4, ::3] = -1                                                                       # FAISS "no neighbour"
    e[4, 1::7] = ntok + 17                                                               # outside the index
    cand, cnt = r.embedding_ids_to_pids(e.cuda(), trim=False)
    _check(cand, cnt, _expect(e, e2p))
    # a shard's view: ids shifted by id_base, foreign rows dropped; a keep-mask over 4 "query tokens"
    if n % 4 == 0:
        keep = torch.tensor([[1, 0, 1, 1]] * nq)
        keep[3] = 0
        base = 777
        eb = (e + base).view(nq, 4, n // 4)
        cand, cnt = r.embedding_ids_to_pids(eb.cuda(), trim=False, keep=keep, id_base=base)
        _check(cand, cnt, _expect(eb, e2p, keep, base))
        assert int(cnt[3]) == 0


def test_sharded_output_D_embedding_on_gpu(ca, tmp_path):
    """ShardedRanker.rank_forward(output_D_embedding=True) (colbert_ranker.py:131-136) on the HIP path: a one-shard "sharded"
    index returns what the single-GPU ColbertRanker returns (pids, D [k, S, h] fp32 incl. the aliased slots past a doc's end
    and the zero tail, mask); the multi-rank exchange of the rows is covered on CPU ranks (tests/test_sharded_files.py)."""
    import os
    from colbert_amd.index_io import save_index
    from colbert_amd.sharded import load_shard
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ragged_rerank_64.npz"))
    save_index(str(tmp_path), [torch.from_numpy(z["part0"]), torch.from_numpy(z["part1"])], [z["doclens0"].tolist(), z["doclens1"].tolist()])
    whole = ca.ColbertRanker(index_path=str(tmp_path), device="cuda:0")
    sh = load_shard(str(tmp_path), 0, 1, device="cuda:0")
    Q = torch.from_numpy(z["Q"]).cuda()
    pad = z["pad_len"].tolist()
    for bucket in sorted(set(pad)):
        one = [p for p in z["pids"].tolist() if pad[p] == bucket]
        gp, gD, gm = sh.rank_forward(Q, one, depth=6, output_D_embedding=True)
        ep, eD, em = whole.rank_forward(Q, one, depth=6, output_D_embedding=True)
        assert gp == ep and torch.equal(gD, eD) and torch.equal(gm, em)
    with pytest.raises(RuntimeError, match="Sizes of tensors must match"):
        sh.rank_forward(Q, z["pids"].tolist(), depth=6, output_D_embedding=True)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16], ids=["fp16", "bf16"])
@pytest.mark.parametrize("L,Lq", [(16, 16), (16, 8), (8, 13), (16, 17), (4, 32)])
def test_multiview_dim768_launch_forms_are_bit_identical(ca, dtype, L, Lq):
    """Dim 768 (the reference's default, dense.yaml:8) with uniform short docs: the static grid (<= 16 query tokens: the 16-row
    query image + 12 waves per workgroup), counted rows (the work-list form), one query per launch, the promise withdrawn,
    fp32 and 16-bit queries, q_mask / q_len -- all return the same bits, and agree with the oracle's closed form."""
    from oracle.maxsim_oracle import ragged_scores_f64
    gen = torch.Generator().manual_seed(900 + L + Lq)
    ndocs, h = 700, 768
    emb = nrm(gen, ndocs * L, h).to(dtype)
    r = ca.ColbertRanker(parts=[emb], parts_doclens=[[L] * ndocs], dim=h, index_dtype=dtype)
    g = without_promise(r)
    for nq, ncand in ((1, 1), (3, 40), (2, 1000), (40, 130)):
        Q = nrm(gen, nq, Lq, h)
        cand = torch.randint(0, ndocs, (nq, ncand), generator=gen)
        if ncand > 3:
            cand[0, 1] = -1
            cand[-1, ncand - 2] = ndocs + 3
        qm = (torch.rand(nq, Lq, generator=gen) > 0.25).long()
        qm[:, 0] = 1
        ql = torch.randint(1, Lq + 1, (nq,), generator=gen)
        counts = torch.randint(0, ncand + 1, (nq,), generator=gen)
        counts[0] = ncand
        cnt_rows = torch.full((nq, ncand), -1, dtype=torch.int64)
        for q in range(nq):
            cnt_rows[q, :int(counts[q])] = cand[q, :int(counts[q])].clamp(0, ndocs - 1)
        for kw in (dict(), dict(q_mask=qm), dict(q_len=ql)):
            for Qx in (Q, Q.to(dtype)):
                a = r.score_candidates(Qx, cand.cuda(), **kw).cpu()
                assert torch.equal(a, g.score_candidates(Qx, cand.cuda(), **kw).cpu()), (L, Lq, nq, ncand, list(kw), Qx.dtype)
                for q in (0, nq - 1):                                  # one query per launch
                    one_kw = {k: v[q:q + 1] for k, v in kw.items()}
                    assert torch.equal(r.score_candidates(Qx[q:q + 1], cand[q:q + 1].cuda(), **one_kw).cpu()[0], a[q])
                full = r.score_candidates(Qx, cnt_rows.cuda(), **kw).cpu()
                counted = r.score_candidates(Qx, cnt_rows.cuda(), cand_count=counts.int().cuda(), **kw).cpu()
                assert torch.equal(full, counted), (L, Lq, nq, ncand, list(kw), Qx.dtype)
        a = r.score_candidates(Q, cand.cuda(), q_mask=qm).cpu()
        if nq <= 3:
            for qi in range(nq):
                ok = [(c, p) for c, p in enumerate(cand[qi].tolist()) if 0 <= p < ndocs][:25]
                exp = ragged_scores_f64(emb.float(), r.doclens, r.doclens_pfxsum, r.d_pad_len.cpu(), Q[qi][qm[qi].bool()], [p for _, p in ok])
                np.testing.assert_allclose(a[qi, [c for c, _ in ok]].numpy(), exp, rtol=0, atol=ATOL16)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16], ids=["fp16", "bf16"])
@pytest.mark.parametrize("Lq", [5, 16, 17])
def test_short_queries_on_long_ragged_docs_at_dim768(ca, dtype, Lq):
    """<= 16 query tokens at dim 768 on LONG ragged docs: the 16-row query image with the 8-wave rings (not the 12-wave shape of
    the multi-view case), fp32 and 16-bit queries (two- and one-piece image), against the float64 closed form; static grid ==
    counted rows == one query per launch; Lq = 17 is the 32-row image next to it."""
    from oracle.maxsim_oracle import ragged_scores_f64
    gen = torch.Generator().manual_seed(1700 + Lq)
    ndocs, h = 300, 768
    doclens = torch.randint(70, 330, (ndocs,), generator=gen).tolist()
    doclens[5], doclens[6] = 1, 511
    emb = nrm(gen, sum(doclens), h).to(dtype)
    r = ca.ColbertRanker(parts=[emb], parts_doclens=[doclens], dim=h, index_dtype=dtype)
    for nq, ncand in ((1, 7), (3, 60), (33, 140)):
        Q = nrm(gen, nq, Lq, h)
        cand = torch.randint(0, ndocs, (nq, ncand), generator=gen)
        cand[0, :2] = torch.tensor([5, 6])
        counts = torch.randint(1, ncand + 1, (nq,), generator=gen)
        counts[0] = ncand
        rows = torch.full((nq, ncand), -1, dtype=torch.int64)
        for q in range(nq):
            rows[q, :int(counts[q])] = cand[q, :int(counts[q])]
        for Qx in (Q, Q.to(dtype)):
            a = r.score_candidates(Qx, cand.cuda()).cpu()
            assert torch.equal(r.score_candidates(Qx[:1], cand[:1].cuda()).cpu()[0], a[0])
            full = r.score_candidates(Qx, rows.cuda()).cpu()
            counted = r.score_candidates(Qx, rows.cuda(), cand_count=counts.int().cuda()).cpu()
            assert torch.equal(full, counted), (Lq, nq, ncand, Qx.dtype)
            live = rows >= 0
            assert torch.equal(full[live], a[live])
            for qi in range(min(nq, 2)):
                pids = cand[qi].tolist()[:30]
                exp = ragged_scores_f64(emb.float(), r.doclens, r.doclens_pfxsum, r.d_pad_len.cpu(), Qx[qi].float(), pids)
                np.testing.assert_allclose(a[qi, :len(pids)].numpy(), exp, rtol=0, atol=ATOL16)
