"""GPU parity tests added in round 4: the hash-set form of ids -> distinct pids (maxsim_embedding_ids_to_pids_ex: row-block
table, keep-mask and id_base applied in the kernel, full-sort overflow path), bit-exact against the reference's
emb2pid + set() (colbert_ranker.py:163-174, :212-229, :234)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ca():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    import colbert_amd
    return colbert_amd


def emb2pid_of(doclens):
    """build_emb2pid, colbert_ranker.py:163-174."""
    return torch.repeat_interleave(torch.arange(len(doclens)), torch.as_tensor(doclens))


def expect_rows(e, emb2pid, keep=None, id_base=0):
    """Per query: sorted(set(emb2pid[ids])) of the ids that count (>= 0 after id_base, inside the index, token kept)."""
    nq = e.size(0)
    flat = e.reshape(nq, -1)
    out = []
    for q in range(nq):
        ids = flat[q]
        if keep is not None:
            per = ids.numel() // keep.size(1)
            ids = ids[keep[q].bool().repeat_interleave(per)]
        ids = ids - id_base
        ids = ids[(ids >= 0) & (ids < emb2pid.numel())]
        out.append(sorted(set(emb2pid[ids].tolist())))
    return out


def check(cand, cnt, exp):
    cand, cnt = cand.cpu(), cnt.cpu()
    for q, want in enumerate(exp):
        c = int(cnt[q])
        assert c == len(want), (q, c, len(want))
        assert cand[q, :c].tolist() == want, q
        assert bool((cand[q, c:] == -1).all()), q


def make_ranker(ca, doclens):
    return ca.ColbertRanker(parts=[torch.zeros(sum(doclens), 8)], parts_doclens=[doclens], dim=8)


def test_row_block_table(ca):
    """maxsim_build_row_blocks: entry b = {doc of token row 64 b (empty docs skipped), offset of the next doc inside the block
    or 0, "search" flag}; last entry = n_docs - 1.  Decoded here against build_emb2pid for EVERY row."""
    g = torch.Generator().manual_seed(1)
    doclens = torch.randint(0, 300, (3000,), generator=g).tolist()
    doclens[100:400] = [0] * 300                       # a long run of empty docs
    doclens[500:540] = [3] * 40                        # many docs inside one block
    doclens[-1] = 5
    r = make_ranker(ca, doclens)
    e2p = emb2pid_of(doclens)
    ntok = e2p.numel()
    nblocks = (ntok + 63) // 64
    tbl = r.d_row_blocks.view(torch.int64)[:nblocks + 1].cpu()
    d0 = tbl & 0xFFFFFFFF
    bnd = (tbl >> 32) & 127
    multi = (tbl >> 39) & 1
    assert d0[:nblocks].tolist() == e2p[torch.arange(nblocks) * 64].tolist()
    assert int(d0[nblocks]) == len(doclens) - 1
    rows = torch.arange(ntok)
    b = rows >> 6
    simple = multi[b] == 0
    got = d0[b] + ((bnd[b] != 0) & ((rows & 63) >= bnd[b])).long()
    assert torch.equal(got[simple], e2p[simple])       # one 8-byte entry answers these rows
    assert int(simple.sum()) > ntok // 2 and int((~simple).sum()) > 0
    # flagged blocks: the doc lies between this entry's and the next one's
    nb = b[~simple]
    assert bool(((e2p[~simple] >= d0[nb]) & (e2p[~simple] <= d0[nb + 1])).all())


@pytest.mark.parametrize("ndocs,lo,hi,n,distinct", [
    (20000, 1, 3, 16384, None),        # ~11000 distinct docs in the 16384-slot set (load factor 0.68)
    (200000, 1, 3, 16384, None),       # ~15700 distinct docs: probe chains pass 64 -> the full-sort path
    (20000, 1, 3, 16384, 6000),
    (20000, 1, 3, 16384, 8192),        # sorted as exactly 8192 keys
    (20000, 1, 3, 16384, 8193),        # ... and as 16384
    (20000, 1, 3, 16384, 13000),       # load factor 0.79: long probe chains, either path
    (20000, 1, 3, 16384, 15000),
    (20000, 1, 3, 16384, 16384),       # every id another doc: as many distinct docs as slots
    (20000, 1, 3, 9000, None),
    (3000, 0, 400, 16384, None),       # long docs (several 64-row blocks each) and empty docs
    (3000, 0, 400, 4096, None),        # n <= 4096: 8192-slot set
    (3000, 0, 400, 2048, None),        # 4096-slot set
    (3000, 0, 400, 2048, 2048),        # ... full to load factor 0.5
    (3000, 0, 400, 700, None),
    (5, 1, 2, 16384, None),            # a handful of docs
])
def test_ids_to_pids_hash_and_overflow_paths(ca, ndocs, lo, hi, n, distinct):
    g = torch.Generator().manual_seed(ndocs + n + (distinct or 0))
    doclens = torch.randint(lo, hi + 1, (ndocs,), generator=g).tolist()
    if lo == 0:
        doclens[10:150] = [0] * 140
    doclens[0] = max(doclens[0], 1)
    r = make_ranker(ca, doclens)
    e2p = emb2pid_of(doclens)
    ntok = e2p.numel()
    nq = 4
    if distinct is None:
        e = torch.randint(0, ntok, (nq, n), generator=g)
    else:
        offs = torch.tensor([0] + doclens).cumsum(0)
        e = torch.empty(nq, n, dtype=torch.int64)
        for q in range(nq):
            live = torch.tensor([i for i, dl in enumerate(doclens) if dl > 0])
            docs = live[torch.randperm(live.numel(), generator=g)[:distinct - (q & 1)]]    # odd rows: one doc fewer
            pick = torch.cat([docs, docs[torch.randint(0, docs.numel(), (n - docs.numel(),), generator=g)]])
            e[q] = offs[pick][torch.randperm(n, generator=g)]
    e[1, ::97] = -1                                                                    # FAISS "no neighbour"
    e[2, 3] = ntok + 5                                                                 # outside the index: dropped
    e[3] = e[3, 0]                                                                     # one doc, n times (a mixed launch)
    cand, cnt = r.embedding_ids_to_pids(e, trim=False)
    check(cand, cnt, expect_rows(e, e2p))
    if distinct is not None:
        assert int(cnt[0]) == distinct


def test_ids_to_pids_keep_mask_and_id_base(ca):
    """The driver's two elementwise passes folded into the kernel: neighbours of dropped query tokens are ignored, ids are
    shifted by a shard's first token row and foreign rows dropped -- same result as doing both on the host first."""
    g = torch.Generator().manual_seed(4)
    doclens = torch.randint(1, 200, (4000,), generator=g).tolist()
    r = make_ranker(ca, doclens)
    e2p = emb2pid_of(doclens)
    ntok = e2p.numel()
    nq, Lq, depth = 6, 32, 128
    keep = (torch.rand(nq, Lq, generator=g) > 0.4).long()
    keep[0] = 0                                   # nothing kept: an empty row
    keep[1] = 1
    base = 12345
    # global rows of an index in which this one starts at row `base` and is followed by more rows
    e = torch.randint(0, ntok + 3 * base, (nq, Lq, depth), generator=g)
    e[2, :, :5] = -1
    for kw, exp in ((dict(keep=keep), expect_rows(e, e2p, keep)),
                    (dict(id_base=base), expect_rows(e, e2p, None, base)),
                    (dict(keep=keep, id_base=base), expect_rows(e, e2p, keep, base))):
        cand, cnt = r.embedding_ids_to_pids(e.cuda(), trim=False, **kw)
        assert cand.shape == (nq, Lq * depth)
        check(cand, cnt, exp)
    assert int(cnt[0]) == 0
    # the 2-D form (colbert_ranker.py:178's reshape) gives the same rows
    c2, n2 = r.embedding_ids_to_pids(e.reshape(nq, -1).cuda(), trim=False, keep=keep, id_base=base)
    assert torch.equal(c2, cand) and torch.equal(n2, cnt)


def test_ids_to_pids_legacy_entry_point_without_row_blocks(ca):
    """maxsim_embedding_ids_to_pids (no table: binary search over the whole prefix sum) == the _ex form."""
    from colbert_amd import _lib
    g = torch.Generator().manual_seed(9)
    doclens = torch.randint(0, 50, (7000,), generator=g).tolist()
    r = make_ranker(ca, doclens)
    e2p = emb2pid_of(doclens)
    for n in (16384, 3000):
        e = torch.randint(-3, e2p.numel(), (3, n), generator=g).cuda()
        out = torch.empty_like(e)
        cnt = torch.empty(3, dtype=torch.int32, device="cuda")
        rc = _lib.lib.maxsim_embedding_ids_to_pids(e.data_ptr(), 3, n, r.d_offsets.data_ptr(), r.n_docs, r.num_embeddings,
                                                   out.data_ptr(), cnt.data_ptr(), torch.cuda.current_stream().cuda_stream)
        assert rc == 0
        check(out, cnt, expect_rows(e.cpu(), e2p))
        c2, n2 = r.embedding_ids_to_pids(e, trim=False)
        assert torch.equal(c2, out) and torch.equal(n2, cnt)
    # argument checks of the _ex form
    st = torch.cuda.current_stream().cuda_stream
    km = torch.ones(3, 7, dtype=torch.uint8, device="cuda")
    bad = _lib.lib.maxsim_embedding_ids_to_pids_ex(e.data_ptr(), 3, 3000, 7, km.data_ptr(), 0, r.d_offsets.data_ptr(), r.n_docs,
                                                   r.num_embeddings, None, out.data_ptr(), cnt.data_ptr(), st)
    assert bad == _lib.EINVAL                      # 3000 is not a multiple of ids_per_token = 7
    assert _lib.lib.maxsim_embedding_ids_to_pids_ex(e.data_ptr(), 3, 20000, 1, None, 0, r.d_offsets.data_ptr(), r.n_docs,
                                                   r.num_embeddings, None, out.data_ptr(), cnt.data_ptr(), st) == _lib.ERANGE


def test_retrieve_batch_leaves_the_callers_ids_alone(ca):
    """retrieve_batch with masked query tokens: same lists as masking the ids on the host first; the ids tensor the caller
    passed is not written."""
    g = torch.Generator().manual_seed(12)
    doclens = torch.randint(1, 90, (2000,), generator=g).tolist()
    parts = [F.normalize(torch.randn(sum(doclens), 128, generator=g), dim=-1).half()]
    r = ca.ColbertRanker(parts=parts, parts_doclens=[doclens], dim=128)
    nq, Lq, depth = 7, 32, 64
    Q = F.normalize(torch.randn(nq, Lq, 128, generator=g), dim=-1).cuda()
    keep = (torch.rand(nq, Lq, generator=g) > 0.3).long().cuda()
    ids = torch.randint(0, sum(doclens), (nq, Lq, depth), generator=g).cuda()
    before = ids.clone()
    got = ca.retrieve_batch(r, Q, keep, topk=50, embedding_ids=ids)
    assert torch.equal(ids, before)
    masked = ids.masked_fill(keep.unsqueeze(-1) == 0, -1)
    exp = ca.retrieve_batch(r, Q, torch.ones_like(keep), topk=50, embedding_ids=masked)
    # (the second call scores with ALL query tokens: only the candidate SETS are comparable) ...
    cand_a, cnt_a = r.embedding_ids_to_pids(ids, trim=False, keep=keep)
    cand_b, cnt_b = r.embedding_ids_to_pids(masked, trim=False)
    assert torch.equal(cand_a, cand_b) and torch.equal(cnt_a, cnt_b)
    # ... and the lists equal a rerank of exactly those candidates with the keep-mask
    tp, ts = r.rerank_batch(Q, cand_a, depth=50, q_mask=keep, cand_count=cnt_a)
    for q, (p, s) in enumerate(got):
        n = min(50, int(cnt_a[q]))
        assert p == tp[q, :n].tolist() and s == ts[q, :n].tolist()
    assert len(exp) == nq


@pytest.mark.parametrize("dtype,atol,nq", [(torch.float32, 2e-4, 40), (torch.bfloat16, 3e-2, 40),
                                           (torch.float32, 2e-3, 530)])     # 16960 items per doc: batches beyond the register-held ones
def test_backward_with_skewed_argmax_buckets(ca, dtype, atol, nq):
    """The dD pass counting-sorts a doc's (q, m) items by arg-max token (maxsim_backward.h, k_maxsim_bwd_index).  Docs whose
    arg-maxes all fall on ONE token (a doc with a single live token; a doc with one dominant token), empty buckets, masked
    query tokens and a doc with no contribution at all, against torch autograd through the four ops of BaseModel.py:41-45;
    and bitwise reproducible from run to run (the sort is stable: fixed summation order)."""
    g = torch.Generator().manual_seed(41)
    nd, lq, ld, h = 9, 32, 96, 64
    Q = F.normalize(torch.randn(nq, lq, h, generator=g), dim=-1)
    D = F.normalize(torch.randn(nd, ld, h, generator=g), dim=-1) * 0.3
    D[1, 5] = F.normalize(Q.mean((0, 1)), dim=-1) * 3.0          # doc 1: token 5 wins nearly every (q, m)
    qm = (torch.rand(nq, lq, generator=g) > 0.2).long()
    dm = torch.ones(nd, ld, dtype=torch.long)
    dm[0, 1:] = 0                                                # doc 0: one live token -> one bucket of nq * Lq items
    dm[2, 50:] = 0
    w = torch.randn(nq, nd, generator=g)
    w[:, 3] = 0.0                                                # doc 3 receives no gradient: every item dropped

    def run(fn, Q0, D0):
        q = Q0.clone().cuda().requires_grad_(True)
        dd = D0.clone().cuda().requires_grad_(True)
        out = fn(q, dd, qm.cuda(), dm.cuda())
        (out.float() * w.cuda()).sum().backward()
        return out.detach().float().cpu(), q.grad.float().cpu(), dd.grad.float().cpu()

    def torch_score(q, dd, qmask, dmask):
        return torch.einsum("qmh,dnh->qdmn", q * qmask[..., None], dd * dmask[..., None]).max(-1).values.sum(-1)

    Qt, Dt = Q.to(dtype), D.to(dtype)
    o1, gq1, gd1 = run(ca.score, Qt, Dt)
    o2, gq2, gd2 = run(torch_score, Qt.float(), Dt.float())
    assert (o1 - o2).abs().max() <= atol * 10
    assert (gq1 - gq2).abs().max() <= atol and (gd1 - gd2).abs().max() <= atol * 4
    assert float(gd1[3].abs().max()) == 0.0 and float(gd1[0, 1:].abs().max()) == 0.0
    o3, gq3, gd3 = run(ca.score, Qt, Dt)
    assert torch.equal(gd1, gd3) and torch.equal(gq1, gq3)


@pytest.mark.parametrize("h,dtype,lo,hi,gib", [(128, torch.float16, 40, 180, 4.6), (128, torch.float32, 8, 8, 4.4),
                                               (768, torch.float16, 100, 300, 4.5), (128, torch.float32, 60, 180, 4.4),
                                               # round 5's kernels: fixed-length 16-bit (k_maxsim_stream_uni16), and the 16-row query
                                               # image + 12 waves of the LDS-query kernel (multi-view at dim 768)
                                               (128, torch.float16, 8, 8, 4.3), (768, torch.float16, 16, 16, 4.3)])
def test_candidates_beyond_4_gib_of_index(ca, h, dtype, lo, hi, gib):
    """Byte offsets past 2^32: an index of > 4 GiB per kernel family (h = 128 16-bit ragged, the fixed-length fp32 kernel,
    the LDS-query kernel at dim 768, h = 128 fp32 ragged), candidates drawn from its LAST docs, against the float64 closed form
    on the gathered rows (oracle.ragged_scores_f64); rank_forward and the counted form on the same rows."""
    from oracle.maxsim_oracle import ragged_scores_f64
    dev = "cuda"
    esz = 2 if dtype == torch.float16 else 4
    ntok_target = int(gib * (1 << 30) / (h * esz))
    g = torch.Generator().manual_seed(h + lo)
    ndocs = ntok_target // ((lo + hi) // 2)
    doclens = torch.randint(lo, hi + 1, (ndocs,), generator=g).tolist()
    ntok = sum(doclens)
    gd = torch.Generator(device=dev).manual_seed(7)
    idx = torch.empty(ntok, h, dtype=dtype, device=dev)
    step = 1 << 22
    for s in range(0, ntok, step):
        e = min(s + step, ntok)
        idx[s:e] = F.normalize(torch.randn(e - s, h, generator=gd, device=dev), dim=-1).to(dtype)
    assert idx.numel() * esz > (1 << 32)
    r = ca.ColbertRanker.from_device_tensor(idx, doclens)
    nq, ncand, Lq = 3, 40, 32 if lo > 16 else lo
    Q = F.normalize(torch.randn(nq, Lq, h, generator=g), dim=-1)
    cand = torch.randint(ndocs - 400, ndocs, (nq, ncand), generator=g)                 # the tail of the index: offsets > 4 GiB
    assert int(r.doclens_pfxsum[ndocs - 400]) * h * esz > (1 << 32)
    got = r.score_candidates(Q.cuda(), cand.cuda()).cpu()
    cnt = torch.full((nq,), ncand, dtype=torch.int32)
    got_c = r.score_candidates(Q.cuda(), cand.cuda(), cand_count=cnt.cuda()).cpu()
    assert torch.equal(got, got_c)
    offs, pad = r.doclens_pfxsum, r.d_pad_len.cpu()
    rows_lo = int(offs[ndocs - 400])
    tail = idx[rows_lo:].cpu()                                                           # ~1 % of the index
    for q in range(nq):
        exp = ragged_scores_f64(tail, doclens, (offs[:-1] - rows_lo).tolist(), pad.tolist(), Q[q], cand[q].tolist())
        np.testing.assert_allclose(got[q].numpy(), exp, rtol=0, atol=1e-3 if dtype == torch.float16 else 1e-4)
    p, s = r.rank_forward(Q[:1].cuda().permute(0, 2, 1), cand[0].tolist(), depth=5)
    order = np.argsort(-got[0].numpy(), kind="stable")[:5]
    assert s == got[0].numpy()[order].tolist() and p == cand[0][order].tolist()
