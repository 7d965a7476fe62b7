"""GPU parity tests: the HIP path (through the C ABI of libmaxsim.so) against the CPU oracle and the golden
vectors.  Tolerances: fp32 scores |d| <= 1e-4 absolute (scores have magnitude <= Lq = 32), as stated in
BASELINE/SURVEY 8c; 16-bit-input paths |d| <= 1e-3 against the oracle run in fp32 on identically rounded inputs."""
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


# ------------------------------------------------------------------------------------------------------
# score(): the operator seam, BaseModel.py:39-46
# ------------------------------------------------------------------------------------------------------
def test_kat_test_score(ca, golden):
    g = golden("kat_test_score")
    out = ca.score(g["Q"].cuda(), g["D"].cuda(), g["q_mask"].cuda(), g["d_mask"].cuda())
    assert out.dtype == torch.float32 and out.is_cuda
    assert out.cpu().tolist() == [[21.0, 41.0]]            # BaseModel.py:70-75


def test_zero_floor(ca, golden):
    g = golden("zero_floor")
    Q, D, qm = g["Q"].cuda(), g["D"].cuda(), g["q_mask"].cuda()
    assert ca.score(Q, D, qm, g["d_mask_full"].cuda()).item() == -1.0
    assert ca.score(Q, D, qm, g["d_mask_floor"].cuda()).item() == 0.0


@pytest.mark.parametrize("name", ["c1_1q_10d", "allpairs_4x6_masked", "allpairs_4x6_floatmask", "c4_multiview"])
def test_dense_goldens(ca, golden, name):
    g = golden(name)
    Qc, Dc = g["Q"].cuda(), g["D"].cuda()
    Q0, D0 = Qc.clone(), Dc.clone()
    out = ca.score(Qc, Dc, g["q_mask"].cuda(), g["d_mask"].cuda())
    assert out.shape == g["expected"].shape and out.dtype == g["expected"].dtype
    torch.testing.assert_close(out.cpu(), g["expected"], rtol=0, atol=ATOL32)
    assert torch.equal(Qc, Q0) and torch.equal(Dc, D0)      # inputs are borrowed, never mutated


def test_c5_bf16_golden(ca, golden):
    g = golden("c5_bf16_768")
    out = ca.score(g["Q"].cuda(), g["D"].cuda(), g["q_mask"].cuda(), g["d_mask"].cuda())
    # torch promotion: bf16 * int64 -> bf16 (the reference returns the promoted dtype)
    assert out.dtype == torch.bfloat16
    out32 = ca.score(g["Q"].cuda(), g["D"].cuda(), g["q_mask"].float().cuda(), g["d_mask"].float().cuda())
    assert out32.dtype == torch.float32
    torch.testing.assert_close(out32.cpu(), g["expected"], rtol=0, atol=ATOL16)


def test_model_object_is_dropin(ca):
    from oracle.maxsim_oracle import ref_score
    gen = torch.Generator().manual_seed(11)
    Q, D = nrm(gen, 1, 32, 128), nrm(gen, 37, 50, 128)
    mask = (torch.arange(50) + 1).unsqueeze(0) <= torch.randint(1, 51, (37, 1), generator=gen)
    m = ca.MaxSimModel()
    # exactly the call colbert_ranker.py:111-112 makes
    out = m.score(Q=Q.cuda(), D=D.cuda(), q_mask=torch.ones((1, 32), dtype=torch.long).cuda(),
                  d_mask=mask.to(torch.long).cuda())[0].cpu()
    exp = ref_score(Q, D, torch.ones(1, 32, dtype=torch.long), mask.long())[0]
    torch.testing.assert_close(out, exp, rtol=0, atol=ATOL32)


@pytest.mark.parametrize("shape", [(3, 5, 7, 9, 16), (2, 33, 40, 65, 128), (1, 1, 1, 1, 1), (2, 32, 3, 31, 128), (2, 65, 5, 40, 256),
                                   (2, 32, 3, 33, 128), (5, 17, 4, 64, 128), (1, 40, 2, 10, 64)])
@pytest.mark.parametrize("mask_dtype", [torch.int64, torch.float32, torch.bool, torch.int32])
def test_dense_random_shapes(ca, shape, mask_dtype):
    from oracle.maxsim_oracle import ref_score
    nq, Lq, nd, Ld, h = shape
    gen = torch.Generator().manual_seed(sum(shape))
    Q, D = nrm(gen, nq, Lq, h), nrm(gen, nd, Ld, h)
    qm = (torch.rand(nq, Lq, generator=gen) > 0.2).to(mask_dtype)
    dm = (torch.rand(nd, Ld, generator=gen) > 0.3).to(mask_dtype)
    out = ca.score(Q.cuda(), D.cuda(), qm.cuda(), dm.cuda())
    exp = ref_score(Q, D, qm, dm)
    assert out.dtype == exp.dtype
    torch.testing.assert_close(out.cpu(), exp, rtol=0, atol=ATOL32)


def test_dense_errors(ca):
    Q = torch.zeros(1, 2, 4).cuda()
    with pytest.raises(IndexError):           # max over an empty doc axis, BaseModel.py:44
        ca.score(Q, torch.zeros(2, 0, 4).cuda(), torch.ones(1, 2).cuda(), torch.ones(2, 0).cuda())
    with pytest.raises(ValueError):
        ca.score(Q, torch.zeros(2, 3, 5).cuda(), torch.ones(1, 2).cuda(), torch.ones(2, 3).cuda())
    out = ca.score(torch.zeros(2, 0, 4).cuda(), torch.zeros(3, 2, 4).cuda(), torch.ones(2, 0).cuda(), torch.ones(3, 2).cuda())
    assert out.shape == (2, 3) and float(out.abs().sum()) == 0.0


def test_f32_mfma_is_an_exact_fmaf_chain(ca):
    """Bit-exactness of the flagship kernel: its scores equal a CPU fp32 fmaf chain walked in the kernel's
    k-order, max over tokens, pairwise-tree sum over the 32 query-token lanes -- no tolerance."""
    from oracle.maxsim_oracle import score_chain_f32
    gen = torch.Generator().manual_seed(3)
    Q, D = nrm(gen, 1, 32, 128), nrm(gen, 3, 45, 128)
    ones_q, ones_d = torch.ones(1, 32), torch.ones(3, 45)
    order = []
    for s in range(4):
        for u in range(4):
            for t in range(4):
                order += [32 * s + 8 * u + t, 32 * s + 8 * u + 4 + t]
    # per (q-token, doc) maxima with the chain oracle, then the kernel's pairwise-tree sum over 32 lanes
    Qm, Dm = Q.numpy(), D.numpy()
    acc = np.zeros((3, 32, 45), dtype=np.float32)
    for k in order:
        prod = Qm[0, None, :, None, k].astype(np.float64) * Dm[:, None, :, k].astype(np.float64)
        acc = (prod + acc.astype(np.float64)).astype(np.float32)
    mx = acc.max(-1)                                    # [3, 32]
    exp = []
    for d in range(3):
        v = mx[d].copy()
        while len(v) > 1:                                   # ((q0+q1)+(q2+q3))+... as the kernel's DPP adds do
            v = (v[0::2] + v[1::2]).astype(np.float32)
        exp.append(v[0])
    out = ca.score(Q.cuda(), D.cuda(), ones_q.cuda(), ones_d.cuda()).cpu().numpy()[0]
    assert out.tobytes() == np.array(exp, dtype=np.float32).tobytes()
    # and the generic chain helper agrees to rounding
    np.testing.assert_allclose(score_chain_f32(Q, D, ones_q, ones_d, order)[0], out, rtol=0, atol=1e-5)


def test_f32_rerank_is_an_exact_fmaf_chain(ca):
    """The fp32 rerank kernel (v_mfma_f32_16x16x4_f32, two 16-column blocks): bit-equal to a CPU fp32 fmaf chain in
    the kernel's k-order (instruction (j, t) consumes dims 16 j + 4 g + t, g = 0..3, one after the other), max over
    the doc's tokens, pairwise-tree sum over the query tokens -- no tolerance; also for <= 16 query tokens."""
    gen = torch.Generator().manual_seed(31)
    L, nd = 45, 5
    D = nrm(gen, nd * L, 128)
    r = ca.ColbertRanker(parts=[D], parts_doclens=[[L] * nd], dim=128, index_dtype=torch.float32)
    order = [16 * j + 4 * g + t for j in range(8) for t in range(4) for g in range(4)]
    Dm = D.view(nd, L, 128).numpy()
    for Lq in (32, 16, 9):
        Q = nrm(gen, 1, Lq, 128)
        Qm = Q.numpy()
        acc = np.zeros((nd, Lq, L), dtype=np.float32)
        for k in order:
            prod = Qm[0, None, :, None, k].astype(np.float64) * Dm[:, None, :, k].astype(np.float64)
            acc = (prod + acc.astype(np.float64)).astype(np.float32)      # fmaf: one rounding per step
        mx = acc.max(-1)                                                    # [nd, Lq]
        width = 32 if Lq > 16 else 16
        exp = []
        for d in range(nd):
            v = np.zeros(width, dtype=np.float32)
            v[:Lq] = mx[d]                       # dead query-token lanes hold a zero query: similarity 0
            while len(v) > 1:
                v = (v[0::2] + v[1::2]).astype(np.float32)
            exp.append(v[0])
        out = r.score_candidates(Q, torch.arange(nd)[None]).cpu().numpy()[0]
        assert out.tobytes() == np.array(exp, dtype=np.float32).tobytes(), Lq


# ------------------------------------------------------------------------------------------------------
# rank_forward / fused ragged rerank: colbert_ranker.py:75-137
# ------------------------------------------------------------------------------------------------------
def _golden_ranker(ca, g, dtype):
    parts = [g["part0"], g["part1"]]
    pdl = [g["doclens0"].tolist(), g["doclens1"].tolist()]
    return ca.ColbertRanker(parts=parts, parts_doclens=pdl, dim=128, index_dtype=dtype)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
def test_ragged_rerank_golden(ca, golden, dtype):
    g = golden("ragged_rerank_64")
    r = _golden_ranker(ca, g, dtype)
    assert r.strides == g["strides"].tolist()
    assert torch.equal(r.d_pad_len.cpu().long(), g["pad_len"])
    pids = g["pids"]
    for key, exp in (("Q", "expected_scores"), ("Q_neg", "expected_scores_neg")):
        sc = r.score_candidates(g[key].permute(0, 2, 1), pids.view(1, -1))
        torch.testing.assert_close(sc.cpu()[0], g[exp], rtol=0, atol=ATOL32)
    tp, ts = r.rank_forward(g["Q"], pids.tolist(), depth=10)
    assert isinstance(tp, list) and isinstance(ts, list) and isinstance(tp[0], int) and isinstance(ts[0], float)
    assert tp == g["top10_pids"].tolist()
    np.testing.assert_allclose(ts, g["top10_scores"].numpy(), rtol=0, atol=ATOL32)
    # LongTensor pids and depth > n
    tp2, ts2 = r.rank_forward(g["Q"].cuda(), pids[:7], depth=10)
    assert len(tp2) == 7 and ts2 == sorted(ts2, reverse=True)


def test_ranker_from_index_files_and_output_D(ca, golden, tmp_path):
    """The reference's on-disk format ({i}.pt + doclens.{i}.json) loads into the HBM-resident ranker
    (colbert_ranker.py:16-29,61-73), and output_D_embedding=True returns (pids, D, mask) (:131-136)."""
    from colbert_amd import index_io
    from oracle.maxsim_oracle import RefRanker
    g = golden("ragged_rerank_64")
    parts = [g["part0"], g["part1"]]
    pdl = [g["doclens0"].tolist(), g["doclens1"].tolist()]
    d = str(tmp_path / "idx")
    index_io.save_index(d, parts, pdl)
    r = ca.ColbertRanker(index_path=d, model=ca.MaxSimModel(), dim=128)        # the reference's ctor keywords
    assert r.tensor.dtype == torch.float16 and r.num_embeddings == sum(map(sum, pdl))
    tp, ts = r.rank_forward(g["Q"], g["pids"].tolist(), depth=10)
    assert tp == g["top10_pids"].tolist()
    np.testing.assert_allclose(ts, g["top10_scores"].numpy(), rtol=0, atol=ATOL32)
    # output_D_embedding: the reference can only cat() candidates of ONE length bucket; pick such a set
    ref = RefRanker(parts, pdl, dim=128)
    pad = ref.bucket_strides(list(range(64)))
    S = int(pad.max())
    same = [i for i in range(64) if int(pad[i]) == S][:6]
    ep, eD, em = ref.rank_forward(g["Q"], same, depth=4, output_D_embedding=True)
    gp, gD, gm = r.rank_forward(g["Q"], same, depth=4, output_D_embedding=True)
    assert gp == ep and tuple(gD.shape) == tuple(eD.shape) and torch.equal(gm.cpu(), em)
    # EVERY slot equals the reference's strided view (colbert_ranker.py:49,105): under the mask the doc's own tokens, past
    # the doc's end the next docs' tokens (or the zero tail behind the last doc) -- no multiplication by the mask here
    assert bool((~em).any()) and float(eD[~em].abs().max()) > 0            # the aliased slots exist and are not zero
    torch.testing.assert_close(gD.cpu(), eD, rtol=0, atol=0)
    # ... including the docs at the very end of the index, whose slots run into the +512 zero rows (:62)
    last = [i for i in range(64) if int(pad[i]) == int(pad[63])][-5:]
    ep, eD, em = ref.rank_forward(g["Q"], last, depth=5, output_D_embedding=True)
    gp, gD, gm = r.rank_forward(g["Q"], last, depth=5, output_D_embedding=True)
    assert gp == ep and torch.equal(gm.cpu(), em)
    torch.testing.assert_close(gD.cpu(), eD, rtol=0, atol=0)
    with pytest.raises(RuntimeError):
        r.rank_forward(g["Q"], list(range(64)), depth=64, output_D_embedding=True)   # spans several buckets
    # the reference cat()s the D of ALL candidates (:132), so it fails whenever they span several buckets -- also when the
    # top-`depth` docs alone would share one
    other = [i for i in range(64) if int(pad[i]) != S][:1]
    with pytest.raises(RuntimeError):
        ref.rank_forward(g["Q"], same + other, depth=1, output_D_embedding=True)
    with pytest.raises(RuntimeError):
        r.rank_forward(g["Q"], same + other, depth=1, output_D_embedding=True)


def test_rank_forward_asserts(ca, golden):
    g = golden("ragged_rerank_64")
    r = _golden_ranker(ca, g, torch.float16)
    with pytest.raises(AssertionError):
        r.rank_forward(g["Q"], [], depth=3)                         # colbert_ranker.py:76
    with pytest.raises(AssertionError):
        r.rank_forward(g["Q"].expand(3, -1, -1), [1, 2], depth=3)   # colbert_ranker.py:77


def _random_index(gen, ndocs, h, lo, hi, dtype=torch.float16):
    doclens = torch.randint(lo, hi + 1, (ndocs,), generator=gen).tolist()
    half = ndocs // 2
    pdl = [doclens[:half], doclens[half:]]
    parts = [nrm(gen, sum(d), h).to(dtype) for d in pdl]
    return parts, pdl


@pytest.mark.parametrize("cfg", [
    dict(ndocs=300, h=128, lo=1, hi=180, nq=5, ncand=97, Lq=32, dtype=torch.float32),
    dict(ndocs=300, h=128, lo=170, hi=180, nq=3, ncand=64, Lq=20, dtype=torch.float32),
    dict(ndocs=64, h=128, lo=8, hi=8, nq=4, ncand=64, Lq=8, dtype=torch.float32),      # multi-view shape
    dict(ndocs=200, h=128, lo=1, hi=90, nq=3, ncand=50, Lq=32, dtype=torch.float16),
    dict(ndocs=201, h=128, lo=1, hi=180, nq=4, ncand=77, Lq=32, dtype=torch.bfloat16),  # 16-bit MFMA, 3-way Q split
    dict(ndocs=202, h=128, lo=150, hi=180, nq=2, ncand=40, Lq=7, dtype=torch.float16),
    dict(ndocs=40, h=64, lo=1, hi=40, nq=2, ncand=30, Lq=12, dtype=torch.float32),      # generic kernel
    dict(ndocs=12, h=768, lo=100, hi=256, nq=2, ncand=12, Lq=32, dtype=torch.bfloat16),   # wide kernel, Q hi+lo
    dict(ndocs=14, h=768, lo=100, hi=256, nq=2, ncand=14, Lq=32, dtype=torch.bfloat16, qdtype=torch.bfloat16),
    dict(ndocs=30, h=256, lo=1, hi=70, nq=3, ncand=30, Lq=32, dtype=torch.float32),        # wide kernel, f32 MFMA
    dict(ndocs=31, h=384, lo=1, hi=70, nq=2, ncand=31, Lq=9, dtype=torch.float16),
    dict(ndocs=32, h=1024, lo=20, hi=40, nq=2, ncand=32, Lq=32, dtype=torch.float16, qdtype=torch.float16),
    dict(ndocs=20, h=640, lo=1, hi=40, nq=1, ncand=20, Lq=32, dtype=torch.float32),
    dict(ndocs=20, h=1152, lo=1, hi=40, nq=1, ncand=20, Lq=4, dtype=torch.bfloat16),        # > 1024: generic kernel
    dict(ndocs=50, h=64, lo=1, hi=90, nq=3, ncand=40, Lq=32, dtype=torch.float16),          # partial 128-dim block
    dict(ndocs=50, h=96, lo=1, hi=90, nq=2, ncand=40, Lq=20, dtype=torch.float32),
    dict(ndocs=50, h=200, lo=30, hi=70, nq=2, ncand=40, Lq=32, dtype=torch.bfloat16, qdtype=torch.bfloat16),
    dict(ndocs=50, h=20, lo=1, hi=40, nq=2, ncand=40, Lq=5, dtype=torch.float32),
    dict(ndocs=60, h=128, lo=1, hi=90, nq=3, ncand=50, Lq=40, dtype=torch.float32),         # Lq > 32: two query slices
    dict(ndocs=60, h=128, lo=1, hi=90, nq=2, ncand=50, Lq=64, dtype=torch.float16),
    dict(ndocs=16, h=768, lo=50, hi=120, nq=2, ncand=16, Lq=70, dtype=torch.bfloat16, qdtype=torch.bfloat16),
])
def test_rerank_random_vs_oracle(ca, cfg):
    from oracle.maxsim_oracle import RefRanker
    gen = torch.Generator().manual_seed(cfg["ndocs"] + cfg["h"])
    parts, pdl = _random_index(gen, cfg["ndocs"], cfg["h"], cfg["lo"], cfg["hi"], cfg["dtype"])
    ref = RefRanker(parts, pdl, dim=cfg["h"], index_dtype=cfg["dtype"])
    r = ca.ColbertRanker(parts=parts, parts_doclens=pdl, dim=cfg["h"], index_dtype=cfg["dtype"])
    assert r.strides == ref.strides
    Q = nrm(gen, cfg["nq"], cfg["Lq"], cfg["h"])
    if "qdtype" in cfg:                       # a 16-bit query goes through the ABI in its own dtype
        Qdev = Q.to(cfg["qdtype"])
        Q = Qdev.float()
    else:
        Qdev = Q
    cand = torch.stack([torch.randperm(cfg["ndocs"], generator=gen)[:cfg["ncand"]] for _ in range(cfg["nq"])])
    sc = r.score_candidates(Qdev, cand).cpu()
    # the MFMA paths keep every query bit (split Q) and meet the fp32 tolerance on 16-bit indexes too; only a bf16
    # index with an fp32 query at h > 128 keeps 16 query bits (hi+lo in LDS), and the generic kernel sums in one chain
    loose = cfg["dtype"] == torch.bfloat16 and cfg["h"] != 128 and "qdtype" not in cfg
    atol = ATOL16 if loose else ATOL32
    for qi in range(cfg["nq"]):
        exp = ref.all_scores(Q[qi:qi + 1].permute(0, 2, 1), cand[qi].tolist())
        torch.testing.assert_close(sc[qi], exp, rtol=0, atol=atol)
    # batched top-k agrees with the reference's per-query rank_forward
    tp, ts = r.rerank_batch(Qdev, cand, depth=10)
    for qi in range(cfg["nq"]):
        ep, es = ref.rank_forward(Q[qi:qi + 1].permute(0, 2, 1), cand[qi].tolist(), depth=10)
        np.testing.assert_allclose(ts[qi].cpu().numpy(), np.array(es), rtol=0, atol=atol)
        # pid sets agree except among candidates whose oracle score lies within atol of the k-th score (the only place
        # where a rounding difference or the reference's unstable sort may swap members in and out of the top-k)
        full = ref.all_scores(Q[qi:qi + 1].permute(0, 2, 1), cand[qi].tolist())
        kth = es[-1]
        near = {p for p, v in zip(cand[qi].tolist(), full.tolist()) if abs(v - kth) <= 2 * atol}
        assert (set(tp[qi].tolist()) ^ set(ep)) <= near, (set(tp[qi].tolist()) ^ set(ep), near)


def test_rerank_edge_cases(ca):
    """padding slots, empty docs, q_len, one-token docs, the last doc of the index, duplicates, ncand = 1."""
    from oracle.maxsim_oracle import ragged_scores_f64
    gen = torch.Generator().manual_seed(77)
    doclens = [1, 0, 33, 32, 31, 180, 0, 2, 64, 1]
    parts = [nrm(gen, sum(doclens), 128)]
    r = ca.ColbertRanker(parts=parts, parts_doclens=[doclens], dim=128, index_dtype=torch.float32)
    Q = nrm(gen, 2, 32, 128)
    cand = torch.tensor([[9, -1, 1, 5, 5, 0, 3, 2, 4, 6, 7, 8, 100, 9], [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, -5, 9, 9, 9]])
    q_len = torch.tensor([32, 5], dtype=torch.int32)
    sc = r.score_candidates(Q, cand, q_len=q_len).cpu()
    offs = r.doclens_pfxsum
    for qi in range(2):
        ql = int(q_len[qi])
        for j, pid in enumerate(cand[qi].tolist()):
            if pid < 0 or pid >= len(doclens):
                assert sc[qi, j] == float("-inf")
            elif doclens[pid] == 0:
                assert sc[qi, j] == 0.0
            else:
                e = ragged_scores_f64(parts[0], doclens, offs, r.d_pad_len.cpu(), Q[qi, :ql], [pid])[0]
                assert abs(sc[qi, j].item() - e) <= ATOL32, (qi, j, pid)
    one = r.score_candidates(Q[:1], torch.tensor([[5]])).cpu()
    assert abs(one[0, 0] - sc[0, 3]) == 0.0                   # duplicates / batch composition: bitwise equal
    assert sc[0, 3] == sc[0, 4]


def test_misaligned_operands_take_the_generic_path(ca):
    """Device pointers that are not 16-byte aligned (a C caller's sub-buffer; here: views at a 4-byte offset) must
    not reach the 16-byte LDS-DMA kernels: same scores as the aligned call, to tolerance."""
    gen = torch.Generator().manual_seed(15)
    doclens = [7, 180, 33, 1, 64]
    ntok = sum(doclens)
    emb, Q = nrm(gen, ntok, 128), nrm(gen, 2, 32, 128)
    cand = torch.tensor([[4, 0, 1, 2, 3], [1, 1, 3, -1, 0]])
    r0 = ca.ColbertRanker(parts=[emb], parts_doclens=[doclens], dim=128, index_dtype=torch.float32)
    base = r0.score_candidates(Q, cand).cpu()
    buf = torch.empty(ntok * 128 + 1, device="cuda")
    buf[1:] = emb.flatten().cuda()
    idx_mis = buf[1:].view(ntok, 128)
    assert idx_mis.data_ptr() % 16 == 4 and idx_mis.is_contiguous()
    r1 = ca.ColbertRanker.from_device_tensor(idx_mis, doclens)
    qb = torch.empty(2 * 32 * 128 + 1, device="cuda")
    qb[1:] = Q.flatten().cuda()
    Q_mis = qb[1:].view(2, 32, 128)
    for rr, qq in ((r1, Q.cuda()), (r0, Q_mis), (r1, Q_mis)):
        got = rr.score_candidates(qq, cand).cpu()
        fin = torch.isfinite(base)
        assert torch.equal(fin, torch.isfinite(got))
        assert float((got[fin] - base[fin]).abs().max()) <= ATOL32
    D = nrm(gen, 3, 20, 128)
    db = torch.empty(D.numel() + 1, device="cuda")
    db[1:] = D.flatten().cuda()
    ones_q, ones_d = torch.ones(2, 32).cuda(), torch.ones(3, 20).cuda()
    torch.testing.assert_close(ca.score(Q_mis, db[1:].view(3, 20, 128), ones_q, ones_d),
                               ca.score(Q.cuda(), D.cuda(), ones_q, ones_d), rtol=0, atol=ATOL32)


# ------------------------------------------------------------------------------------------------------
# top-k: colbert_ranker.py:128-130
# ------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("ncand,k", [(1, 1), (7, 3), (1000, 100), (1000, 10), (1024, 1024), (1025, 5), (16384, 100),
                                      (5, 9),
                                      # every size of the register sort network against torch (2 / 4 / 8 keys per thread)
                                      (2049, 5), (3000, 100), (5000, 3000), (8192, 10)])
def test_topk_vs_torch(ca, ncand, k):
    gen = torch.Generator().manual_seed(ncand + k)
    nq = 3
    s = torch.randn(nq, ncand, generator=gen)
    s[0, : ncand // 2] = s[0, ncand // 2: 2 * (ncand // 2)]     # ties
    if ncand > 3:
        s[1, 2] = float("-inf")
    pids = torch.randint(0, 10 ** 12, (nq, ncand), generator=gen)
    r = ca.ColbertRanker(parts=[torch.zeros(4, 8)], parts_doclens=[[1, 1, 1, 1]], dim=8)
    tp, ts = r.topk(s.cuda(), pids.cuda(), k)
    kk = min(k, ncand)
    es, ei = torch.sort(s, dim=1, descending=True, stable=True)
    assert torch.equal(ts.cpu()[:, :kk], es[:, :kk])
    assert torch.equal(tp.cpu()[:, :kk], torch.gather(pids, 1, ei[:, :kk]))     # stable: lower position first
    if k > ncand:
        assert bool((ts.cpu()[:, ncand:] == float("-inf")).all()) and bool((tp.cpu()[:, ncand:] == -1).all())
    tp2, _ = r.topk(s.cuda(), None, kk)
    assert torch.equal(tp2.cpu(), ei[:, :kk])


@pytest.mark.parametrize("width,count", [(5000, 1500), (5000, 2048), (16384, 1025), (16384, 3000), (9000, 5000)])
def test_counted_topk_vs_torch(ca, width, count):
    """Counted rows against torch on the live prefix (not against the uncounted kernel): a row wider than 2048 whose count
    lands in (1024, 2048] is sorted as 2048 keys (2 per thread), 3000 as 4096, 5000 as 8192."""
    gen = torch.Generator().manual_seed(width + count)
    nq, k = 3, 100
    s = torch.randn(nq, width, generator=gen)
    s[0, : count // 2] = s[0, count // 2: 2 * (count // 2)]                 # ties inside the live part
    pids = torch.randint(0, 10 ** 12, (nq, width), generator=gen)
    counts = torch.tensor([count, count - 1, 1], dtype=torch.int32)
    for q in range(nq):
        s[q, int(counts[q]):] = float("-inf")
        pids[q, int(counts[q]):] = -1
    r = ca.ColbertRanker(parts=[torch.zeros(4, 8)], parts_doclens=[[1, 1, 1, 1]], dim=8)
    tp, ts = r.topk(s.cuda(), pids.cuda(), k, counts.cuda())
    for q in range(nq):
        c = int(counts[q])
        es, ei = torch.sort(s[q, :c], descending=True, stable=True)
        kk = min(k, c)
        assert torch.equal(ts[q, :kk].cpu(), es[:kk]) and torch.equal(tp[q, :kk].cpu(), pids[q, :c][ei[:kk]]), q
        assert bool((ts[q, kk:].cpu() == float("-inf")).all()) and bool((tp[q, kk:].cpu() == -1).all())


# ------------------------------------------------------------------------------------------------------
# BASELINE config C2 at full candidate-batch size: size-independent properties
# ------------------------------------------------------------------------------------------------------
def test_c2_full_batch_properties(ca):
    """256 queries x 1000 candidates, 32 x 180 tokens, dim 128 fp32 (BASELINE configs[1]) on a 20k-doc index:
    (i) a sample of scores against the CPU oracle, (ii) invariance to candidate order and batch composition
    (bitwise), (iii) top-k sorted and drawn from the candidates, (iv) a doc queried with its own first 32
    tokens scores ~32 (unit vectors: each query token finds itself)."""
    from oracle.maxsim_oracle import ref_score
    dev = "cuda"
    gen = torch.Generator(device=dev).manual_seed(1234)
    ndocs, L, h, nq, ncand = 20000, 180, 128, 256, 1000
    idx = F.normalize(torch.randn(ndocs * L, h, generator=gen, device=dev), dim=-1)
    r = ca.ColbertRanker(parts=[idx], parts_doclens=[[L] * ndocs], dim=h, index_dtype=torch.float32)
    assert r.strides == [L]
    Q = F.normalize(torch.randn(nq, 32, h, generator=gen, device=dev), dim=-1)
    cand = torch.stack([torch.randperm(ndocs, generator=gen, device=dev)[:ncand] for _ in range(nq)])
    sc = r.score_candidates(Q, cand)
    assert sc.shape == (nq, ncand) and bool(torch.isfinite(sc).all())
    # (i) oracle on a sample
    D3 = idx.view(ndocs, L, h)
    for qi in (0, 17, 255):
        cols = torch.arange(0, ncand, 53, device=dev)
        Dq = D3[cand[qi, cols]].cpu()
        exp = ref_score(Q[qi:qi + 1].cpu(), Dq, torch.ones(1, 32, dtype=torch.long), torch.ones(len(cols), L, dtype=torch.long))[0]
        torch.testing.assert_close(sc[qi, cols].cpu(), exp, rtol=0, atol=ATOL32)
    # (ii) permutation / batch-composition invariance, bitwise
    perm = torch.randperm(ncand, generator=gen, device=dev)
    sc_p = r.score_candidates(Q, cand[:, perm])
    assert torch.equal(sc_p, sc[:, perm])
    sc_sub = r.score_candidates(Q[100:103], cand[100:103, :77])
    assert torch.equal(sc_sub, sc[100:103, :77])
    # (iii) top-k
    tp, ts = r.rerank_batch(Q, cand, depth=100)
    assert tp.shape == (nq, 100)
    assert bool((ts[:, :-1] >= ts[:, 1:]).all())
    es, ei = torch.sort(sc, dim=1, descending=True, stable=True)
    assert torch.equal(ts, es[:, :100]) and torch.equal(tp, torch.gather(cand, 1, ei[:, :100]))
    # (iv) self-retrieval
    own = D3[cand[:, 0], :32].contiguous()
    s_own = r.score_candidates(own, cand[:, :1])
    torch.testing.assert_close(s_own.cpu(), torch.full((nq, 1), 32.0), rtol=0, atol=1e-3)


@pytest.mark.parametrize("name,L,h,Lq,dtype,ndocs,atol", [
    ("c4", 8, 128, 8, torch.float32, 200000, ATOL32),       # BASELINE configs[3]: multi-view, 8 viewer tokens per doc
    ("c4_q32", 8, 128, 32, torch.float32, 200000, ATOL32),  # its secondary form: 32 query tokens
    ("c5", 256, 768, 32, torch.bfloat16, 6000, ATOL16),     # BASELINE configs[4]: bf16, dim 768, 256 doc tokens
])
def test_c4_c5_full_batch_properties(ca, name, L, h, Lq, dtype, ndocs, atol):
    """256 queries x 1000 candidates at the BASELINE shapes C4 / C5: a sample against the CPU oracle (fp32 on the
    identically rounded inputs), bitwise invariance to candidate order and batch composition, top-k consistency,
    self-retrieval (a doc queried with its own first Lq tokens scores ~Lq)."""
    from oracle.maxsim_oracle import ref_score
    dev = "cuda"
    gen = torch.Generator(device=dev).manual_seed(4321)
    nq, ncand = 256, 1000
    idx = F.normalize(torch.randn(ndocs * L, h, generator=gen, device=dev), dim=-1).to(dtype)
    r = ca.ColbertRanker(parts=[idx], parts_doclens=[[L] * ndocs], dim=h, index_dtype=dtype)
    assert r.strides == [L]
    Q = F.normalize(torch.randn(nq, Lq, h, generator=gen, device=dev), dim=-1).to(dtype)
    cand = torch.stack([torch.randperm(ndocs, generator=gen, device=dev)[:ncand] for _ in range(nq)])
    sc = r.score_candidates(Q, cand)
    assert sc.shape == (nq, ncand) and sc.dtype == torch.float32 and bool(torch.isfinite(sc).all())
    D3 = idx.view(ndocs, L, h)
    for qi in (0, 101, 255):
        cols = torch.arange(0, ncand, 97, device=dev)
        Dq = D3[cand[qi, cols]].cpu().float()
        exp = ref_score(Q[qi:qi + 1].cpu().float(), Dq, torch.ones(1, Lq, dtype=torch.long),
                        torch.ones(len(cols), L, dtype=torch.long))[0]
        torch.testing.assert_close(sc[qi, cols].cpu(), exp, rtol=0, atol=atol)
    perm = torch.randperm(ncand, generator=gen, device=dev)
    assert torch.equal(r.score_candidates(Q, cand[:, perm]), sc[:, perm])
    assert torch.equal(r.score_candidates(Q[7:10], cand[7:10, :130]), sc[7:10, :130])
    tp, ts = r.rerank_batch(Q, cand, depth=100)
    es, ei = torch.sort(sc, dim=1, descending=True, stable=True)
    assert torch.equal(ts, es[:, :100]) and torch.equal(tp, torch.gather(cand, 1, ei[:, :100]))
    lq_own = min(Lq, L)
    own = D3[cand[:, 0], :lq_own].contiguous()
    s_own = r.score_candidates(own, cand[:, :1])
    torch.testing.assert_close(s_own.cpu(), torch.full((nq, 1), float(lq_own)), rtol=0, atol=0.05 if dtype != torch.float32 else 1e-3)


@pytest.mark.parametrize("index_dtype", [torch.float32, torch.float16])
def test_rerank_bsize_candidates(ca, index_dtype):
    """The reference's largest candidate list: BSIZE = 16384 = 32 query tokens x faiss_depth 512 (colbert_ranker.py:11),
    ragged docs, duplicates and padding slots included; full score vector against the oracle's closed form on a
    sample, top-k against torch.sort on the GPU scores."""
    from oracle.maxsim_oracle import ragged_scores_f64
    gen = torch.Generator().manual_seed(16384)
    ndocs = 3000
    doclens = torch.randint(1, 181, (ndocs,), generator=gen).tolist()
    part = nrm(gen, sum(doclens), 128).to(index_dtype)
    r = ca.ColbertRanker(parts=[part], parts_doclens=[doclens], dim=128, index_dtype=index_dtype)
    Q = nrm(gen, 2, 32, 128)
    cand = torch.randint(0, ndocs, (2, 16384), generator=gen)
    cand[0, 5] = -1
    cand[1, 16383] = ndocs + 7
    sc = r.score_candidates(Q, cand).cpu()
    assert sc[0, 5] == float("-inf") and sc[1, 16383] == float("-inf")
    cols = list(range(0, 16383, 331)) + [16382]
    for qi in range(2):
        pids = [int(cand[qi, c]) for c in cols if c != 5]
        e = ragged_scores_f64(part.float(), doclens, r.doclens_pfxsum, r.d_pad_len.cpu(), Q[qi], pids)
        got = torch.tensor([sc[qi, c].item() for c in cols if c != 5], dtype=torch.float64)
        assert float((got - torch.as_tensor(e, dtype=torch.float64)).abs().max()) <= (ATOL32 if index_dtype == torch.float32 else ATOL16)
    tp, ts = r.rerank_batch(Q, cand, depth=100)
    es, ei = torch.sort(sc, dim=1, descending=True, stable=True)
    assert torch.equal(ts.cpu(), es[:, :100]) and torch.equal(tp.cpu(), torch.gather(cand, 1, ei[:, :100]))


# ------------------------------------------------------------------------------------------------------
# candidate-side glue: colbert_ranker.py:163-174 (emb2pid) + :212-229 (per-query set())
# ------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n", [1, 37, 1000, 1025, 2048, 4096, 6000, 8192, 9000, 16384])
def test_embedding_ids_to_pids(ca, n):
    gen = torch.Generator().manual_seed(n)
    doclens = torch.randint(0, 40, (500,), generator=gen).tolist()       # includes empty docs
    doclens[0] = 3
    ntok = sum(doclens)
    r = ca.ColbertRanker(parts=[torch.zeros(ntok, 8)], parts_doclens=[doclens], dim=8)
    # the reference's table, build_emb2pid (colbert_ranker.py:163-174)
    emb2pid = torch.zeros(ntok, dtype=torch.int64)
    o = 0
    for pid, dl in enumerate(doclens):
        emb2pid[o:o + dl] = pid
        o += dl
    nq = 3
    e = torch.randint(0, ntok, (nq, n), generator=gen)
    if n >= 37:
        e[0, :30] = e[0, 0]                    # heavy duplication
        e[1, 5] = -1                           # FAISS "no neighbour"
        e[2, :] = torch.randint(0, 50, (n,), generator=gen)
    cand, cnt = r.embedding_ids_to_pids(e, trim=False)
    assert cand.shape == (nq, n)
    for qi in range(nq):
        ids = e[qi][e[qi] >= 0]
        exp = sorted(set(emb2pid[ids].tolist()))                          # uniq(), colbert_ranker.py:234
        c = int(cnt[qi])
        assert cand[qi, :c].tolist() == exp
        assert bool((cand[qi, c:] == -1).all())
    cand_t, _ = r.embedding_ids_to_pids(e)
    assert cand_t.shape[1] == max(int(cnt.max()), 1) and torch.equal(cand_t, cand[:, :cand_t.shape[1]])


def test_retrieve_then_rerank_end_to_end(ca):
    """ANN ids -> distinct pids -> fused rerank -> top-k equals the reference flow (search(): colbert_ranker.py:176-181
    + rank_forward) on the oracle."""
    from oracle.maxsim_oracle import RefRanker
    gen = torch.Generator().manual_seed(5)
    doclens = torch.randint(1, 60, (300,), generator=gen).tolist()
    parts = [nrm(gen, sum(doclens), 128).half()]
    ref = RefRanker(parts, [doclens], dim=128)
    r = ca.ColbertRanker(parts=parts, parts_doclens=[doclens], dim=128)
    Q = nrm(gen, 4, 32, 128)
    e = torch.randint(0, sum(doclens), (4, 32 * 16), generator=gen)
    cand, cnt = r.embedding_ids_to_pids(e)
    tp, ts = r.rerank_batch(Q, cand, depth=10)
    for qi in range(4):
        pids = cand[qi, :int(cnt[qi])].tolist()
        ep, es = ref.rank_forward(Q[qi:qi + 1].permute(0, 2, 1), pids, depth=10)
        np.testing.assert_allclose(ts[qi].cpu().numpy(), np.array(es), rtol=0, atol=ATOL32)
        assert tp[qi].tolist() == ep


def test_fp32_bf16x3_mode_is_fp32_accurate(ca):
    """fp32 index, "3 x bf16" contraction (exact bf16 splits, six piece products on the bf16 matrix pipe): its error
    against a float64 evaluation is of the same size as the exact f32 MFMA path's -- fp32-class accuracy -- also for
    un-normalised, large-magnitude embeddings (no magnitude restriction, unlike the fp16-split fast mode)."""
    from oracle.maxsim_oracle import ragged_scores_f64
    gen = torch.Generator().manual_seed(22)
    for scale in (1.0, 3.0e4):
        doclens = torch.randint(1, 181, (200,), generator=gen).tolist()
        emb = nrm(gen, sum(doclens), 128) * scale
        Q = nrm(gen, 3, 32, 128) * scale
        cand = torch.stack([torch.randperm(200, generator=gen)[:90] for _ in range(3)])
        rs = {m: ca.ColbertRanker(parts=[emb], parts_doclens=[doclens], dim=128, index_dtype=torch.float32, fp32_mode=m)
              for m in ("exact", "bf16x3")}
        offs = rs["exact"].doclens_pfxsum
        err = {}
        for m, r in rs.items():
            pad = r.d_pad_len.cpu()                                # the reference's bucket strides (0-floor where it pads)
            sc = r.score_candidates(Q, cand).cpu().double().numpy()
            ref = np.stack([ragged_scores_f64(emb, doclens, offs, pad, Q[qi], cand[qi].tolist()) for qi in range(3)])
            err[m] = np.abs(sc - ref).max() / (scale * scale)
        assert err["exact"] <= 2e-5 and err["bf16x3"] <= 2e-5, err
        assert err["bf16x3"] <= 4 * err["exact"] + 1e-6, err


def test_fp32_fast_mode(ca, golden):
    """fp32 index, opt-in "fast" contraction (both operands split into fp16 pieces, 16-bit MFMA): same scores to 1e-5."""
    from oracle.maxsim_oracle import RefRanker
    gen = torch.Generator().manual_seed(21)
    parts, pdl = _random_index(gen, 300, 128, 1, 180, torch.float32)
    ref = RefRanker(parts, pdl, dim=128, index_dtype=torch.float32)
    exact = ca.ColbertRanker(parts=parts, parts_doclens=pdl, dim=128, index_dtype=torch.float32)
    fast = ca.ColbertRanker(parts=parts, parts_doclens=pdl, dim=128, index_dtype=torch.float32, fp32_mode="fast")
    Q = nrm(gen, 4, 32, 128)
    cand = torch.stack([torch.randperm(300, generator=gen)[:120] for _ in range(4)])
    se, sf = exact.score_candidates(Q, cand).cpu(), fast.score_candidates(Q, cand).cpu()
    for qi in range(4):
        exp = ref.all_scores(Q[qi:qi + 1].permute(0, 2, 1), cand[qi].tolist())
        torch.testing.assert_close(sf[qi], exp, rtol=0, atol=1e-5)
    assert float((se - sf).abs().max()) <= 1e-5
    g = golden("ragged_rerank_64")
    r = ca.ColbertRanker(parts=[g["part0"], g["part1"]], parts_doclens=[g["doclens0"].tolist(), g["doclens1"].tolist()],
                         dim=128, index_dtype=torch.float32, fp32_mode="fast")
    tp, ts = r.rank_forward(g["Q"], g["pids"].tolist(), depth=10)
    assert tp == g["top10_pids"].tolist()
    np.testing.assert_allclose(ts, g["top10_scores"].numpy(), rtol=0, atol=1e-5)


# ------------------------------------------------------------------------------------------------------
# training form: score() under autograd (second caller of the operator, colbert_model.py:87-96)
# ------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("cfg", [
    dict(nq=3, Lq=32, nd=5, Ld=40, h=128, dtype=torch.float32, masks="01"),
    dict(nq=2, Lq=16, nd=4, Ld=33, h=256, dtype=torch.float32, masks="float"),
    dict(nq=4, Lq=32, nd=6, Ld=70, h=768, dtype=torch.bfloat16, masks="01"),
    dict(nq=2, Lq=20, nd=3, Ld=50, h=128, dtype=torch.float16, masks="01"),
    dict(nq=2, Lq=5, nd=3, Ld=7, h=24, dtype=torch.float32, masks="01"),          # generic kernel
    dict(nq=2, Lq=40, nd=2, Ld=9, h=128, dtype=torch.float32, masks="none"),       # Lq > 32: generic kernel
    dict(nq=70, Lq=3, nd=130, Ld=5, h=200, dtype=torch.float32, masks="01"),       # > 64 docs / items per lane batch
    dict(nq=1, Lq=2, nd=1, Ld=700, h=64, dtype=torch.float32, masks="none"),       # long doc: atomic dD fallback
    # enough (doc, query block) tiles for the GEMM-blocked all-pairs kernel (maxsim_allpairs.h), its three row-block forms
    dict(nq=20, Lq=32, nd=70, Ld=384, h=64, dtype=torch.bfloat16, masks="01"),      # R = 3 (doc_maxlen), partial query block
    dict(nq=17, Lq=20, nd=66, Ld=200, h=128, dtype=torch.float16, masks="01"),      # R = 2, rows past Ld, Lq < 32
    dict(nq=24, Lq=32, nd=50, Ld=100, h=768, dtype=torch.bfloat16, masks="none"),   # R = 1, the reference's dim
    dict(nq=9, Lq=7, nd=131, Ld=333, h=96, dtype=torch.float16, masks="01"),        # odd everything
])
def test_score_autograd_matches_torch(ca, cfg):
    from oracle.maxsim_oracle import ref_score
    gen = torch.Generator().manual_seed(cfg["h"] + cfg["Ld"])
    Q0 = nrm(gen, cfg["nq"], cfg["Lq"], cfg["h"]).to(cfg["dtype"])
    D0 = nrm(gen, cfg["nd"], cfg["Ld"], cfg["h"]).to(cfg["dtype"])
    if cfg["masks"] == "01":
        qm = (torch.rand(cfg["nq"], cfg["Lq"], generator=gen) > 0.2).long()
        dm = (torch.rand(cfg["nd"], cfg["Ld"], generator=gen) > 0.3).long()
    elif cfg["masks"] == "float":
        qm = torch.rand(cfg["nq"], cfg["Lq"], generator=gen) + 0.1
        dm = torch.rand(cfg["nd"], cfg["Ld"], generator=gen) + 0.1
    else:
        qm, dm = torch.ones(cfg["nq"], cfg["Lq"], dtype=torch.long), torch.ones(cfg["nd"], cfg["Ld"], dtype=torch.long)
    w = torch.randn(cfg["nq"], cfg["nd"], generator=gen)          # a generic upstream gradient
    # oracle: torch autograd through the reference's four ops, fp32 on the same (rounded) inputs
    Qr, Dr = Q0.float().clone().requires_grad_(True), D0.float().clone().requires_grad_(True)
    out_r = ref_score(Qr, Dr, qm, dm)
    (out_r * w).sum().backward()
    # ours
    Qg, Dg = Q0.detach().cuda().requires_grad_(True), D0.detach().cuda().requires_grad_(True)
    out_g = ca.score(Qg, Dg, qm.cuda(), dm.cuda())
    assert out_g.requires_grad
    (out_g.float() * w.cuda()).sum().backward()
    tol = ATOL32 if cfg["dtype"] == torch.float32 else 2e-2
    torch.testing.assert_close(out_g.float().cpu(), out_r.detach(), rtol=0, atol=ATOL32 if cfg["dtype"] == torch.float32 else 5e-2)
    assert Qg.grad.dtype == cfg["dtype"] and Dg.grad.dtype == cfg["dtype"]
    torch.testing.assert_close(Qg.grad.float().cpu(), Qr.grad, rtol=0, atol=tol)
    torch.testing.assert_close(Dg.grad.float().cpu(), Dr.grad, rtol=0, atol=tol)
    # only D needs grad
    D2 = D0.detach().cuda().requires_grad_(True)
    (ca.score(Q0.cuda(), D2, qm.cuda(), dm.cuda()).float() * w.cuda()).sum().backward()
    torch.testing.assert_close(D2.grad.float().cpu(), Dr.grad, rtol=0, atol=tol)
    # no_grad: plain forward (no arg-max), same values
    with torch.no_grad():
        plain = ca.score(Qg, Dg, qm.cuda(), dm.cuda())
        assert not plain.requires_grad
        torch.testing.assert_close(plain.float().cpu(), out_g.detach().float().cpu(), rtol=0, atol=1e-5 if cfg["dtype"] == torch.float32 else 2e-2)


# ------------------------------------------------------------------------------------------------------
# randomized differential sweep (seeded): every dispatch path against the oracle
# ------------------------------------------------------------------------------------------------------
def _sweep_cases():
    rng = np.random.RandomState(20261004)
    cases = []
    for i in range(36):
        h = int(rng.choice([128, 128, 128, 256, 384, 768, 64, 96, 1024]))
        dtype = [torch.float32, torch.float16, torch.bfloat16][int(rng.randint(3))]
        Lq = int(rng.choice([1, 3, 8, 16, 31, 32, 32]))
        shape = rng.choice(["long", "short", "mixed", "tiny", "multiview"])
        cases.append((i, h, dtype, Lq, str(shape), bool(rng.rand() < 0.3), bool(rng.rand() < 0.3)))
    return cases


@pytest.mark.parametrize("case", _sweep_cases(), ids=lambda c: f"{c[0]}-h{c[1]}-{str(c[2]).split('.')[-1]}-Lq{c[3]}-{c[4]}")
def test_randomized_rerank_sweep(ca, case):
    from oracle.maxsim_oracle import RefRanker
    i, h, dtype, Lq, shape, use_qlen, fast = case
    gen = torch.Generator().manual_seed(1000 + i)
    ndocs = 90
    if shape == "long":
        doclens = torch.randint(100, 300, (ndocs,), generator=gen)
    elif shape == "short":
        doclens = torch.randint(1, 12, (ndocs,), generator=gen)
    elif shape == "mixed":
        doclens = torch.randint(0, 200, (ndocs,), generator=gen)      # includes empty docs
        doclens[::7] = 0
    elif shape == "tiny":
        doclens = torch.randint(1, 3, (ndocs,), generator=gen)
    else:
        doclens = torch.full((ndocs,), 8)
    doclens = doclens.tolist()
    if sum(doclens) == 0:
        doclens[0] = 5
    half = ndocs // 2
    pdl = [doclens[:half], doclens[half:]]
    parts = [nrm(gen, max(sum(d), 0), h).to(dtype) for d in pdl]
    nonempty = [k for k, L in enumerate(doclens) if L > 0]
    # the reference cannot hold empty docs in a bucket view of stride 0; the oracle index uses the same doclens
    ref = RefRanker(parts, pdl, dim=h, index_dtype=dtype)
    kw = dict(fp32_mode="fast") if (fast and dtype == torch.float32 and h == 128) else {}
    r = ca.ColbertRanker(parts=parts, parts_doclens=pdl, dim=h, index_dtype=dtype, **kw)
    nq, ncand = 3, 41
    Q = nrm(gen, nq, Lq, h)
    cand = torch.stack([torch.randint(0, ndocs, (ncand,), generator=gen) for _ in range(nq)])
    cand[0, 3] = -1
    cand[1, 0] = ndocs + 5
    q_len = torch.randint(1, Lq + 1, (nq,), generator=gen).int() if use_qlen else None
    sc = r.score_candidates(Q, cand, q_len=q_len).cpu()
    loose = dtype == torch.bfloat16 and h != 128 and h % 8 == 0 and h <= 1024
    generic16 = dtype != torch.float32 and (h % 8 != 0 or h > 1024)
    atol = ATOL16 if (loose or generic16) else (1e-5 * 10 if kw else ATOL32)
    for qi in range(nq):
        ql = int(q_len[qi]) if use_qlen else Lq
        for j, pid in enumerate(cand[qi].tolist()):
            if pid < 0 or pid >= ndocs:
                assert sc[qi, j] == float("-inf")
            elif doclens[pid] == 0:
                assert sc[qi, j] == 0.0
            else:
                e = ref.all_scores(Q[qi:qi + 1, :ql].permute(0, 2, 1), [pid])[0]
                assert abs(sc[qi, j].item() - e.item()) <= atol, (qi, j, pid, sc[qi, j].item(), e.item())


# ------------------------------------------------------------------------------------------------------
# streams and graphs: the library is asynchronous on the caller's stream and capturable into a hipGraph
# ------------------------------------------------------------------------------------------------------
def test_side_stream_and_hipgraph_replay(ca):
    gen = torch.Generator().manual_seed(9)
    parts, pdl = _random_index(gen, 400, 128, 20, 180, torch.float16)
    r = ca.ColbertRanker(parts=parts, parts_doclens=pdl, dim=128)
    Q = nrm(gen, 8, 32, 128).cuda()
    cand = torch.stack([torch.randperm(400, generator=gen)[:200] for _ in range(8)]).cuda()
    exp_p, exp_s = r.rerank_batch(Q, cand, depth=10)
    torch.cuda.synchronize()
    # a non-default stream
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        p1, s1 = r.rerank_batch(Q, cand, depth=10)
    side.synchronize()
    assert torch.equal(p1, exp_p) and torch.equal(s1, exp_s)
    # capture rerank + top-k into a graph, replay it on new inputs written into the static buffers
    Qs, cs = Q.clone(), cand.clone()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        gp, gs = r.rerank_batch(Qs, cs, depth=10)
    Q2 = nrm(gen, 8, 32, 128).cuda()
    cand2 = torch.stack([torch.randperm(400, generator=gen)[:200] for _ in range(8)]).cuda()
    Qs.copy_(Q2)
    cs.copy_(cand2)
    g.replay()
    torch.cuda.synchronize()
    e2p, e2s = r.rerank_batch(Q2, cand2, depth=10)
    assert torch.equal(gp, e2p) and torch.equal(gs, e2s)


def test_concurrent_streams_and_threads(ca):
    """The library keeps no state between calls: launches racing on several streams, issued from several host threads,
    return bit-identical results to the serial call (fp32, fp16 and dim-768 kernels, i.e. every LDS-attribute path)."""
    import threading
    gen = torch.Generator().manual_seed(99)
    cfgs = []
    for h, dt in ((128, torch.float32), (128, torch.float16), (768, torch.bfloat16)):
        parts, pdl = _random_index(gen, 300, h, 5, 180, dt)
        r = ca.ColbertRanker(parts=parts, parts_doclens=pdl, dim=h, index_dtype=dt)
        Q = nrm(gen, 6, 32, h).cuda()
        cand = torch.stack([torch.randperm(300, generator=gen)[:150] for _ in range(6)]).cuda()
        cfgs.append((r, Q, cand, r.rerank_batch(Q, cand, depth=20)))
    torch.cuda.synchronize()
    errors = []

    def worker(tid):
        try:
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                for it in range(12):
                    r, Q, cand, (ep, es) = cfgs[(tid + it) % len(cfgs)]
                    p_, s_ = r.rerank_batch(Q, cand, depth=20)
                    st.synchronize()
                    if not (torch.equal(p_, ep) and torch.equal(s_, es)):
                        errors.append((tid, it))
        except Exception as e:  # noqa: BLE001
            errors.append((tid, repr(e)))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


def test_sharded_ranker_local_leg_on_gpu(ca):
    """One shard of a doc-sharded index on the GPU (the collective leg is covered by the gloo test): candidates outside
    the shard's pid range are padding, the shard's own are compacted to the front, top-k carries GLOBAL pids."""
    from colbert_amd.sharded import ShardedRanker
    from oracle.maxsim_oracle import RefRanker
    gen = torch.Generator().manual_seed(31)
    parts, pdl = _random_index(gen, 100, 128, 5, 60, torch.float16)
    ref = RefRanker(parts, pdl, dim=128)
    r = ca.ColbertRanker(parts=parts, parts_doclens=pdl, dim=128)
    lo, hi = 1000, 1100                                   # this shard holds global pids [1000, 1100)
    sh = ShardedRanker(r, lo, hi)
    Q = nrm(gen, 3, 32, 128)
    cand = torch.stack([torch.randperm(400, generator=gen)[:64] + 900 for _ in range(3)]).cuda()   # global pids 900..1299
    tp, ts = sh.rerank_batch(Q, cand, depth=8)
    for qi in range(3):
        mine = [p for p in cand[qi].tolist() if lo <= p < hi]
        ep, es = ref.rank_forward(Q[qi:qi + 1].permute(0, 2, 1), [p - lo for p in mine], depth=8)
        n = len(ep)
        assert tp[qi, :n].tolist() == [p + lo for p in ep]
        np.testing.assert_allclose(ts[qi, :n].cpu().numpy(), np.array(es), rtol=0, atol=ATOL32)
        assert bool((tp[qi, n:] == -1).all()) and bool((ts[qi, n:] == float("-inf")).all())


def test_against_plain_c_oracle(ca):
    """The HIP path against the plain-C restatement (no torch, no BLAS, double accumulation): every index dtype."""
    from oracle import c_oracle
    gen = torch.Generator().manual_seed(41)
    doclens = torch.randint(0, 70, (40,), generator=gen).tolist()
    doclens[3] = 0
    emb = nrm(gen, sum(doclens), 128)
    Q = nrm(gen, 2, 32, 128)
    cand = torch.stack([torch.randperm(40, generator=gen)[:25] for _ in range(2)])
    for dt, atol in ((torch.float32, ATOL32), (torch.float16, ATOL32), (torch.bfloat16, ATOL32)):
        idx = emb.to(dt)
        r = ca.ColbertRanker(parts=[idx], parts_doclens=[doclens], dim=128, index_dtype=dt)
        sc = r.score_candidates(Q, cand).cpu().double().numpy()
        offs = r.doclens_pfxsum[:-1].numpy()
        for qi in range(2):
            exp = c_oracle.rerank_one(idx.float().numpy(), offs, np.array(doclens), r.d_pad_len.cpu().numpy(),
                                      Q[qi].numpy(), cand[qi].numpy())
            np.testing.assert_allclose(sc[qi], exp, rtol=0, atol=atol)
    out = ca.score(Q.cuda(), emb[:60].view(3, 20, 128).cuda(), torch.ones(2, 32).cuda(), torch.ones(3, 20).cuda()).cpu().double().numpy()
    exp = c_oracle.score_dense(Q.numpy(), emb[:60].view(3, 20, 128).numpy(), np.ones((2, 32)), np.ones((3, 20)))
    np.testing.assert_allclose(out, exp, rtol=0, atol=ATOL32)


# ------------------------------------------------------------------------------------------------------
# the C ABI without Python: a gcc-built C program linked against libmaxsim.so (tests/capi/kat.c)
# ------------------------------------------------------------------------------------------------------
def test_c_abi_from_plain_c(ca):
    import os
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "capi", "kat")
    if not os.path.exists(exe):          # normally built by __graft_entry__.build(); gcc is on the GPU box too
        import __graft_entry__
        __graft_entry__.build_c_client()
    res = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert res.returncode == 0 and "ALL OK" in res.stdout and "FAIL" not in res.stdout, res.stdout + res.stderr
