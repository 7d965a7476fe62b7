"""CPU: the oracle restatement against the committed golden vectors (generated from the imported reference
``BaseModel.score`` by tests/golden/make_golden.py) and against the reference's only KAT."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle.maxsim_oracle import (RefRanker, keep_nonzero, ragged_scores_f64, ref_score, score_chain_f32,
                                  score_f64, torch_percentile)


def test_kat_test_score(golden):
    g = golden("kat_test_score")            # BaseModel.py:70-75
    out = ref_score(g["Q"], g["D"], g["q_mask"], g["d_mask"])
    assert out.tolist() == [[21.0, 41.0]]
    assert torch.equal(out, g["expected"])


def test_zero_floor(golden):
    g = golden("zero_floor")
    assert ref_score(g["Q"], g["D"], g["q_mask"], g["d_mask_full"]).item() == -1.0
    assert ref_score(g["Q"], g["D"], g["q_mask"], g["d_mask_floor"]).item() == 0.0


@pytest.mark.parametrize("name", ["c1_1q_10d", "allpairs_4x6_masked", "allpairs_4x6_floatmask", "c4_multiview"])
def test_dense_goldens(golden, name):
    g = golden(name)
    out = ref_score(g["Q"], g["D"], g["q_mask"], g["d_mask"])
    assert out.dtype == g["expected"].dtype
    # same ops as the reference; the CPU BLAS blocking may differ between hosts -> 1e-5, not bitwise
    torch.testing.assert_close(out, g["expected"], rtol=0, atol=1e-5)
    f64 = score_f64(g["Q"], g["D"], g["q_mask"], g["d_mask"])
    np.testing.assert_allclose(g["expected"].numpy(), f64, rtol=0, atol=2e-5)


def test_c5_bf16(golden):
    g = golden("c5_bf16_768")
    out = ref_score(g["Q"].float(), g["D"].float(), g["q_mask"], g["d_mask"])
    torch.testing.assert_close(out, g["expected"], rtol=0, atol=2e-5)


def test_chain_oracle_matches(golden):
    g = golden("allpairs_4x6_masked")
    out = score_chain_f32(g["Q"][:2], g["D"][:3], g["q_mask"][:2], g["d_mask"][:3])
    np.testing.assert_allclose(out, g["expected"][:2, :3].numpy(), rtol=0, atol=1e-5)


def _ranker(g, score_fn=ref_score):
    parts = [g["part0"], g["part1"]]
    pdl = [g["doclens0"].tolist(), g["doclens1"].tolist()]
    return RefRanker(parts, pdl, dim=128, score_fn=score_fn)


def test_ragged_rerank_golden(golden):
    g = golden("ragged_rerank_64")
    r = _ranker(g)
    assert r.strides == g["strides"].tolist()
    assert torch.equal(r.bucket_strides(list(range(64))), g["pad_len"])
    pids = g["pids"].tolist()
    sc = r.all_scores(g["Q"], pids)
    torch.testing.assert_close(sc, g["expected_scores"], rtol=0, atol=1e-5)
    scn = r.all_scores(g["Q_neg"], pids)
    torch.testing.assert_close(scn, g["expected_scores_neg"], rtol=0, atol=1e-5)
    tp, ts = r.rank_forward(g["Q"], pids, depth=10)
    assert tp == g["top10_pids"].tolist()
    np.testing.assert_allclose(ts, g["top10_scores"].numpy(), rtol=0, atol=1e-5)
    # the zero floor decides some scores of the negative query
    doclens = torch.cat([g["doclens0"], g["doclens1"]])
    floored = g["pad_len"] > doclens
    assert floored.any() and (~floored).any()


def test_ragged_rerank_768_golden(golden):
    """The reference's default deployment shape (dim 768, fp16 index, ragged docs): oracle restatement + closed form
    against the fixture the imported reference wrote (tests/golden/make_golden.py, section 8)."""
    g = golden("ragged_rerank_768")
    r = RefRanker([g["part0"], g["part1"]], [g["doclens0"].tolist(), g["doclens1"].tolist()], dim=768)
    n = len(r.doclens)
    assert r.strides == g["strides"].tolist() and torch.equal(r.bucket_strides(list(range(n))), g["pad_len"])
    pids = g["pids"].tolist()
    for key, exp in (("Q", "expected_scores"), ("Q_neg", "expected_scores_neg")):
        torch.testing.assert_close(r.all_scores(g[key], pids), g[exp], rtol=0, atol=1e-5)
        f64 = ragged_scores_f64(r.tensor, r.doclens, r.doclens_pfxsum, g["pad_len"], g[key][0].permute(1, 0), pids)
        np.testing.assert_allclose(g[exp].numpy(), f64, rtol=0, atol=3e-5)
    tp, ts = r.rank_forward(g["Q"], pids, depth=10)
    assert tp == g["top10_pids"].tolist()
    np.testing.assert_allclose(ts, g["top10_scores"].numpy(), rtol=0, atol=1e-5)
    floored = g["pad_len"] > torch.cat([g["doclens0"], g["doclens1"]])
    assert floored.any() and (~floored).any()


def test_masked_query_rerank_golden(golden):
    """Masked-token compaction before the rerank (training_utils.py:48-53 via dense_server_client.py:45): the oracle's
    keep_nonzero + RefRanker against the fixture written by the IMPORTED keep_nonzero / qd_mask_to_realinput / score
    (make_golden.py section 9): mid-sequence holes, a padded tail, a query with one live token."""
    from oracle.maxsim_oracle import keep_nonzero
    g = golden("masked_query_rerank")
    r = RefRanker([g["part0"], g["part1"]], [g["doclens0"].tolist(), g["doclens1"].tolist()], dim=128)
    pids = g["pids"].tolist()
    pad_len = r.bucket_strides(list(range(len(r.doclens))))
    for q in range(3):
        real_q, real_m = keep_nonzero(g["Q"][q], g["q_word_mask"][q])
        assert real_q.size(0) == int(g["q_word_mask"][q].sum()) and bool((real_m == 1).all())
        got = r.all_scores(real_q.unsqueeze(0).permute(0, 2, 1).contiguous(), pids)
        torch.testing.assert_close(got, g["expected_scores"][q], rtol=0, atol=1e-5)
        f64 = ragged_scores_f64(r.tensor, r.doclens, r.doclens_pfxsum, pad_len, real_q, pids)
        np.testing.assert_allclose(g["expected_scores"][q].numpy(), f64, rtol=0, atol=3e-5)
    assert int(g["q_word_mask"][2].sum()) == 1


def test_ragged_closed_form_equals_bucketed(golden):
    """The fused kernel's definition (real tokens + analytic 0-floor) equals the reference's bucket/pad/mask."""
    g = golden("ragged_rerank_64")
    r = _ranker(g)
    pids = g["pids"].tolist()
    for key, exp in (("Q", "expected_scores"), ("Q_neg", "expected_scores_neg")):
        q = g[key][0].permute(1, 0)                       # [Lq, h]
        f64 = ragged_scores_f64(r.tensor, r.doclens, r.doclens_pfxsum, g["pad_len"], q, pids)
        np.testing.assert_allclose(g[exp].numpy(), f64, rtol=0, atol=2e-5)
    # and the floor matters: without it the negative query's scores differ
    q = g["Q_neg"][0].permute(1, 0)
    nofloor = ragged_scores_f64(r.tensor, r.doclens, r.doclens_pfxsum, torch.zeros(64, dtype=torch.long), q, pids)
    assert np.abs(nofloor - g["expected_scores_neg"].numpy()).max() > 1e-2


def test_rank_forward_contract():
    gen = torch.Generator().manual_seed(1)
    dl = [[5, 9, 3, 12], [7, 7, 1, 30]]
    parts = [F.normalize(torch.randn(sum(d), 16, generator=gen), dim=-1).half() for d in dl]
    r = RefRanker(parts, dl)
    Q = F.normalize(torch.randn(4, 16, generator=gen), dim=-1).unsqueeze(0).permute(0, 2, 1)
    p, s = r.rank_forward(Q, [7, 0, 3, 5], depth=3)
    assert len(p) == len(s) == 3 and s == sorted(s, reverse=True) and set(p) <= {7, 0, 3, 5}
    p, s = r.rank_forward(Q, torch.tensor([2, 6]), depth=10)
    assert len(p) == 2
    with pytest.raises(AssertionError):
        r.rank_forward(Q, [], depth=3)                    # colbert_ranker.py:76
    with pytest.raises(AssertionError):
        r.rank_forward(Q.expand(3, -1, -1), [1, 2], depth=3)   # colbert_ranker.py:77


def test_percentile_and_keep_nonzero():
    t = torch.tensor([5, 1, 9, 3, 7, 2, 8, 4])
    assert [torch_percentile(t, p) for p in (25, 50, 75)] == [2, 4, 7]
    with pytest.raises(Exception):
        torch_percentile(torch.tensor([1, 2, 3]), 25)     # kthvalue(0): the reference raises for N < 4
    Q = torch.arange(12.).view(4, 3)
    q, m = keep_nonzero(Q, torch.tensor([1, 0, 1, 0]))
    assert q.tolist() == [[0, 1, 2], [6, 7, 8]] and m.tolist() == [1, 1]


@pytest.mark.parametrize("name,view,dim", [("mv768_fp16", 16, 768), ("mv128_fp16", 8, 128)])
def test_multiview_fp16_goldens(golden, name, view, dim):
    """The reference's multi-view deployment shapes in its storage dtype (dense.yaml:8,29-32; colbert_ranker.py:62), written
    by tests/golden/make_golden_multiview.py through the imported get_representation + score: the torch and plain-C
    restatements of `score`, and the restated ranker on an fp16 index of those docs (one length bucket = the view count, no
    0-floor), reproduce the expected matrix."""
    from oracle import c_oracle
    g = golden(name)
    assert g["D"].dtype == torch.float16 and tuple(g["D"].shape[1:]) == (view, dim)
    Q, D, exp = g["Q"], g["D"].float(), g["expected"]
    nq, nd = Q.size(0), D.size(0)
    qm, dm = torch.ones(nq, view, dtype=torch.long), torch.ones(nd, view, dtype=torch.long)
    torch.testing.assert_close(ref_score(Q, D, qm, dm), exp, rtol=0, atol=1e-5)
    np.testing.assert_allclose(c_oracle.score_dense(Q.numpy(), D.numpy(), qm.numpy(), dm.numpy()), exp.numpy(), rtol=0, atol=2e-5)
    r = RefRanker([g["D"].reshape(nd * view, dim)], [[view] * nd], dim=dim)
    assert r.strides == [view]
    for qi in range(nq):
        sc = r.all_scores(Q[qi:qi + 1].permute(0, 2, 1).contiguous(), list(range(nd)))
        torch.testing.assert_close(torch.as_tensor(sc, dtype=torch.float32), exp[qi], rtol=0, atol=1e-5)


# ---- the plain-C restatement (oracle/maxsim_oracle.c), independent of torch/BLAS ---------------------------------
def test_c_oracle_against_goldens_and_torch_oracle(golden):
    from oracle import c_oracle
    g = golden("kat_test_score")
    assert c_oracle.score_dense(g["Q"].numpy(), g["D"].numpy(), g["q_mask"].numpy(), g["d_mask"].numpy()).tolist() == [[21.0, 41.0]]
    g = golden("zero_floor")
    assert c_oracle.score_dense(g["Q"].numpy(), g["D"].numpy(), g["q_mask"].numpy(), g["d_mask_floor"].numpy()).item() == 0.0
    assert c_oracle.score_dense(g["Q"].numpy(), g["D"].numpy(), g["q_mask"].numpy(), g["d_mask_full"].numpy()).item() == -1.0
    for name in ("allpairs_4x6_masked", "allpairs_4x6_floatmask", "c4_multiview"):
        g = golden(name)
        out = c_oracle.score_dense(g["Q"].numpy(), g["D"].numpy(), g["q_mask"].numpy(), g["d_mask"].numpy())
        np.testing.assert_allclose(out, g["expected"].numpy(), rtol=0, atol=2e-5)
    # ragged closed form == the reference's bucket / pad / mask flow (golden generated through the imported score)
    g = golden("ragged_rerank_64")
    index = torch.cat([g["part0"], g["part1"]]).float().numpy()
    doclens = torch.cat([g["doclens0"], g["doclens1"]]).numpy()
    offs = np.concatenate([[0], np.cumsum(doclens)[:-1]])
    for key, exp in (("Q", "expected_scores"), ("Q_neg", "expected_scores_neg")):
        q = g[key][0].permute(1, 0).contiguous().numpy()
        out = c_oracle.rerank_one(index, offs, doclens, g["pad_len"].numpy(), q, g["pids"].numpy())
        np.testing.assert_allclose(out, g[exp].numpy(), rtol=0, atol=2e-5)
