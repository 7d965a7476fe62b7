"""Generates tests/golden/*.npz -- run ONCE in the build container, where the reference is importable:

    PYTHONPATH=/root/reference python tests/golden/make_golden.py

Expected outputs come from the REAL reference operator, ``colbert.modeling.BaseModel.BaseModel.score``
(/root/reference/colbert/modeling/BaseModel.py:39-46), imported here; the rerank fixture drives that imported
``score`` through ``oracle.maxsim_oracle.RefRanker`` (the line-by-line CPU restatement of
colbert/ranking/colbert_ranker.py:16-137, which is itself not importable: module-level ``import faiss``).
The fixtures are data only (inputs + expected outputs); no reference source text is stored.
The script also asserts that the oracle restatement ``ref_score`` equals the imported reference bitwise.
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from colbert.modeling.BaseModel import BaseModel          # the reference (needs PYTHONPATH=/root/reference)
from oracle.maxsim_oracle import RefRanker, ref_score

REF = BaseModel.score


def norm_randn(gen, *shape):
    return F.normalize(torch.randn(*shape, generator=gen), p=2, dim=-1)   # BaseModel.py:26 output contract


def check(Q, D, qm, dm):
    exp = REF(Q, D, qm, dm)
    mine = ref_score(Q, D, qm, dm)
    assert exp.dtype == mine.dtype and torch.equal(exp, mine), "oracle restatement != reference"
    return exp


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            if v.dtype == torch.bfloat16:
                out[k + "__bf16bits"] = v.view(torch.int16).numpy()
                continue
            v = v.numpy()
        out[k] = np.asarray(v)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, {k: (a.shape, str(a.dtype)) for k, a in out.items()})


def main():
    torch.manual_seed(0)
    torch.set_num_threads(1)

    # (1) the reference's only known-answer test, BaseModel.py:70-75 -> [[21, 41]]
    Q = torch.tensor([[[1, 5, 4], [2, 8, 1]]]).float()
    D = torch.tensor([[[0, 0, 0], [1, 1, 1]], [[3, 2, 1], [1, 1, 3]]]).float()
    qm, dm = torch.ones(Q.size()[:2]), torch.ones(D.size()[:2])
    exp = check(Q, D, qm, dm)
    assert exp.tolist() == [[21.0, 41.0]]
    save("kat_test_score", Q=Q, D=D, q_mask=qm, d_mask=dm, expected=exp)

    # (2) zero-floor pair: a masked doc token floors the max at 0; unmasked -> -1
    Q = torch.tensor([[[1.0, 0.0]]])
    D = torch.tensor([[[-1.0, 0.0], [-2.0, 0.0]]])
    qm = torch.ones(1, 1, dtype=torch.long)
    e_full = check(Q, D, qm, torch.tensor([[1, 1]]))
    e_floor = check(Q, D, qm, torch.tensor([[1, 0]]))
    assert e_full.item() == -1.0 and e_floor.item() == 0.0
    save("zero_floor", Q=Q, D=D, q_mask=qm, d_mask_full=torch.tensor([[1, 1]]), d_mask_floor=torch.tensor([[1, 0]]),
         expected_full=e_full, expected_floor=e_floor)

    # (3) C1: BASELINE.json configs[0] -- 1 query x 10 docs, 32 x 180 tokens, dim 128, fp32, ones masks, seed 0
    g = torch.Generator().manual_seed(0)
    Q = norm_randn(g, 1, 32, 128)
    D = norm_randn(g, 10, 180, 128)
    qm, dm = torch.ones(1, 32, dtype=torch.long), torch.ones(10, 180, dtype=torch.long)
    save("c1_1q_10d", Q=Q, D=D, q_mask=qm, d_mask=dm, expected=check(Q, D, qm, dm))

    # (4) all-pairs 4 x 6 with random 0/1 masks (masked query tokens included), Ld = 70 (not a multiple of 32)
    g = torch.Generator().manual_seed(4)
    Q = norm_randn(g, 4, 32, 128)
    D = norm_randn(g, 6, 70, 128)
    qm = (torch.rand(4, 32, generator=g) > 0.25).long()
    dm = (torch.rand(6, 70, generator=g) > 0.3).long()
    dm[2] = 1                     # one doc with no masked token (no floor)
    dm[3, 1:] = 0                 # one doc with a single live token
    qm[1] = 0                     # one query fully masked -> score 0 everywhere
    save("allpairs_4x6_masked", Q=Q, D=D, q_mask=qm, d_mask=dm, expected=check(Q, D, qm, dm))
    # float-valued (non 0/1) masks exercise the multiply itself
    qf = torch.rand(4, 32, generator=g)
    df = torch.rand(6, 70, generator=g) - 0.2
    save("allpairs_4x6_floatmask", Q=Q, D=D, q_mask=qf, d_mask=df, expected=check(Q, D, qf, df))

    # (5) ragged rerank through the restated rank_forward with the IMPORTED score as model.score
    g = torch.Generator().manual_seed(5)
    ndocs, dim, Lq = 64, 128, 32
    doclens = torch.randint(1, 101, (ndocs,), generator=g)
    doclens[:6] = torch.tensor([180, 179, 1, 2, 33, 64])
    # strides are kth values at 25/50/75 % and the max; force docs exactly at, one below and one above each
    srt = doclens.sort().values
    p25, p50, p75 = (int(srt[int(p * ndocs / 100.0) - 1]) for p in (25, 50, 75))
    doclens[6:15] = torch.tensor([p25, p25 - 1, p25 + 1, p50, p50 - 1, p50 + 1, p75, p75 - 1, p75 + 1]).clamp(min=1)
    doclens = doclens.tolist()
    half = ndocs // 2
    parts_doclens = [doclens[:half], doclens[half:]]
    parts = [norm_randn(g, sum(dl), dim).half() for dl in parts_doclens]      # encoder.py:175 stores fp16
    ranker = RefRanker(parts, parts_doclens, dim=dim, score_fn=REF)
    q = norm_randn(g, Lq, dim)
    q[:, :] = q - 0.35 * q.mean(0, keepdim=True)   # keep it generic; still unit-ish
    q = F.normalize(q, dim=-1)
    Qr = q.unsqueeze(0).permute(0, 2, 1).contiguous()          # [1, h, Lq] as faiss_indexers.py:232-233 hands it over
    pids = torch.randperm(ndocs, generator=g).tolist()
    all_scores = ranker.all_scores(Qr, pids)
    top_p, top_s = ranker.rank_forward(Qr, pids, depth=10)
    # a "negative query": every real similarity < 0 for some tokens so the 0-floor decides the score
    qneg = -parts[0][:Lq].float()
    qneg = F.normalize(qneg + 0.05 * norm_randn(g, Lq, dim), dim=-1)
    Qn = qneg.unsqueeze(0).permute(0, 2, 1).contiguous()
    all_scores_neg = ranker.all_scores(Qn, pids)
    save("ragged_rerank_64", part0=parts[0], part1=parts[1], doclens0=np.array(parts_doclens[0]),
         doclens1=np.array(parts_doclens[1]), strides=np.array(ranker.strides),
         pad_len=ranker.bucket_strides(list(range(ndocs))), Q=Qr, Q_neg=Qn, pids=np.array(pids),
         expected_scores=all_scores, expected_scores_neg=all_scores_neg,
         top10_pids=np.array(top_p), top10_scores=np.array(top_s, dtype=np.float64))

    # (6) C4 multi-view: 8 viewer tokens per doc and per query (dense.yaml q_view/d_view), no masks
    g = torch.Generator().manual_seed(6)
    Q = norm_randn(g, 2, 8, 128)
    D = norm_randn(g, 32, 8, 128)
    qm, dm = torch.ones(2, 8, dtype=torch.long), torch.ones(32, 8, dtype=torch.long)
    save("c4_multiview", Q=Q, D=D, q_mask=qm, d_mask=dm, expected=check(Q, D, qm, dm))

    # (7) C5: bf16-rounded inputs, dim 768, scored by the reference in fp32 on the rounded values
    g = torch.Generator().manual_seed(7)
    Q = norm_randn(g, 1, 32, 768).bfloat16()
    D = norm_randn(g, 3, 200, 768).bfloat16()
    qm, dm = torch.ones(1, 32, dtype=torch.long), torch.ones(3, 200, dtype=torch.long)
    save("c5_bf16_768", Q=Q, D=D, q_mask=qm, d_mask=dm, expected=check(Q.float(), D.float(), qm, dm))

    # (8) the reference's DEFAULT deployment shape (proj_conf/dense.yaml:6-8: dim 768; encoder.py:175: fp16 index) as a
    #     ragged rerank: 16 docs of 1..48 tokens incl. docs exactly at / next to every percentile stride, one query and
    #     one "negative" query whose score the 0-floor decides -- same construction as (5), dim 768
    g = torch.Generator().manual_seed(8)
    ndocs, dim, Lq = 16, 768, 32
    doclens = torch.randint(1, 41, (ndocs,), generator=g)
    doclens[:3] = torch.tensor([48, 47, 1])
    srt = doclens.sort().values
    p25, p50, p75 = (int(srt[int(p * ndocs / 100.0) - 1]) for p in (25, 50, 75))
    doclens[3:9] = torch.tensor([p25, p25 + 1, p50, p50 - 1, p75, p75 + 1]).clamp(min=1)
    doclens = doclens.tolist()
    parts_doclens = [doclens[:7], doclens[7:]]
    parts = [norm_randn(g, sum(dl), dim).half() for dl in parts_doclens]
    ranker = RefRanker(parts, parts_doclens, dim=dim, score_fn=REF)
    q = norm_randn(g, Lq, dim)
    Qr = q.unsqueeze(0).permute(0, 2, 1).contiguous()
    pids = torch.randperm(ndocs, generator=g).tolist()
    all_scores = ranker.all_scores(Qr, pids)
    top_p, top_s = ranker.rank_forward(Qr, pids, depth=10)
    qneg = F.normalize(-parts[1][:Lq].float() + 0.05 * norm_randn(g, Lq, dim), dim=-1)
    Qn = qneg.unsqueeze(0).permute(0, 2, 1).contiguous()
    all_scores_neg = ranker.all_scores(Qn, pids)
    save("ragged_rerank_768", part0=parts[0], part1=parts[1], doclens0=np.array(parts_doclens[0]),
         doclens1=np.array(parts_doclens[1]), strides=np.array(ranker.strides),
         pad_len=ranker.bucket_strides(list(range(ndocs))), Q=Qr, Q_neg=Qn, pids=np.array(pids),
         expected_scores=all_scores, expected_scores_neg=all_scores_neg,
         top10_pids=np.array(top_p), top10_scores=np.array(top_s, dtype=np.float64))

    # (9) the batched driver's masked-token compaction: the reference drops a query's masked tokens with
    #     training_utils.keep_nonzero before search() (dense_server_client.py:45 via qd_mask_to_realinput(keep_dim=False),
    #     training_utils.py:48-53,84-93) -- both imported here -- and scores what is left.  Holes in the middle of the
    #     sequence (the tokenizer zeroes punctuation and [SEP]), a ragged rerank behind it: 12 docs of 1..40 tokens, fp16
    #     index, three queries with different masks (one with a single live token).
    from colbert.training.training_utils import keep_nonzero as REF_KEEP, qd_mask_to_realinput as REF_REALINPUT
    from oracle.maxsim_oracle import keep_nonzero as my_keep
    g = torch.Generator().manual_seed(9)
    ndocs, dim, Lq = 12, 128, 32
    doclens = torch.randint(1, 41, (ndocs,), generator=g).tolist()
    parts_doclens = [doclens[:5], doclens[5:]]
    parts = [norm_randn(g, sum(dl), dim).half() for dl in parts_doclens]
    ranker = RefRanker(parts, parts_doclens, dim=dim, score_fn=REF)
    Qs = norm_randn(g, 3, Lq, dim)
    masks = (torch.rand(3, Lq, generator=g) < 0.7).long()
    masks[:, 0] = 1
    masks[1, 20:] = 0                      # a padded tail as well
    masks[2] = 0
    masks[2, 5] = 1                        # one live token, not the first
    pids = torch.randperm(ndocs, generator=g).tolist()
    exp = []
    for q in range(3):
        real_q, real_m = REF_REALINPUT(t=Qs[q], t_mask=masks[q], max_length=Lq, keep_dim=False)
        rq2, rm2 = REF_KEEP(Qs[q], masks[q])
        mq, mm = my_keep(Qs[q], masks[q])
        assert torch.equal(real_q, rq2) and torch.equal(real_q, mq) and torch.equal(real_m, mm), "oracle keep_nonzero != reference"
        assert real_q.size(0) == int(masks[q].sum())
        exp.append(ranker.all_scores(real_q.unsqueeze(0).permute(0, 2, 1).contiguous(), pids))    # faiss_indexers.py:232-234
    save("masked_query_rerank", part0=parts[0], part1=parts[1], doclens0=np.array(parts_doclens[0]),
         doclens1=np.array(parts_doclens[1]), Q=Qs, q_word_mask=masks, pids=np.array(pids),
         expected_scores=torch.stack(exp))

    # dtype propagation facts: fp32*int64 -> fp32 ; fp16*int64 -> fp16
    a = REF(torch.ones(1, 2, 4), torch.ones(1, 2, 4), torch.ones(1, 2, dtype=torch.long), torch.ones(1, 2, dtype=torch.long))
    assert a.dtype == torch.float32
    b = REF(torch.ones(1, 2, 4).half(), torch.ones(1, 2, 4).half(), torch.ones(1, 2, dtype=torch.long),
            torch.ones(1, 2, dtype=torch.long))
    assert b.dtype == torch.float16
    print("ok")


if __name__ == "__main__":
    main()
