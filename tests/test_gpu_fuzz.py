"""GPU: seeded random launches across the dispatch boundaries of the rerank path -- one query to hundreds, one candidate
to thousands per row, Lq 1..40, narrow and wide embeddings, every index dtype, ragged / uniform / very short docs,
padding slots, dropped query tokens -- checked three ways:
  * a sample of (query, slot) entries against the oracle's float64 closed form (oracle/maxsim_oracle.py
    ragged_scores_f64: the reference's bucket / pad / mask / max / sum, colbert_ranker.py:88-112 + BaseModel.py:41-45,
    restated per candidate); tolerance as everywhere: fp32 queries |d| <= 1e-4, 16-bit inputs |d| <= 1e-3;
  * EVERY entry against the same rows scored through the other launch forms, bit for bit: counted rows (the device-built
    work list), one query per launch (the small / split forms), and the list with its rows permuted;
  * padding slots are -inf, empty docs 0, and the top-k of a row is the sorted head of its scores.
The launch forms are chosen by the library from the shape; what this file varies is the shape."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

ATOL32 = 1e-4
ATOL16 = 1e-3
SEED = int(os.environ.get("MAXSIM_FUZZ_SEED", "0"))     # (longer hunts: other cases AND other data than the suite's)


@pytest.fixture(scope="module")
def ca():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    import colbert_amd
    return colbert_amd


def _cases():
    rng = np.random.RandomState(20261004 + SEED)
    out = []
    for i in range(int(os.environ.get("MAXSIM_FUZZ_CASES", "48"))):     # (more cases: a longer hunt, same seeds first)
        h = int(rng.choice([128, 128, 128, 128, 768, 768, 256, 384, 64, 96]))
        dtype = ["fp32", "fp16", "bf16"][int(rng.randint(3))]
        lq = int(rng.choice([1, 5, 8, 16, 17, 32, 32, 32, 40])) if h == 128 else int(rng.choice([1, 8, 16, 32, 32]))
        docs = str(rng.choice(["ragged", "ragged", "uniform180", "uniform8", "uniform16", "short", "long", "holes"]))
        nq, ncand = [(1, 1000), (1, 37), (2, 1000), (5, 300), (64, 1000), (300, 125), (40, 2500), (3, 1)][int(rng.randint(8))]
        if h >= 256 and docs in ("uniform180", "long", "ragged"):
            nq, ncand = min(nq, 64), min(ncand, 1000)
        if docs == "uniform8" and i % 3 == 2:      # (round 5: 4-token docs too, without disturbing the sequence of the earlier cases)
            docs = "uniform4"
        out.append(dict(i=i, h=h, dtype=dtype, lq=lq, docs=docs, nq=nq, ncand=ncand, q16=bool(rng.rand() < 0.25),
                        qdrop=str(rng.choice(["none", "none", "len", "mask"])), pad=bool(rng.rand() < 0.5),
                        mode=str(rng.choice(["exact", "exact", "fast", "bf16x3"]))))
    return out


def _doclens(kind, n, gen):
    if kind == "ragged":
        d = (torch.randn(n, generator=gen) * 40 + 120).round().clamp(8, 180)
    elif kind == "uniform180":
        d = torch.full((n,), 180.0)
    elif kind == "uniform8":
        d = torch.full((n,), 8.0)
    elif kind == "uniform4":
        d = torch.full((n,), 4.0)
    elif kind == "uniform16":
        d = torch.full((n,), 16.0)
    elif kind == "short":
        d = torch.randint(1, 24, (n,), generator=gen).float()
    elif kind == "long":
        d = torch.randint(150, 384, (n,), generator=gen).float()
    else:  # holes: empty docs between ragged ones
        d = torch.randint(0, 90, (n,), generator=gen).float()
        d[::5] = 0
    d = d.long().tolist()
    if sum(d) == 0:
        d[0] = 3
    return d


@pytest.mark.parametrize("c", _cases(), ids=lambda c: f"{c['i']}-h{c['h']}-{c['dtype']}-Lq{c['lq']}-{c['docs']}-{c['nq']}x{c['ncand']}")
def test_random_launch_shapes(ca, c):
    from oracle.maxsim_oracle import ragged_scores_f64
    gen = torch.Generator().manual_seed(7000 + c["i"] + 100003 * SEED)
    h, lq, nq, ncand = c["h"], c["lq"], c["nq"], c["ncand"]
    tdt = {"fp32": torch.float32, "fp16": torch.float16, "bf16": torch.bfloat16}[c["dtype"]]
    ndocs = 1500 if h <= 128 else 400
    doclens = _doclens(c["docs"], ndocs, gen)
    emb = F.normalize(torch.randn(sum(doclens), h, generator=gen), dim=-1).to(tdt)
    kw = dict(fp32_mode=c["mode"]) if (tdt == torch.float32 and h == 128) else {}
    r = ca.ColbertRanker(parts=[emb], parts_doclens=[doclens], dim=h, index_dtype=tdt, **kw)
    q16 = c["q16"] and tdt != torch.float32
    Q = F.normalize(torch.randn(nq, lq, h, generator=gen), dim=-1)
    if q16:
        Q = Q.to(tdt)
    # rows: live pids first (duplicates allowed, as an ANN list never has them but the interface allows), -1 behind
    counts = torch.randint(0 if nq > 1 else 1, ncand + 1, (nq,), generator=gen) if c["pad"] else torch.full((nq,), ncand)
    counts[0] = max(int(counts[0]), 1)
    cand = torch.full((nq, ncand), -1, dtype=torch.int64)
    for q in range(nq):
        k = int(counts[q])
        cand[q, :k] = torch.randint(0, ndocs, (k,), generator=gen)
    q_len = q_mask = None
    live_tok = [list(range(lq)) for _ in range(nq)]
    if c["qdrop"] == "len":
        q_len = torch.randint(1, lq + 1, (nq,), generator=gen).int()
        live_tok = [list(range(int(q_len[q]))) for q in range(nq)]
    elif c["qdrop"] == "mask":
        q_mask = (torch.rand(nq, lq, generator=gen) < 0.8).long()
        q_mask[:, 0] = 1
        live_tok = [q_mask[q].nonzero().flatten().tolist() for q in range(nq)]
    dc, dcnt = cand.cuda(), counts.int().cuda()
    sc = r.score_candidates(Q, dc, q_len=q_len, q_mask=q_mask)
    got = sc.cpu()

    # ---- the other launch forms, bit for bit ------------------------------------------------------------------
    counted = r.score_candidates(Q, dc, q_len=q_len, q_mask=q_mask, cand_count=dcnt).cpu()
    live = torch.arange(ncand)[None, :] < counts[:, None]
    assert torch.equal(counted[live], got[live])
    assert bool((counted[~live] == float("-inf")).all()) and bool((got[~live] == float("-inf")).all())
    for q in sorted(set([0, nq - 1, nq // 2])):            # one query per launch: the small-launch forms
        one = r.score_candidates(Q[q:q + 1], dc[q:q + 1], q_len=None if q_len is None else q_len[q:q + 1],
                                 q_mask=None if q_mask is None else q_mask[q:q + 1]).cpu()
        assert torch.equal(one[0], got[q]), (q, float((one[0] - got[q]).abs().nan_to_num(0, 0, 0).max()))
    perm = torch.randperm(ncand, generator=gen)
    shuffled = r.score_candidates(Q, dc[:, perm.cuda()], q_len=q_len, q_mask=q_mask).cpu()
    assert torch.equal(shuffled, got[:, perm])               # a candidate's score does not depend on its slot
    if r._iv.uniform_len:                                    # a fixed-length index: the general kernels (promise withdrawn) agree
        import copy
        import ctypes
        g = copy.copy(r)
        g._iv = r._index_view()
        g._iv.uniform_len = 0
        g._iv_ref, g._iv_addr = ctypes.byref(g._iv), ctypes.addressof(g._iv)
        assert torch.equal(g.score_candidates(Q, dc, q_len=q_len, q_mask=q_mask).cpu(), got)
        assert torch.equal(g.score_candidates(Q, dc, q_len=q_len, q_mask=q_mask, cand_count=dcnt).cpu()[live], got[live])

    # ---- a sample of entries against the oracle ---------------------------------------------------------------
    atol = ATOL16 if (q16 or (tdt == torch.bfloat16 and h != 128)) else ATOL32
    rows = torch.randint(0, nq, (160,), generator=gen).tolist()
    pad_len = r.d_pad_len.cpu()
    checked = 0
    for q in rows:
        k = int(counts[q])
        if k == 0:
            continue
        j = int(torch.randint(0, k, (1,), generator=gen))
        pid = int(cand[q, j])
        if doclens[pid] == 0:
            assert got[q, j] == 0.0
            checked += 1
            continue
        Qq = Q[q].float()[live_tok[q]]
        exp = ragged_scores_f64(emb, r.doclens, r.doclens_pfxsum, pad_len, Qq, [pid])[0]
        assert abs(float(got[q, j]) - exp) <= atol, (q, j, pid, float(got[q, j]), exp)
        checked += 1
    assert checked > 0

    # ---- top-k of the rows ------------------------------------------------------------------------------------
    k = min(100, ncand)
    tp, ts = r.topk(sc, dc, k)
    tpc, tsc = r.topk(sc, dc, k, counts=dcnt)
    ts, tp, tsc, tpc = ts.cpu(), tp.cpu(), tsc.cpu(), tpc.cpu()
    exp_s = torch.sort(got, dim=1, descending=True).values[:, :k]
    assert torch.equal(ts, exp_s)
    for q in range(nq):
        n = min(int(counts[q]), k)
        assert torch.equal(tsc[q, :n], exp_s[q, :n])
        # every returned (pid, score) pair is a pair of the row
        pairs = set(zip(cand[q].tolist(), got[q].tolist()))
        assert all((int(p), float(s)) in pairs for p, s in zip(tpc[q, :n].tolist(), tsc[q, :n].tolist()))


# ------------------------------------------------------------------------------------------------------
# the operator itself: BaseModel.score (BaseModel.py:39-46), all-pairs, with masks, under random shapes / dtypes
# ------------------------------------------------------------------------------------------------------
def _dense_cases():
    rng = np.random.RandomState(4242 + SEED)
    out = []
    for i in range(int(os.environ.get("MAXSIM_FUZZ_CASES", "48"))):
        h = int(rng.choice([1, 3, 16, 24, 64, 96, 128, 128, 200, 256, 384, 768, 1024, 1100]))
        out.append(dict(i=i, h=h, nq=int(rng.choice([1, 2, 3, 7, 33])), nd=int(rng.choice([1, 2, 5, 19, 64])),
                        lq=int(rng.choice([1, 2, 8, 16, 31, 32, 33, 40, 64])), ld=int(rng.choice([1, 2, 7, 8, 31, 32, 33, 100, 180, 257, 384])),
                        dtype=str(rng.choice(["fp32", "fp32", "fp16", "bf16"])),
                        mask=str(rng.choice(["int64", "float32", "bool", "int32", "float16", "weights"])),
                        grad=bool(rng.rand() < 0.3)))
    return out


@pytest.mark.parametrize("c", _dense_cases(), ids=lambda c: f"{c['i']}-{c['nq']}x{c['nd']}-{c['lq']}x{c['ld']}-h{c['h']}-{c['dtype']}-{c['mask']}{'-grad' if c['grad'] else ''}")
def test_random_dense_score(ca, c):
    """ca.score against the oracle's four torch ops (oracle.maxsim_oracle.ref_score == the imported BaseModel.score,
    tests/golden/make_golden.py) on the same tensors: result dtype as torch promotes it, values within the stated
    tolerance (fp32 operands 1e-4; 16-bit operands: the oracle evaluated in fp32 on the rounded operands, 1e-3 + the
    result's own 16-bit rounding), and -- where asked -- gradients against torch autograd through the oracle."""
    from oracle.maxsim_oracle import ref_score
    gen = torch.Generator().manual_seed(9000 + c["i"] + 100003 * SEED)
    tdt = {"fp32": torch.float32, "fp16": torch.float16, "bf16": torch.bfloat16}[c["dtype"]]
    nq, nd, lq, ld, h = c["nq"], c["nd"], c["lq"], c["ld"], c["h"]
    Q = F.normalize(torch.randn(nq, lq, h, generator=gen), dim=-1).to(tdt)
    D = F.normalize(torch.randn(nd, ld, h, generator=gen), dim=-1).to(tdt)
    if c["mask"] == "weights":                       # arbitrary non-negative float weights, not only 0/1
        qm, dm = torch.rand(nq, lq, generator=gen), torch.rand(nd, ld, generator=gen)
    else:
        mdt = {"int64": torch.int64, "float32": torch.float32, "bool": torch.bool, "int32": torch.int32, "float16": torch.float16}[c["mask"]]
        qm = (torch.rand(nq, lq, generator=gen) > 0.2).to(mdt)
        dm = (torch.rand(nd, ld, generator=gen) > 0.3).to(mdt)
    exp32 = ref_score(Q.float(), D.float(), qm.float(), dm.float())
    exp_dtype = (Q[:1, :1, :1] * qm[:1, :1, None]).dtype          # BaseModel.py:41-42: the promoted type of operand * mask
    use_grad = c["grad"] and tdt == torch.float32 and c["mask"] != "weights"
    Qg, Dg = Q.cuda(), D.cuda()
    if use_grad:
        Qg.requires_grad_(True)
        Dg.requires_grad_(True)
    out = ca.score(Qg, Dg, qm.cuda(), dm.cuda())
    assert tuple(out.shape) == (nq, nd)
    assert out.dtype == exp_dtype, (out.dtype, exp_dtype)
    if tdt == torch.float32 and exp_dtype == torch.float32:
        atol = ATOL32
    else:
        atol = ATOL16 + float(exp32.abs().max()) * (2.0 ** -8 if exp_dtype == torch.bfloat16 else 2.0 ** -11 if exp_dtype == torch.float16 else 0.0)
    torch.testing.assert_close(out.detach().float().cpu(), exp32, rtol=0, atol=atol)
    if use_grad:
        Qr, Dr = Q.clone().requires_grad_(True), D.clone().requires_grad_(True)
        w = torch.randn(nq, nd, generator=gen)
        (ref_score(Qr, Dr, qm, dm) * w).sum().backward()
        (out * w.cuda()).sum().backward()
        # the gradient of a max is not continuous where two doc tokens tie: a (query token, doc) whose two best
        # similarities are closer than fp32 summation order can resolve may legitimately route through either -- the rows
        # such a pair feeds are left out (3 of 3000 cases have one; dim 1 ties everywhere and is left out altogether)
        sim = torch.einsum("qmh,dnh->qdmn", Q * qm[..., None].float(), D * dm[..., None].float())
        top2 = sim.topk(min(2, ld), dim=-1).values
        tie = (top2[..., 0] - top2[..., -1]).abs() < 1e-5 if ld > 1 else torch.zeros(nq, nd, lq, dtype=torch.bool)
        okq = ~tie.any(dim=1)                                    # [nq, lq]: query-token rows fed by no tied pair
        okd = ~tie.any(dim=2).any(dim=0)                         # [nd]: docs none of whose pairs is tied
        if h > 1:
            scale_q, scale_d = float(Qr.grad.abs().max()) + 1e-6, float(Dr.grad.abs().max()) + 1e-6
            torch.testing.assert_close(Qg.grad.cpu()[okq], Qr.grad[okq], rtol=1e-4, atol=1e-5 * max(1.0, scale_q))
            torch.testing.assert_close(Dg.grad.cpu()[okd], Dr.grad[okd], rtol=1e-4, atol=1e-5 * max(1.0, scale_d))


def test_backward_on_rows_wider_than_the_backward_kernels(ca):
    """h > 1024 (found by the sweep above at 600 cases): the backward kernels take rows up to 1024 wide; wider rows go
    through them in slabs of columns (colbert_amd/scoring.py) -- gradients against torch autograd through the oracle."""
    from oracle.maxsim_oracle import ref_score
    gen = torch.Generator().manual_seed(104)
    Q = F.normalize(torch.randn(2, 5, 1100, generator=gen), dim=-1)
    D = F.normalize(torch.randn(3, 40, 1100, generator=gen), dim=-1)
    qm = (torch.rand(2, 5, generator=gen) > 0.2).long()
    dm = (torch.rand(3, 40, generator=gen) > 0.3).long()
    Qg, Dg = Q.cuda().requires_grad_(True), D.cuda().requires_grad_(True)
    Qr, Dr = Q.clone().requires_grad_(True), D.clone().requires_grad_(True)
    w = torch.randn(2, 3, generator=gen)
    out = ca.score(Qg, Dg, qm.cuda(), dm.cuda())
    exp = ref_score(Qr, Dr, qm, dm)
    torch.testing.assert_close(out.detach().cpu(), exp.detach(), rtol=0, atol=ATOL32)
    (out * w.cuda()).sum().backward()
    (exp * w).sum().backward()
    torch.testing.assert_close(Qg.grad.cpu(), Qr.grad, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(Dg.grad.cpu(), Dr.grad, rtol=1e-4, atol=1e-5)


# ------------------------------------------------------------------------------------------------------
# doc-sharded rerank on one GPU: N shards scored one after the other, merged, against ONE whole-index ranker
# ------------------------------------------------------------------------------------------------------
def _shard_cases():
    rng = np.random.RandomState(777 + SEED)
    return [dict(i=i, world=int(rng.choice([2, 3, 5, 8])), nq=int(rng.choice([1, 3, 9, 40])), ncand=int(rng.choice([1, 17, 100, 400])),
                 k=int(rng.choice([1, 10, 100])), dtype=str(rng.choice(["fp32", "fp16"])), docs=str(rng.choice(["ragged", "short", "uniform8"])),
                 skew=bool(rng.rand() < 0.4), qdrop=bool(rng.rand() < 0.3))
            for i in range(max(8, int(os.environ.get("MAXSIM_FUZZ_CASES", "48")) // 3))]


@pytest.mark.parametrize("c", _shard_cases(), ids=lambda c: f"{c['i']}-w{c['world']}-{c['nq']}x{c['ncand']}-k{c['k']}-{c['dtype']}-{c['docs']}{'-skew' if c['skew'] else ''}")
def test_random_sharded_rerank(ca, c):
    """Pid-range shards bucketed by the strides of the WHOLE index, global candidate lists handed to every shard, local
    top-k with global pids, merge: the merged (pid, score) lists against one unsharded ranker over the same docs -- the same
    kernels, so the scores are the same bits; ranks that hold none of a query's candidates contribute padding."""
    from colbert_amd.ranker import reference_strides
    from colbert_amd.sharded import ShardedRanker, merge_gathered, shard_range
    gen = torch.Generator().manual_seed(5000 + c["i"] + 100003 * SEED)
    world, nq, ncand, k = c["world"], c["nq"], c["ncand"], c["k"]
    tdt = torch.float32 if c["dtype"] == "fp32" else torch.float16
    ndocs, h = 900, 128
    doclens = _doclens(c["docs"], ndocs, gen)
    emb = F.normalize(torch.randn(sum(doclens), h, generator=gen), dim=-1).to(tdt)
    whole = ca.ColbertRanker(parts=[emb], parts_doclens=[doclens], dim=h, index_dtype=tdt)
    Q = F.normalize(torch.randn(nq, 32, h, generator=gen), dim=-1)
    hi_pid = ndocs // world if c["skew"] else ndocs            # skew: every candidate lives on the first shard
    cand = torch.stack([torch.randperm(hi_pid, generator=gen)[:ncand] if ncand <= hi_pid else torch.randint(0, hi_pid, (ncand,), generator=gen)
                        for _ in range(nq)])
    if c["i"] % 3 == 0 and ncand > 2:                          # padding slots anywhere in the global lists (-1: FAISS / a short list)
        cand[torch.rand(nq, ncand, generator=gen) < 0.25] = -1
    q_len = torch.randint(1, 33, (nq,), generator=gen).int() if c["qdrop"] else None
    kk = min(k, ncand)
    ref_scores = whole.score_candidates(Q, cand.cuda(), q_len=q_len)
    exp_s = torch.sort(ref_scores.cpu(), dim=1, descending=True).values[:, :kk]
    offs = np.concatenate([[0], np.cumsum(doclens)])
    gstr = reference_strides(torch.tensor(doclens))
    tops = []
    for rank in range(world):
        lo, hi = shard_range(ndocs, rank, world)
        r = ca.ColbertRanker(parts=[emb[offs[lo]:offs[hi]]], parts_doclens=[doclens[lo:hi]], dim=h, index_dtype=tdt, strides=gstr)
        tops.append(ShardedRanker(r, lo, hi, sync_strides=False).local_topk(Q, cand.cuda(), kk, q_len=q_len))
    gs, gp = torch.stack([t[1] for t in tops]), torch.stack([t[0] for t in tops])
    mp, ms = merge_gathered(gs, gp, kk, whole.topk)
    assert torch.equal(ms.cpu(), exp_s)                         # the same bits as the unsharded ranker
    look = [dict(zip(cand[q].tolist(), ref_scores[q].cpu().tolist())) for q in range(nq)]
    for q in range(nq):
        got = mp[q].cpu().tolist()
        real = [p for p in got if p >= 0]
        assert len(set(real)) == len(real) or ncand > hi_pid     # distinct candidates stay distinct
        assert all(look[q][p] == s for p, s in zip(got, ms[q].cpu().tolist()))


# ------------------------------------------------------------------------------------------------------
# the online call, as the reference's caller makes it and in the other forms its signature admits
# ------------------------------------------------------------------------------------------------------
def _rf_cases():
    rng = np.random.RandomState(31337 + SEED)
    return [dict(i=i, h=int(rng.choice([128, 128, 768, 64])), dtype=str(rng.choice(["fp32", "fp16", "bf16"])),
                 lq=int(rng.choice([1, 8, 32, 32])), n=int(rng.choice([1, 2, 9, 100, 1000, 2500])), depth=int(rng.choice([1, 10, 100, 5000])),
                 form=str(rng.choice(["list", "list", "tensor", "neg", "dup", "cpuQ", "halfQ"])), docs=str(rng.choice(["ragged", "short", "uniform180", "uniform8", "holes"])))
            for i in range(max(8, int(os.environ.get("MAXSIM_FUZZ_CASES", "48")) // 2))]


@pytest.mark.parametrize("c", _rf_cases(), ids=lambda c: f"{c['i']}-h{c['h']}-{c['dtype']}-Lq{c['lq']}-n{c['n']}-d{c['depth']}-{c['form']}-{c['docs']}")
def test_random_rank_forward(ca, c):
    """rank_forward(Q[1,h,Lq], pids, depth) -> (list[int], list[float]) (colbert_ranker.py:75-137) against the oracle's
    restatement of it on the same index: lists and LongTensors, negative pids (torch indexing wraps them), duplicates,
    depth beyond the list, Q handed over on the CPU or in fp16 (':78' moves and widens it)."""
    from oracle.maxsim_oracle import RefRanker
    gen = torch.Generator().manual_seed(8000 + c["i"] + 100003 * SEED)
    tdt = {"fp32": torch.float32, "fp16": torch.float16, "bf16": torch.bfloat16}[c["dtype"]]
    h, lq, n, depth = c["h"], c["lq"], c["n"], c["depth"]
    ndocs = 600 if h <= 128 else 200
    doclens = _doclens(c["docs"], ndocs, gen)
    if c["docs"] == "holes":
        doclens = [max(d, 1) for d in doclens]                 # (the reference cannot gather a 0-length doc from a view)
    half = ndocs // 2
    pdl = [doclens[:half], doclens[half:]]
    parts = [F.normalize(torch.randn(sum(d), h, generator=gen), dim=-1).to(tdt) for d in pdl]
    ref = RefRanker(parts, pdl, dim=h, index_dtype=tdt)
    r = ca.ColbertRanker(parts=parts, parts_doclens=pdl, dim=h, index_dtype=tdt)
    q = F.normalize(torch.randn(lq, h, generator=gen), dim=-1)
    if c["form"] == "dup":
        pids = torch.randint(0, ndocs, (n,), generator=gen).tolist()
    else:
        pids = (torch.randperm(ndocs, generator=gen)[:n] if n <= ndocs else torch.randint(0, ndocs, (n,), generator=gen)).tolist()
    ref_pids = list(pids)
    if c["form"] == "neg":                                    # a negative pid p stands for doc ndocs + p (torch indexing)
        pids = [p - ndocs if i % 3 == 0 else p for i, p in enumerate(pids)]
    Qr = q.unsqueeze(0).permute(0, 2, 1)                      # [1, h, Lq] as faiss_indexers.py:232-233 hands it over
    Qin = Qr.cuda()
    if c["form"] == "cpuQ":
        Qin = Qr.clone()
    elif c["form"] == "halfQ":
        Qin = Qr.cuda().half()
        Qr = Qin.float().cpu()
    arg = torch.tensor(pids) if c["form"] == "tensor" else pids
    got_p, got_s = r.rank_forward(Qin, arg, depth=depth)
    exp_scores = ref.all_scores(Qr.contiguous(), ref_pids)
    k = min(depth, len(pids))
    assert isinstance(got_p, list) and isinstance(got_s, list) and len(got_p) == k and len(got_s) == k
    atol = ATOL16 if (tdt == torch.bfloat16 and h != 128) else ATOL32
    es = torch.sort(exp_scores, descending=True).values[:k]
    np.testing.assert_allclose(np.array(got_s), es.numpy(), rtol=0, atol=atol)
    assert all(a >= b for a, b in zip(got_s, got_s[1:]))      # sorted by score, descending
    by_pid = {}
    for p, s in zip(pids, exp_scores.tolist()):
        by_pid.setdefault(p, []).append(s)
    for p, s in zip(got_p, got_s):                            # every returned pid is the caller's own value with ITS score
        assert p in by_pid and min(abs(s - e) for e in by_pid[p]) <= atol, (p, s)


# ------------------------------------------------------------------------------------------------------
# the batched driver: ids of the ANN search -> distinct pids -> counted rerank -> counted top-k
# ------------------------------------------------------------------------------------------------------
def _retrieve_cases():
    rng = np.random.RandomState(2718 + SEED)
    return [dict(i=i, h=int(rng.choice([128, 128, 768])), dtype=str(rng.choice(["fp32", "fp16"])), bs=int(rng.choice([1, 2, 7, 20])),
                 lq=int(rng.choice([4, 16, 32])), depth=int(rng.choice([1, 4, 16, 64])), topk=int(rng.choice([1, 10, 100])),
                 docs=str(rng.choice(["ragged", "short", "uniform8", "uniform180"])), pad=bool(rng.rand() < 0.4))
            for i in range(max(6, int(os.environ.get("MAXSIM_FUZZ_CASES", "48")) // 4))]


@pytest.mark.parametrize("c", _retrieve_cases(), ids=lambda c: f"{c['i']}-h{c['h']}-{c['dtype']}-bs{c['bs']}-Lq{c['lq']}-fd{c['depth']}-k{c['topk']}-{c['docs']}{'-pad' if c['pad'] else ''}")
def test_random_retrieve_batch(ca, c):
    """colbert_amd.retrieve_batch against the reference's per-query loop restated on the oracle (dense_server_client.py:44-48
    -> faiss_indexers.py:224-235 -> colbert_ranker.py:176-229, 75-137): keep_nonzero, the neighbours' token rows -> pids
    through emb2pid + set(), rank_forward.  The ANN search is a stand-in (random token rows, some -1 as FAISS pads)."""
    from oracle.maxsim_oracle import RefRanker, keep_nonzero
    gen = torch.Generator().manual_seed(6000 + c["i"] + 100003 * SEED)
    tdt = torch.float32 if c["dtype"] == "fp32" else torch.float16
    h, bs, lq, depth, topk = c["h"], c["bs"], c["lq"], c["depth"], c["topk"]
    ndocs = 500 if h == 128 else 150
    doclens = [max(d, 1) for d in _doclens(c["docs"], ndocs, gen)]
    parts = [F.normalize(torch.randn(sum(doclens), h, generator=gen), dim=-1).to(tdt)]
    ref = RefRanker(parts, [doclens], dim=h, index_dtype=tdt)
    r = ca.ColbertRanker(parts=parts, parts_doclens=[doclens], dim=h, index_dtype=tdt)
    Q = F.normalize(torch.randn(bs, lq, h, generator=gen), dim=-1)
    mask = (torch.rand(bs, lq, generator=gen) < 0.75).long()
    mask[:, 0] = 1
    ntok = sum(doclens)
    ids = torch.randint(0, ntok, (bs, lq, depth), generator=gen)
    if c["pad"]:
        ids[torch.rand(bs, lq, depth, generator=gen) < 0.2] = -1
        ids[:, 0, 0] = torch.randint(0, ntok, (bs,), generator=gen)      # (at least one neighbour per query)
    emb2pid = torch.repeat_interleave(torch.arange(ndocs), torch.tensor(doclens))            # colbert_ranker.py:163-174
    out = ca.retrieve_batch(r, Q, mask, topk=topk, embedding_ids=ids)
    assert len(out) == bs
    atol = ATOL32
    for qi in range(bs):
        q_live, _ = keep_nonzero(Q[qi], mask[qi])                                              # dense_server_client.py:45
        live_ids = ids[qi][mask[qi].bool()].reshape(-1)
        live_ids = live_ids[live_ids >= 0]
        pids = sorted(set(emb2pid[live_ids].tolist()))                                         # :212-229
        ep, es = ref.rank_forward(q_live.unsqueeze(0).permute(0, 2, 1), pids, depth=topk)     # faiss_indexers.py:232-234
        gp, gs = out[qi]
        assert len(gp) == len(ep) == len(gs)
        np.testing.assert_allclose(np.array(gs), np.array(es), rtol=0, atol=atol)
        all_s = dict(zip(pids, ref.all_scores(q_live.unsqueeze(0).permute(0, 2, 1), pids).tolist()))
        assert len(set(gp)) == len(gp) and all(abs(all_s[p] - s) <= atol for p, s in zip(gp, gs))


# ------------------------------------------------------------------------------------------------------
# the doc-sharded batched driver (round 4): GLOBAL token rows -> every shard's rows -> distinct pids -> counted rerank ->
# counted local top-k -> merge, against the unsharded driver on the same ids
# ------------------------------------------------------------------------------------------------------
def _shard_retrieve_cases():
    rng = np.random.RandomState(4242 + SEED)
    return [dict(i=i, world=int(rng.choice([2, 3, 5, 8])), bs=int(rng.choice([1, 3, 11])), lq=int(rng.choice([4, 16, 32])),
                 depth=int(rng.choice([1, 8, 64, 512])), topk=int(rng.choice([1, 10, 100])), dtype=str(rng.choice(["fp32", "fp16"])),
                 docs=str(rng.choice(["ragged", "short", "uniform8"])), pad=bool(rng.rand() < 0.4), skew=bool(rng.rand() < 0.3))
            for i in range(max(6, int(os.environ.get("MAXSIM_FUZZ_CASES", "48")) // 4))]


@pytest.mark.parametrize("c", _shard_retrieve_cases(), ids=lambda c: f"{c['i']}-w{c['world']}-bs{c['bs']}-Lq{c['lq']}-fd{c['depth']}-k{c['topk']}-{c['dtype']}-{c['docs']}{'-pad' if c['pad'] else ''}{'-skew' if c['skew'] else ''}")
def test_random_sharded_retrieve(ca, c):
    """ShardedRanker.local_retrieve_topk on 2-8 pid-range shards (one after another on this GPU), merged, equals
    colbert_amd.retrieve_batch on the whole index: same score lists bit for bit, same pids up to ties, short rows padded
    with (-1, -inf).  The ANN result is a stand-in: random GLOBAL token rows, some -1, optionally all inside one shard."""
    from colbert_amd.ranker import reference_strides
    from colbert_amd.retriever import prepare_embedding_ids
    from colbert_amd.sharded import ShardedRanker, merge_gathered, shard_range
    gen = torch.Generator().manual_seed(8000 + c["i"] + 100003 * SEED)
    world, bs, lq, depth, topk = c["world"], c["bs"], c["lq"], c["depth"], c["topk"]
    tdt = torch.float32 if c["dtype"] == "fp32" else torch.float16
    ndocs, h = 700, 128
    doclens = [max(d, 1) for d in _doclens(c["docs"], ndocs, gen)]
    emb = F.normalize(torch.randn(sum(doclens), h, generator=gen), dim=-1).to(tdt)
    whole = ca.ColbertRanker(parts=[emb], parts_doclens=[doclens], dim=h, index_dtype=tdt)
    offs = np.concatenate([[0], np.cumsum(doclens)])
    ntok = int(offs[-1])
    Q = F.normalize(torch.randn(bs, lq, h, generator=gen), dim=-1).cuda()
    keep = (torch.rand(bs, lq, generator=gen) < 0.75).long()
    keep[:, 0] = 1
    hi_row = int(offs[shard_range(ndocs, 0, world)[1]]) if c["skew"] else ntok      # skew: every neighbour lives on shard 0
    ids = torch.randint(0, hi_row, (bs, lq, depth), generator=gen)
    if c["pad"]:
        ids[torch.rand(bs, lq, depth, generator=gen) < 0.2] = -1
    ids, keep = ids.cuda(), keep.cuda()
    exp = ca.retrieve_batch(whole, Q, keep, topk=topk, embedding_ids=ids)
    k = min(topk, lq * depth)
    gstr = reference_strides(torch.tensor(doclens))
    keep_b, ids_b = prepare_embedding_ids(Q.device, Q, keep, ids, mask_ids=False)
    tops = []
    for rank in range(world):
        lo, hi = shard_range(ndocs, rank, world)
        r = ca.ColbertRanker(parts=[emb[offs[lo]:offs[hi]]], parts_doclens=[doclens[lo:hi]], dim=h, index_dtype=tdt, strides=gstr)
        sh = ShardedRanker(r, lo, hi, sync_strides=False, n_docs_total=ndocs, tok_lo=int(offs[lo]))
        tops.append(sh.local_retrieve_topk(Q, keep_b, ids_b.reshape(bs, -1), k))
    gs, gp = torch.stack([t[1] for t in tops]), torch.stack([t[0] for t in tops])
    mp, ms = merge_gathered(gs, gp, k, whole.topk)
    for qi, (ep, es) in enumerate(exp):
        n = len(ep)
        assert ms[qi, :n].tolist() == es, (qi, n)
        assert sorted(mp[qi, :n].tolist()) == sorted(ep)
        by = dict(zip(ep, es))
        assert all(by[p] == s for p, s in zip(mp[qi, :n].tolist(), ms[qi, :n].tolist()))
        assert bool((mp[qi, n:] == -1).all()) and bool(torch.isinf(ms[qi, n:]).all())
