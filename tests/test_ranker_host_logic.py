"""CPU: the host-side state of colbert_amd.ColbertRanker (doclens prefix sums, length-bucket strides and the per-doc
pad_len that drives the analytic 0-floor) equals the oracle's restatement of colbert_ranker.py:31-51,88-90.
No kernel is launched: the index is kept on the CPU here, which the product path refuses to score."""
import pytest
import torch

from oracle.maxsim_oracle import RefRanker


@pytest.fixture(scope="module")
def ca():
    import colbert_amd
    return colbert_amd


@pytest.mark.parametrize("seed,ndocs,lo,hi", [(0, 64, 1, 180), (1, 5, 3, 9), (2, 400, 8, 8), (3, 97, 1, 3), (4, 1000, 0, 300)])
def test_strides_and_pad_len_match_reference_rule(ca, seed, ndocs, lo, hi):
    g = torch.Generator().manual_seed(seed)
    doclens = torch.randint(lo, hi + 1, (ndocs,), generator=g).tolist()
    if sum(doclens) == 0:
        doclens[0] = 1
    half = ndocs // 2
    pdl = [doclens[:half], doclens[half:]]
    parts = [torch.randn(sum(d), 8, generator=g).half() for d in pdl]
    ref = RefRanker(parts, pdl, dim=8)
    r = ca.ColbertRanker(parts=parts, parts_doclens=pdl, dim=8, device="cpu")
    assert r.strides == ref.strides
    assert torch.equal(r.doclens, ref.doclens) and torch.equal(r.doclens_pfxsum, ref.doclens_pfxsum)
    assert torch.equal(r.d_pad_len.long(), ref.bucket_strides(list(range(ndocs))))
    assert torch.equal(r.d_offsets, ref.doclens_pfxsum[:-1])
    assert r.num_embeddings == ref.num_embeddings and r.n_docs == ndocs
    # the token matrix is the concatenation of the parts (no +512 tail)
    assert torch.equal(r.tensor, torch.cat(parts))


def test_fewer_than_four_docs_raises_like_the_reference(ca):
    parts = [torch.zeros(3, 4).half()]
    with pytest.raises(Exception):               # kthvalue(int(25 * 3 / 100) = 0), colbert_ranker.py:238-241
        ca.ColbertRanker(parts=parts, parts_doclens=[[1, 1, 1]], dim=4, device="cpu")


def test_cpu_index_is_never_scored(ca):
    parts = [torch.zeros(8, 128).half()]
    r = ca.ColbertRanker(parts=parts, parts_doclens=[[2, 2, 2, 2]], dim=128, device="cpu")
    with pytest.raises(Exception):               # no CPU path: the library only takes device pointers
        r.score_candidates(torch.zeros(1, 32, 128), torch.zeros(1, 4, dtype=torch.long))
