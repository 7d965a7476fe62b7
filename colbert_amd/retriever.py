"""Batched retrieve driver: the inner loop of ``DenseRetrieverServer.retrieve``
(reference: colbert/training/dense_server_client.py:44-48) together with ``ColbertRetriever.search``
(colbert/indexing/faiss_indexers.py:224-235) for a whole batch of queries at once.

Per query the reference does: ``keep_nonzero`` (drop the tokens ``q_active_padding`` zeroes -- punctuation and [SEP]
sit MID-sequence, colbert/modeling/tokenizers.py:36) -> ANN search of the live tokens -> embedding ids -> distinct pids
(``emb2pid`` + ``set()``, colbert/ranking/colbert_ranker.py:176-229) -> ``rank_forward`` -> ``(pids, scores)``.
Here every step after the ANN search is one launch for the batch: ids -> distinct pids (``maxsim_embedding_ids_to_pids``),
fused rerank of the counted rows with the keep-mask as a per-token predicate (``maxsim_rerank_counted``; nothing is
compacted, nothing read back), counted top-k, and ONE device->host copy.  The ANN search itself is third-party (FAISS) and stays outside: pass its result, or a callable.
"""
import numpy as np
import torch


def prepare_embedding_ids(dev, Q, q_active_padding, embedding_ids=None, ann_search=None, faiss_depth=None, mask_ids=True):
    """The driver's step before the rerank: the keep-mask of ``q_active_padding`` (``keep_nonzero``,
    training_utils.py:48-53, as a predicate) and the ANN neighbours of the LIVE tokens as ``[bs, Lq, faiss_depth]`` int64 on
    ``dev``.  ``mask_ids``: write -1 into the rows of dropped tokens (a copy); the HIP path passes the keep-mask to the
    kernel instead and leaves the caller's ids as they are.  Shared by the single-GPU and the doc-sharded driver."""
    assert Q.dim() == 3 and tuple(q_active_padding.shape) == tuple(Q.shape[:2])
    bs, Lq, _ = Q.shape
    keep = (q_active_padding.to(dev) != 0)
    if embedding_ids is None:
        if ann_search is None or faiss_depth is None:
            raise ValueError("give embedding_ids, or ann_search and faiss_depth")
        live = ann_search(Q.to(dev)[keep], int(faiss_depth))
        live = torch.as_tensor(live).to(device=dev, dtype=torch.int64)
        embedding_ids = torch.full((bs, Lq, live.size(-1)), -1, dtype=torch.int64, device=dev)
        embedding_ids[keep] = live
    else:
        embedding_ids = embedding_ids.to(device=dev, dtype=torch.int64)
        assert embedding_ids.dim() == 3 and tuple(embedding_ids.shape[:2]) == (bs, Lq)
        if mask_ids:
            embedding_ids = embedding_ids.masked_fill(~keep.unsqueeze(-1), -1)  # (out of place: the caller's tensor is borrowed)
    return keep, embedding_ids


def unpack_topk_lists(top_p, top_s, counts):
    """Device top-k (pids [bs,k] int64, scores [bs,k] fp32, live entries per row [bs] int32) -> the reference's per-query
    ``(pids: list[int], scores: list[float])`` with ONE device -> host copy (and one wait): pids, score bits and counts
    travel packed as int32 words."""
    bs, k = top_p.shape
    packed = torch.cat([top_p.contiguous().view(torch.int32), top_s.contiguous().view(torch.int32),
                        counts.view(torch.int32).unsqueeze(1)], dim=1).cpu().numpy()
    host_p = np.ascontiguousarray(packed[:, :2 * k]).view(np.int64)
    host_s = np.ascontiguousarray(packed[:, 2 * k:3 * k]).view(np.float32)
    host_n = packed[:, 3 * k]
    out = []
    for i in range(bs):
        n = min(k, int(host_n[i]))
        out.append((host_p[i, :n].tolist(), host_s[i, :n].tolist()))
    return out


def retrieve_batch(ranker, Q, q_active_padding, topk, embedding_ids=None, ann_search=None, faiss_depth=None):
    """Q [bs, Lq, h] (the encoder's output, dense_server_client.py:43), q_active_padding [bs, Lq] 0/1.

    Candidates come from ONE of
      embedding_ids [bs, Lq, faiss_depth] int64 : the ANN neighbours (token rows of the index) of every query token;
                      rows of dropped tokens are ignored, -1 entries are skipped (FAISS pads with -1)
      ann_search(q_live [n_live, h], faiss_depth) -> [n_live, faiss_depth] int64 : called once with the live tokens of
                      the whole batch, in batch order -- the reference, too, searches live tokens only
                      (``faiss_index.search``, colbert_ranker.py:200)
    Returns the reference's per-query ``pid_scores``: a list of ``(pids: list[int], scores: list[float])`` sorted by score
    descending, at most ``topk`` long (shorter when a query has fewer distinct candidates).
    ``ranker``: a ``ColbertRanker``, or a ``sharded.ShardedRanker`` (token rows are then GLOBAL rows of the whole
    collection and every rank makes the same call: ``ShardedRanker.retrieve_batch``).
    """
    if not hasattr(ranker, "embedding_ids_to_pids"):         # the doc-sharded form
        return ranker.retrieve_batch(Q, q_active_padding, topk, embedding_ids=embedding_ids, ann_search=ann_search,
                                     faiss_depth=faiss_depth)
    dev = ranker.device
    bs = Q.size(0)
    keep, embedding_ids = prepare_embedding_ids(dev, Q, q_active_padding, embedding_ids, ann_search, faiss_depth, mask_ids=False)
    # distinct pids per query as COUNTED rows (live pids first, -1 behind): the rerank builds its launch from the counts
    # on the device (maxsim_rerank_counted), so the row width is never read back to trim it -- no host sync before the
    # one copy of the results
    cand, counts = ranker.embedding_ids_to_pids(embedding_ids, trim=False, keep=keep)        # colbert_ranker.py:178, :212-229
    k = min(int(topk), cand.size(1))
    top_p, top_s = ranker.rerank_batch(Q, cand, depth=k, q_mask=keep, cand_count=counts)      # :75-137 for every query
    return unpack_topk_lists(top_p, top_s, counts)
