"""CPU oracle (TEST INFRASTRUCTURE, not product code) for the ColBERT MaxSim rerank path.

A restatement, in plain torch/numpy on the CPU, of the two reference functions on the hot path
(paths relative to the reference checkout, wuyaoxuehun/colbert):

* ``ref_score``         <- ``colbert/modeling/BaseModel.py:39-46``  (``BaseModel.score``)
* ``RefRanker``         <- ``colbert/ranking/colbert_ranker.py:16-73``  (index state, strides, views)
* ``RefRanker.rank_forward`` <- ``colbert/ranking/colbert_ranker.py:75-137``
* ``torch_percentile``  <- ``colbert/ranking/colbert_ranker.py:238-241``
* ``keep_nonzero``      <- ``colbert/training/training_utils.py:48-53``

Pinning: ``ref_score`` is checked in ``tests/golden/make_golden.py`` against the imported reference
``BaseModel.score`` (bitwise on this container's torch CPU build) and against the reference's only
known-answer test (``BaseModel.py:70-75`` -> [[21, 41]]).  ``colbert_ranker.py`` itself is NOT importable
here (module-level ``import faiss``; hard-coded "cuda"), so ``RefRanker`` follows it line by line and its
per-bucket scores were produced by the imported ``BaseModel.score`` when the goldens were generated.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this file.
"""
from itertools import accumulate

import numpy as np
import torch


def ref_score(Q, D, q_mask, d_mask, *args, **kwargs):
    """MaxSim, all pairs.  BaseModel.py:39-46.

    Q [q,m,h], D [d,n,h], q_mask [q,m], d_mask [d,n] -> scores [q,d].
    Masked tokens are ZEROED (not -inf): a masked doc token contributes similarity 0 to the max,
    a masked query token contributes 0 to the sum.
    """
    D = D * d_mask[..., None]                       # BaseModel.py:41
    Q = Q * q_mask[..., None]                       # BaseModel.py:42
    simmat = torch.einsum("qmh,dnh->qdmn", Q, D)    # BaseModel.py:43
    scores_match, _ = simmat.max(-1)                # BaseModel.py:44
    return scores_match.sum(-1)                     # BaseModel.py:45


def score_f64(Q, D, q_mask, d_mask):
    """Same arithmetic as ``ref_score`` carried out in float64 numpy (mask multiply done in the INPUT
    dtype first, as the reference does, then widened).  Used to bound the fp32 summation-order error."""
    Qm = (torch.as_tensor(Q) * torch.as_tensor(q_mask)[..., None]).double().numpy()
    Dm = (torch.as_tensor(D) * torch.as_tensor(d_mask)[..., None]).double().numpy()
    sim = np.einsum("qmh,dnh->qdmn", Qm, Dm)
    return sim.max(-1).sum(-1)


def score_chain_f32(Q, D, q_mask, d_mask, k_order=None):
    """fp32 score with the dot product evaluated as a single k-ordered fmaf chain per (m, n) pair --
    the arithmetic an f32-input MFMA performs (one rounding per product-accumulate).  ``k_order`` is the
    permutation of the hidden dimension the chain walks (None = natural).  Pure numpy, small cases only.
    Emulates fmaf through float64 (exact for one fp32 product + fp32 addend, then one rounding)."""
    Qm = (torch.as_tensor(Q) * torch.as_tensor(q_mask)[..., None]).float().numpy()
    Dm = (torch.as_tensor(D) * torch.as_tensor(d_mask)[..., None]).float().numpy()
    nq, m, h = Qm.shape
    nd, n, _ = Dm.shape
    order = range(h) if k_order is None else k_order
    acc = np.zeros((nq, nd, m, n), dtype=np.float32)
    for k in order:
        prod = Qm[:, None, :, None, k].astype(np.float64) * Dm[None, :, None, :, k].astype(np.float64)
        acc = (prod + acc.astype(np.float64)).astype(np.float32)
    return acc.max(-1).astype(np.float32).sum(-1, dtype=np.float32)


def torch_percentile(tensor, p):
    """colbert_ranker.py:238-241 (kthvalue is 1-based: raises for int(p*N/100) == 0, as the reference does)."""
    assert p in range(1, 100 + 1)
    assert tensor.dim() == 1
    return tensor.kthvalue(int(p * tensor.size(0) / 100.0)).values.item()


def keep_nonzero(Q, q_word_mask):
    """training_utils.py:48-53: drop masked query tokens before search()."""
    assert len(Q.size()) == 2
    b = q_word_mask.bool()
    return Q[b], q_word_mask[b]


class RefRanker:
    """CPU restatement of ``ColbertRanker`` (colbert_ranker.py:15-137) built from an in-memory index.

    ``parts`` is a list of [N_i, dim] tensors (what ``load_index_part`` returns per ``{i}.pt``) and
    ``parts_doclens`` the matching list of doclens lists (``doclens.{i}.json``).
    ``score_fn`` is the object's ``model.score`` (default: ``ref_score``).
    """

    def __init__(self, parts, parts_doclens, dim=None, score_fn=ref_score, index_dtype=torch.float16, strides=None):
        self.maxsim_dtype = torch.float32                              # :20
        self.parts_doclens = parts_doclens
        self.doclens = [x for y in parts_doclens for x in y]          # flatten, utils.py:133
        self.num_embeddings = sum(self.doclens)                       # :25
        dim = parts[0].size(-1) if dim is None else dim
        # _load_parts :61-73 -- zeros(num_embeddings + 512, dim) fp16, parts copied in order
        tensor = torch.zeros(self.num_embeddings + 512, dim, dtype=index_dtype)
        offset = 0
        for part, dl in zip(parts, parts_doclens):
            endpos = offset + sum(dl)
            tensor[offset:endpos] = part
            offset = endpos
        self.tensor = tensor
        self.score_fn = score_fn
        self.init_ranker(strides)

    def init_ranker(self, strides=None):                               # :31-43
        # strides (not in the reference): a DOC SHARD of a bigger index is bucketed by the strides of the whole index
        # (tests of the doc-sharded path); None = the reference's own rule below
        self.doclens_pfxsum = [0] + list(accumulate(self.doclens))
        self.doclens = torch.tensor(self.doclens)
        self.doclens_pfxsum = torch.tensor(self.doclens_pfxsum)
        self.dim = self.tensor.size(-1)
        if strides is None:
            self.strides = [torch_percentile(self.doclens, p) for p in [25, 50, 75]]
            self.strides.append(self.doclens.max().item())
            self.strides = sorted(list(set(self.strides)))
        else:
            self.strides = sorted(int(x) for x in strides)
        self.views = self._create_views(self.tensor)

    def _create_views(self, tensor):                                   # :45-51
        views = []
        for stride in self.strides:
            outdim = tensor.size(0) - stride + 1
            views.append(torch.as_strided(tensor, (outdim, stride, self.dim), (self.dim, self.dim, 1)))
        return views

    def bucket_strides(self, pids):
        """Per-pid padded length S_g the reference would gather it at (:88-90)."""
        pids = torch.as_tensor(pids)
        doclens = self.doclens[pids]
        assignments = (doclens.unsqueeze(1) > torch.tensor(self.strides).unsqueeze(0) + 1e-6).sum(-1)
        return torch.tensor(self.strides)[assignments]

    def all_scores(self, Q, pids):
        """Scores in input-pid order (the vector the reference holds at :122 before sorting)."""
        return self._forward(Q, pids)[1]

    def _forward(self, Q, pids):
        assert len(pids) > 0                                           # :76
        assert Q.size(0) in [1, len(pids)]                             # :77
        Q = Q.contiguous().to(dtype=self.maxsim_dtype)                 # :78 (device hop dropped: CPU oracle)
        raw_pids = pids if type(pids) is list else pids.tolist()
        pids = torch.tensor(pids) if type(pids) is list else pids
        doclens, offsets = self.doclens[pids], self.doclens_pfxsum[pids]            # :88
        assignments = (doclens.unsqueeze(1) > torch.tensor(self.strides).unsqueeze(0) + 1e-6).sum(-1)  # :90
        output_pids, output_scores, output_permutation = [], [], []
        one_to_n = torch.arange(len(raw_pids))
        output_D, output_D_mask = [], []
        for group_idx, stride in enumerate(self.strides):              # :96
            locator = (assignments == group_idx)
            if locator.sum() < 1e-5:
                continue
            group_pids, group_doclens, group_offsets = pids[locator], doclens[locator], offsets[locator]
            group_Q = Q if Q.size(0) == 1 else Q[locator]
            D = torch.index_select(self.views[group_idx], 0, group_offsets)          # :105
            D = D.to(dtype=self.maxsim_dtype)                                         # :107
            mask = torch.arange(stride) + 1                                           # :108
            mask = mask.unsqueeze(0) <= group_doclens.unsqueeze(-1)                   # :109
            scores = self.score_fn(group_Q.permute(0, 2, 1), D,
                                   torch.ones((1, group_Q.size(2)), dtype=torch.long),
                                   mask.to(torch.long))[0]                            # :111-112
            output_pids.append(group_pids)
            output_scores.append(scores)
            output_permutation.append(one_to_n[locator])
            output_D.append(D)
            output_D_mask.append(mask)
        output_permutation = torch.cat(output_permutation).sort().indices            # :120
        output_pids = torch.cat(output_pids)[output_permutation].tolist()            # :121
        output_scores = torch.cat(output_scores)[output_permutation]                 # :122
        assert raw_pids == output_pids                                                # :124-126
        return pids, output_scores, output_D, output_D_mask, output_permutation

    def rank_forward(self, Q, pids, views=None, depth=10, output_D_embedding=False):  # :75
        pids, output_scores, output_D, output_D_mask, output_permutation = self._forward(Q, pids)
        scores_sorter = output_scores.sort(descending=True)                           # :128
        out_pids = pids[scores_sorter.indices].tolist()[:depth]                       # :129
        scores = output_scores[scores_sorter.indices].tolist()[:depth]                # :130
        if output_D_embedding:                                                        # :131-136
            output_D = torch.cat(output_D)[output_permutation]
            output_D_mask = torch.cat(output_D_mask)[output_permutation]
            output_D = output_D[scores_sorter.indices][:depth]
            output_D_mask = output_D_mask[scores_sorter.indices][:depth]
            return out_pids, output_D, output_D_mask
        return out_pids, scores


def ragged_scores_f64(index, doclens, offsets, pad_len, Q, pids):
    """Direct float64 evaluation of what the fused ragged kernel computes: for each candidate, the max over
    its REAL tokens of Q.D per query token, floored at 0 iff the reference would have padded it
    (pad_len > doclen), summed over query tokens.  Q [Lq,h]; returns [len(pids)]."""
    Qd = torch.as_tensor(Q).double().numpy()
    out = np.zeros(len(pids))
    idx = torch.as_tensor(index)
    for i, p in enumerate(pids):
        L, o = int(doclens[p]), int(offsets[p])
        if L == 0:
            out[i] = 0.0
            continue
        Dd = idx[o:o + L].double().numpy()
        mx = (Qd @ Dd.T).max(-1)
        if int(pad_len[p]) > L:
            mx = np.maximum(mx, 0.0)
        out[i] = mx.sum()
    return out
