"""Cross-rank glue of the training-form ``score``: the all-gather of Q / D / masks that precedes it.

Reference: ``collection_qd_masks`` / ``distributed_concat`` (colbert/training/training_utils.py:22-45), called from
``ColbertModel.forward`` (colbert/modeling/colbert_model.py:87) right before ``self.score`` (:90): every rank gathers
the other ranks' query and document embeddings and masks, puts its OWN tensor back into its slot so that slot keeps its
autograd history (``all_t[rank] = t``, :41), and concatenates along the batch axis -- in-batch negatives across the
whole job, gradients only into the local slot (DDP averages the parameter gradients afterwards).

Here each tensor is ONE ``all_gather_into_tensor`` (RCCL over xGMI with backend ``nccl``) straight into the
concatenated result -- no per-rank clones, no list, no ``torch.cat`` -- wrapped in an autograd function whose backward
hands the local slot's gradient slice back to the local tensor: exactly the graph the reference builds.
"""
import torch
import torch.distributed as dist


class _GatherKeepLocalGrad(torch.autograd.Function):
    @staticmethod
    def forward(ctx, t, group):
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        t = t.contiguous()
        out = torch.empty((world * t.size(0),) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        dist.all_gather_into_tensor(out, t, group=group)
        ctx.slot = (rank * t.size(0), (rank + 1) * t.size(0))
        return out

    @staticmethod
    def backward(ctx, g):
        return g[ctx.slot[0]:ctx.slot[1]], None


def distributed_concat(tensor, num_total_examples=None, concat=True, group=None):
    """training_utils.py:22-32.  ``concat=False`` returns the per-rank list (views of one gathered buffer)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        res = tensor
        world = 1
    else:
        world = dist.get_world_size(group)
        res = _GatherKeepLocalGrad.apply(tensor, group) if tensor.requires_grad else _gather_plain(tensor, group)
    if not concat:
        return list(res.chunk(world, dim=0))
    return res[:num_total_examples] if num_total_examples else res


def _gather_plain(t, group):
    world = dist.get_world_size(group)
    t = t.contiguous()
    out = torch.empty((world * t.size(0),) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    dist.all_gather_into_tensor(out, t, group=group)
    return out


def collection_qd_masks(data, group=None):
    """training_utils.py:35-45: ``[Q, q_mask, D, d_mask]`` of this rank -> the same four for the whole job, concatenated
    along dim 0 in rank order; this rank's slot of every differentiable tensor keeps its gradient (:41)."""
    return [distributed_concat(t, group=group) for t in data]


def in_batch_scores(Q, D, q_mask, d_mask, score_fn=None, group=None):
    """colbert_model.py:87-90: gather, then all-pairs MaxSim ``[B*W, 2B*W]``.  ``score_fn`` defaults to the HIP operator
    (``colbert_amd.score``, differentiable through arg-max routing)."""
    if score_fn is None:
        from .scoring import score as score_fn
    Qa, qm, Da, dm = collection_qd_masks([Q, q_mask, D, d_mask], group=group)
    return score_fn(Qa, Da, qm, dm)
