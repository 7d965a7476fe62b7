"""Host-side mirror of the reference MaxSim operator, ``BaseModel.score``
(reference: colbert/modeling/BaseModel.py:39-46), backed by ``maxsim_score_dense`` in libmaxsim.so.

``MaxSimModel`` is an object that can be handed to the reference's ``ColbertRanker(model=...)``
(colbert/ranking/colbert_ranker.py:16,28,111): it exposes exactly ``score(Q, D, q_mask, d_mask)``.
"""
import torch

from . import _lib

_DT = {torch.float32: _lib.F32, torch.float16: _lib.F16, torch.bfloat16: _lib.BF16}
_MDT = {torch.int64: _lib.MASK_I64, torch.int32: _lib.MASK_I32, torch.float32: _lib.MASK_F32,
        torch.uint8: _lib.MASK_U8, torch.bool: _lib.MASK_U8}


def _stream(device):
    return torch.cuda.current_stream(device).cuda_stream


def _ptr(t):
    return None if t is None else t.data_ptr()


def _prepare(Q, D, q_mask, d_mask):
    if Q.dim() != 3 or D.dim() != 3 or q_mask.dim() != 2 or d_mask.dim() != 2:
        raise ValueError("score expects Q[q,m,h], D[d,n,h], q_mask[q,m], d_mask[d,n]")
    if not Q.is_cuda:
        raise RuntimeError("colbert_amd.score runs on the GPU only (libmaxsim has no CPU path)")
    nq, Lq, h = Q.shape
    nd, Ld, h2 = D.shape
    if h != h2 or tuple(q_mask.shape) != (nq, Lq) or tuple(d_mask.shape) != (nd, Ld):
        raise ValueError(f"shape mismatch: Q{tuple(Q.shape)} D{tuple(D.shape)} "
                         f"q_mask{tuple(q_mask.shape)} d_mask{tuple(d_mask.shape)}")
    dev = Q.device
    out_dtype = torch.promote_types(torch.promote_types(Q.dtype, q_mask.dtype), torch.promote_types(D.dtype, d_mask.dtype))
    cdt = Q.dtype if Q.dtype == D.dtype and Q.dtype in _DT else torch.float32
    if cdt != torch.float32 and (q_mask.dtype.is_floating_point or d_mask.dtype.is_floating_point):
        # a floating-point mask may hold token WEIGHTS (allowed by the interface, never built by the reference: its masks are
        # int64 0/1, tokenizers.py:36,39,57).  The 16-bit kernels fold Q * q_mask into a 16-bit query image and apply d_mask to
        # finished similarities, where the reference's products are exact in the promoted type (BaseModel.py:41-42) -- up to
        # 3e-3 on a bf16 score -- so 16-bit operands with ANY floating-point mask compute from fp32 copies.  The rule looks at
        # dtypes only: no reduction over the mask, no device-to-host sync in the training step's stream (integer / bool masks,
        # the reference's, keep the 16-bit GEMM path).
        cdt = torch.float32
    Qc = Q.detach().to(device=dev, dtype=cdt).contiguous()
    Dc = D.detach().to(device=dev, dtype=cdt).contiguous()
    mdt = d_mask.dtype if d_mask.dtype in _MDT else torch.float32
    if cdt != torch.float32:
        # the GEMM-blocked all-pairs kernel (16-bit operands) takes its mask rows by LDS-DMA as float words: hand the masks
        # over as float32 (the reference's int64 0/1 masks convert exactly; the kernels compute with float masks anyway)
        mdt = torch.float32
    qm = q_mask.to(device=dev, dtype=mdt).contiguous()
    dm = d_mask.to(device=dev, dtype=mdt).contiguous()
    return Qc, Dc, qm, dm, cdt, mdt, out_dtype


class _MaxSimFn(torch.autograd.Function):
    """Training form of ``score`` (its second caller: colbert/modeling/colbert_model.py:87-96).  The forward records the
    arg-max doc token of every (query, doc, query token); the backward routes gradients through it -- the
    ``[q, d, m, n]`` similarity tensor torch autograd would keep is never materialised."""

    @staticmethod
    def forward(ctx, Q, D, q_mask, d_mask):
        Qc, Dc, qm, dm, cdt, mdt, out_dtype = _prepare(Q, D, q_mask, d_mask)
        nq, Lq, h = Qc.shape
        nd, Ld, _ = Dc.shape
        dev = Qc.device
        out = torch.empty(nq, nd, dtype=torch.float32, device=dev)
        arg = torch.empty(nq, nd, Lq, dtype=torch.int32, device=dev)
        with torch.cuda.device(dev):
            rc = _lib.lib.maxsim_score_dense_fwd(_ptr(Qc), _ptr(Dc), _ptr(qm), _ptr(dm), nq, nd, Lq, Ld, h, _DT[cdt],
                                                 _MDT[mdt], _ptr(out), _ptr(arg), _stream(dev))
        if rc == _lib.EEMPTY:
            raise IndexError("max(): Expected reduction dim 3 to have non-zero size.")
        _lib.check(rc, "maxsim_score_dense_fwd")
        ctx.save_for_backward(Qc, Dc, qm, dm, arg)
        ctx.meta = (cdt, mdt, Q.dtype, D.dtype)
        return out if out_dtype == torch.float32 or not out_dtype.is_floating_point else out.to(out_dtype)

    @staticmethod
    def backward(ctx, g):
        Qc, Dc, qm, dm, arg = ctx.saved_tensors
        cdt, mdt, qdt, ddt = ctx.meta
        nq, Lq, h = Qc.shape
        nd, Ld, _ = Dc.shape
        dev = Qc.device
        g32 = g.to(device=dev, dtype=torch.float32).contiguous()
        dQ = torch.empty(nq, Lq, h, dtype=torch.float32, device=dev) if ctx.needs_input_grad[0] else None
        dD = torch.empty(nd, Ld, h, dtype=torch.float32, device=dev) if ctx.needs_input_grad[1] else None
        ws, ws_bytes = None, 0
        if dD is not None:
            ws_bytes = int(_lib.lib.maxsim_score_dense_bwd_workspace(nq, nd, Lq, Ld))
            if 0 < ws_bytes <= (2 << 30):       # per-doc inverse index scratch
                ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
            else:
                ws_bytes = 0
        HMAX = 1024                               # widest row the backward kernels take (maxsim_backward.h)
        with torch.cuda.device(dev):
            if h <= HMAX:
                rc = _lib.lib.maxsim_score_dense_bwd(_ptr(Qc), _ptr(Dc), _ptr(qm), _ptr(dm), _ptr(arg), _ptr(g32), nq, nd, Lq,
                                                     Ld, h, _DT[cdt], _MDT[mdt], _ptr(dQ), _ptr(dD), _ptr(ws), ws_bytes,
                                                     _stream(dev))
                _lib.check(rc, "maxsim_score_dense_bwd")
            else:
                # both gradients are sums of ROWS picked by the arg-max (dQ[q,m,:] = sum_d g D[d, arg, :], dD likewise):
                # they separate along the hidden dimension, so wider rows go through the same kernels in slabs of columns
                for h0 in range(0, h, HMAX):
                    h1 = min(h0 + HMAX, h)
                    Qs, Ds = Qc[..., h0:h1].contiguous(), Dc[..., h0:h1].contiguous()
                    dQs = None if dQ is None else torch.empty(nq, Lq, h1 - h0, dtype=torch.float32, device=dev)
                    dDs = None if dD is None else torch.empty(nd, Ld, h1 - h0, dtype=torch.float32, device=dev)
                    rc = _lib.lib.maxsim_score_dense_bwd(_ptr(Qs), _ptr(Ds), _ptr(qm), _ptr(dm), _ptr(arg), _ptr(g32), nq, nd,
                                                         Lq, Ld, h1 - h0, _DT[cdt], _MDT[mdt], _ptr(dQs), _ptr(dDs), _ptr(ws),
                                                         ws_bytes, _stream(dev))
                    _lib.check(rc, "maxsim_score_dense_bwd")
                    if dQ is not None:
                        dQ[..., h0:h1] = dQs
                    if dD is not None:
                        dD[..., h0:h1] = dDs
        return (None if dQ is None else dQ.to(qdt)), (None if dD is None else dD.to(ddt)), None, None


def score(Q, D, q_mask, d_mask, *args, **kwargs):
    """``scores[q, d] = sum_m max_n <Q[q,m]*q_mask[q,m], D[d,n]*d_mask[d,n]>`` on the GPU.

    Same signature, argument meaning and result shape as ``BaseModel.score`` (BaseModel.py:39-46); inputs are
    borrowed and never mutated; the result is a fresh tensor on the inputs' device whose dtype follows torch's
    promotion of ``Q * q_mask`` (fp32 for the rerank call, colbert_ranker.py:111-112).  The arithmetic is fp32.
    When autograd is recording and Q or D requires grad (the training call, colbert_model.py:90) the result is
    differentiable: see ``_MaxSimFn``.
    """
    if torch.is_grad_enabled() and (Q.requires_grad or D.requires_grad) and Q.size(1) > 0:
        return _MaxSimFn.apply(Q, D, q_mask, d_mask)
    Qc, Dc, qm, dm, cdt, mdt, out_dtype = _prepare(Q, D, q_mask, d_mask)
    nq, Lq, h = Qc.shape
    nd, Ld, _ = Dc.shape
    dev = Qc.device
    out = torch.empty(nq, nd, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        rc = _lib.lib.maxsim_score_dense(_ptr(Qc), _ptr(Dc), _ptr(qm), _ptr(dm), nq, nd, Lq, Ld, h, _DT[cdt],
                                         _MDT[mdt], _ptr(out), _stream(dev))
    if rc == _lib.EEMPTY:
        # torch: "max(): Expected reduction dim -1 to have non-zero size" (BaseModel.py:44)
        raise IndexError("max(): Expected reduction dim 3 to have non-zero size.")
    _lib.check(rc, "maxsim_score_dense")
    return out if out_dtype == torch.float32 or not out_dtype.is_floating_point else out.to(out_dtype)


class MaxSimModel:
    """Drop-in for the ``model`` argument of ``ColbertRanker`` (colbert_ranker.py:16,28): only ``score`` is used."""
    score = staticmethod(score)
