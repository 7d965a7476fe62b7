"""Host-side mirror of the reference rerank loop, ``ColbertRanker``
(reference: colbert/ranking/colbert_ranker.py:15-137), on an HBM-resident token index.

What changes against the reference, by design: the index lives in HBM once (no per-query CPU gather, pinned
buffer, PCIe copy or fp16->fp32 cast pass -- colbert_ranker.py:105-107), there are no padded per-bucket copies
of D (the fused kernel streams each candidate's real tokens and applies the reference's zero-padding floor
analytically), and a whole batch of queries is one launch (``rerank_batch``) instead of the per-query Python
loop of colbert/training/dense_server_client.py:44-48.  ``rank_forward`` keeps the reference's signature,
asserts and return contract so ``ColbertRetriever.search`` (colbert/indexing/faiss_indexers.py:224-235) can
call it unchanged.
"""
import array
import ctypes
import threading
from itertools import accumulate

import numpy as np
import torch

from . import _lib, index_io
from ._pinned import PinnedBuffer
from .scoring import _DT, _ptr, _stream

try:        # CPython glue of rank_forward's online path (csrc/fastrank.c, built next to libmaxsim.so); optional: without it
    from . import _fastrank          # the same library call is made through ctypes
except ImportError:
    _fastrank = None
_RANK_FORWARD_FN = ctypes.cast(_lib.lib.maxsim_rank_forward, ctypes.c_void_p).value

BSIZE = 1 << 14  # colbert_ranker.py:11


def torch_percentile(tensor, p):
    """colbert_ranker.py:238-241."""
    assert p in range(1, 100 + 1)
    assert tensor.dim() == 1
    return tensor.kthvalue(int(p * tensor.size(0) / 100.0)).values.item()


def reference_strides(doclens):
    """The length-bucket strides of colbert_ranker.py:36-40 for a 1-D int64 tensor of doclens."""
    strides = [torch_percentile(doclens, p) for p in [25, 50, 75]]            # :36
    strides.append(doclens.max().item())                                       # :39
    return sorted(list(set(strides)))                                          # :40


def strides_from_histogram(hist):
    """The same strides from a histogram of doclens (``hist[v]`` = number of docs with ``v`` tokens): ``kthvalue(k)`` is
    the smallest v whose cumulative count reaches k.  Exact, and a histogram can be summed across shards
    (``sharded.global_strides``) where the doclens themselves would have to be gathered."""
    hist = hist.to(torch.int64)
    n = int(hist.sum().item())
    cum = torch.cumsum(hist, 0)
    out = []
    for p in [25, 50, 75]:
        k = int(p * n / 100.0)                                                 # torch_percentile, :238-241
        if k < 1:
            raise RuntimeError("kthvalue(): selected number k out of range for dimension 0")   # what the reference raises for N < 4
        out.append(int(torch.searchsorted(cum, torch.tensor(k), right=False).item()))
    out.append(int(torch.nonzero(hist).max().item()))                          # doclens.max()
    return sorted(list(set(out)))


class _Workspace:
    """Per-thread buffers of ``rank_forward``: coherent pinned host memory the GPU reads the pid list from and writes the
    top-k (and its completion word) to -- no memcpy calls on the way in or out -- and the device scratch of
    ``maxsim_rank_forward`` (zeroed counters + the score vector)."""

    def __init__(self, device):
        # in: ordinary pinned memory (the host writes it before the launch; the GPU reads it through its caches --
        # host-coherent memory would turn the 1000 lanes' 8-byte reads into 1000 uncached PCIe transactions)
        self.pin_in_t = torch.empty(BSIZE, dtype=torch.int64).pin_memory()
        self.pin_in = self.pin_in_t.numpy()
        self.in_ptr = self.pin_in_t.data_ptr()
        # out + completion word: host-COHERENT pinned memory (the host polls it while the kernel is still running)
        self.pin = PinnedBuffer(BSIZE * 8 + BSIZE * 4 + 64)
        self.pin_out_p = self.pin.view(np.int64, 0, BSIZE)
        self.pin_out_s = self.pin.view(np.float32, BSIZE * 8, BSIZE)
        self.pin_flag = self.pin.view(np.uint32, BSIZE * 12, 16)
        self.pin_flag[:] = 0
        self.out_p_ptr, self.out_s_ptr, self.flag_ptr = self.pin.ptr, self.pin.ptr + BSIZE * 8, self.pin.ptr + BSIZE * 12
        self.scratch = torch.zeros(int(_lib.lib.maxsim_rank_forward_workspace_bytes(BSIZE)), dtype=torch.uint8, device=device)
        self.scratch_ptr = self.scratch.data_ptr()
        torch.cuda.synchronize(device)          # the zero fill has landed before the first launch on any stream


class ColbertRanker:
    """``ColbertRanker(index_path, model=None, dim=None)`` as in colbert_ranker.py:16; ``model`` is accepted for
    signature compatibility (the fused kernel replaces ``model.score``).  Keyword-only extras:

    parts / parts_doclens : build from in-memory tensors instead of ``index_path``
    device                : the GPU holding the index
    index_dtype           : storage dtype in HBM (reference: fp16, colbert_ranker.py:62)
    fp32_mode             : for an fp32 index with dim 128: "exact" (default; f32-input MFMA, an exact fp32 fmaf chain),
                            "bf16x3" (both operands cut exactly into three bf16 pieces, six piece products on the bf16
                            matrix pipe: fp32-class accuracy, no magnitude limit, ~7 % faster on uniform 180-token docs,
                            ~15 % on ragged docs) or "fast" (fp16 hi+lo pieces, three products, |error| ~1e-6 on a
                            score, needs |x| < 65504, ~11 % faster).
                            When to pick what: keep "exact" when scores must be bitwise reproducible against an fp32
                            fmaf chain (the parity tests' strictest form); pick "bf16x3" for an fp32 index -- L2-normalised
                            rows or not -- when the stated tolerance (|d| <= 1e-4 on a score) is the requirement and
                            throughput matters: the exact mode is power-capped at 0.73 (uniform) / 0.67-0.70 (ragged) of
                            the HBM peak, bf16x3 runs at 0.78 / 0.80.  The reference stores fp16 (colbert_ranker.py:62):
                            with index_dtype=torch.float16 (the default here) the mode does not apply.
    strides               : length-bucket strides to use instead of the percentiles of THIS index's doclens
                            (colbert_ranker.py:36-40).  A doc-shard must be given the strides of the whole index
                            (``sharded.global_strides``): the 0-floor depends on them (:90, :108-109)
    """

    def __init__(self, index_path=None, model=None, dim=None, *, parts=None, parts_doclens=None, device="cuda",
                 index_dtype=torch.float16, fp32_mode="exact", strides=None):
        part_iter = None
        if index_path is not None:
            _, parts_paths, _ = index_io.get_parts(index_path)                # :18
            parts_doclens = index_io.load_doclens(index_path, flatten=False)  # :22
            # parts are streamed to HBM one file at a time (:61-73 keeps the whole index in host RAM; here it never is)
            part_iter = (index_io.load_index_part(f) for f in parts_paths)
        elif parts is not None:
            part_iter = iter(parts)
        if part_iter is None or parts_doclens is None:
            raise ValueError("give index_path or parts+parts_doclens")
        self.maxsim_dtype = torch.float32                                     # :20
        self.parts_doclens = parts_doclens
        self.model = model
        self.device = torch.device(device)
        if self.device.type == "cuda" and self.device.index is None:          # "cuda" -> the current device, spelled out: tensors
            self.device = torch.device("cuda", torch.cuda.current_device())   # report cuda:N, and rank_forward compares devices
        assert fp32_mode in ("exact", "fast", "bf16x3")
        self.fp32_mode = fp32_mode
        doclens = [int(x) for y in parts_doclens for x in y]                  # flatten, utils.py:133
        self.num_embeddings = sum(doclens)
        assert index_dtype in _DT
        # one HBM-resident [num_embeddings, dim] token matrix (no +512 tail: the kernel never reads past a doc)
        self.tensor = None
        offset = 0
        for part, dl in zip(part_iter, parts_doclens):
            if self.tensor is None:
                dim = part.size(-1) if dim is None else dim
                self.tensor = torch.empty(max(self.num_embeddings, 1), dim, dtype=index_dtype, device=self.device)
            endpos = offset + sum(dl)
            assert part.size(0) == endpos - offset, (part.size(0), endpos - offset)
            self.tensor[offset:endpos] = part.to(device=self.device, dtype=index_dtype)
            offset = endpos
            del part
        assert self.tensor is not None and offset == self.num_embeddings
        self.init_ranker(doclens, strides)

    @classmethod
    def from_device_tensor(cls, tensor, doclens, model=None, fp32_mode="exact", strides=None):
        """Adopts an index that already sits in HBM: ``tensor`` [sum(doclens), dim] (fp32/fp16/bf16, contiguous) is
        used as is, no copy.  (Synthetic benchmarks; shards handed over by another component.)"""
        self = cls.__new__(cls)
        assert tensor.is_cuda and tensor.dim() == 2 and tensor.is_contiguous() and tensor.dtype in _DT
        assert tensor.size(0) == sum(doclens)
        assert fp32_mode in ("exact", "fast", "bf16x3")
        self.maxsim_dtype = torch.float32
        self.parts_doclens = [doclens]
        self.model = model
        self.device = tensor.device
        self.fp32_mode = fp32_mode
        self.num_embeddings = tensor.size(0)
        self.tensor = tensor
        self.init_ranker([int(x) for x in doclens], strides)
        return self

    def init_ranker(self, doclens, strides=None):                             # :31-43
        pfx = [0] + list(accumulate(doclens))
        self.doclens = torch.tensor(doclens, dtype=torch.int64)
        self.doclens_pfxsum = torch.tensor(pfx, dtype=torch.int64)
        self.dim = self.tensor.size(-1)
        dev = self.device
        self.d_doclens = self.doclens.to(dev, torch.int32)
        self.d_offsets = self.doclens_pfxsum[:-1].contiguous().to(dev)
        self.n_docs = len(doclens)
        self._tls = threading.local()
        # candidate-side glue: the doc of every 64th token row (maxsim_build_row_blocks; the reference's emb2pid,
        # colbert_ranker.py:163-174, 64x smaller) -- what embedding_ids_to_pids looks token rows up in
        self.d_row_blocks = None
        if dev.type == "cuda":
            nbytes = int(_lib.lib.maxsim_row_blocks_bytes(self.num_embeddings))
            self.d_row_blocks = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=dev)
            with torch.cuda.device(dev):
                rc = _lib.lib.maxsim_build_row_blocks(_ptr(self.d_offsets), self.n_docs, self.num_embeddings,
                                                      _ptr(self.d_row_blocks), _stream(dev))
            _lib.check(rc, "maxsim_build_row_blocks")
        self.set_strides(reference_strides(self.doclens) if strides is None else strides)

    def set_strides(self, strides):
        """(Re)derives the per-doc bucket stride -- the length the reference would pad the doc to, :90: the smallest
        stride >= doclen -- and the packed descriptor table the kernels read.  Called with the strides of the WHOLE index
        on a doc-shard."""
        self.strides = sorted(set(int(x) for x in strides))
        st = torch.tensor(self.strides)
        assert int(self.doclens.max().item()) <= self.strides[-1], "a doc is longer than the largest stride"
        assignments = (self.doclens.unsqueeze(1) > st.unsqueeze(0) + 1e-6).sum(-1)     # :90
        dev = self.device
        self.d_pad_len = st[assignments].to(dev, torch.int32)
        self.d_doc_table = None
        if dev.type == "cuda":
            nbytes = int(_lib.lib.maxsim_doc_table_bytes(self.n_docs))
            self.d_doc_table = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=dev)
            with torch.cuda.device(dev):
                rc = _lib.lib.maxsim_build_doc_table(_ptr(self.d_offsets), _ptr(self.d_doclens), _ptr(self.d_pad_len),
                                                     self.n_docs, _ptr(self.d_doc_table), _stream(dev))
            _lib.check(rc, "maxsim_build_doc_table")
        self._iv = self._index_view()
        self._iv_ref = ctypes.byref(self._iv)
        self._iv_addr = ctypes.addressof(self._iv)
        self._fast_ok = dev.type == "cuda"
        self._dev_index = (dev.index if dev.index is not None else torch.cuda.current_device()) if dev.type == "cuda" else -1

    def _index_view(self):
        idt = _DT[self.tensor.dtype]
        if idt == _lib.F32:
            idt = {"exact": _lib.F32, "fast": _lib.F32_FAST, "bf16x3": _lib.F32_BF16X3}[getattr(self, "fp32_mode", "exact")]
        # every doc the same length and that length the only bucket (no padding anywhere): the multi-view configuration
        # (dense.yaml:31-32: every doc keeps d_view viewer tokens) -- the library then runs its fixed-length kernel
        lo, hi = int(self.doclens.min().item()), int(self.doclens.max().item())
        uniform = lo if (lo == hi and lo > 0 and lo in self.strides) else 0    # lo in strides: pad_len == doclen
        return _lib.IndexView(_ptr(self.tensor), idt, self.dim, self.num_embeddings, _ptr(self.d_offsets),
                              _ptr(self.d_doclens), _ptr(self.d_pad_len), self.n_docs, _ptr(self.d_doc_table), uniform,
                              ctypes.sizeof(_lib.IndexView))

    # ------------------------------------------------------------------------------------------
    def score_candidates(self, Q, cand_pids, q_len=None, q_mask=None, cand_count=None):
        """Q [nq, Lq, h] (token-major), cand_pids [nq, ncand] int64 LOCAL pids (<0 = padding slot)
        -> scores [nq, ncand] fp32 on the device.  ``q_len`` [nq] drops the tokens from that position on, ``q_mask``
        [nq, Lq] (0 = dropped) any tokens -- the batched form of the per-query ``keep_nonzero`` (training_utils.py:48-53)
        the reference applies to ``q_active_padding`` at dense_server_client.py:45.
        ``cand_count`` [nq] int32 (device): rows are COUNTED -- row q's live entries are its first ``cand_count[q]`` slots,
        every later slot is negative (``shard_candidates`` / ``embedding_ids_to_pids`` rows).  Same scores; the launch is
        scheduled from a device-built dense work list instead of the full-width grid (``maxsim_rerank_counted``)."""
        dev = self.device
        if dev.type != "cuda":
            raise RuntimeError("colbert_amd scores on the GPU only: the index must live in HBM (libmaxsim has no CPU path)")
        # a 16-bit query is passed through in its own dtype (no query bits are invented); anything else as fp32
        qdt = Q.dtype if Q.dtype in (torch.float16, torch.bfloat16) else torch.float32
        Q = Q.to(device=dev, dtype=qdt).contiguous()
        cand = cand_pids.to(device=dev, dtype=torch.int64).contiguous()
        nq, Lq, h = Q.shape
        assert h == self.dim, (h, self.dim)
        assert cand.dim() == 2 and cand.size(0) == nq
        ncand = cand.size(1)
        ql = None if q_len is None else q_len.to(device=dev, dtype=torch.int32).contiguous()
        qm = None
        if q_mask is not None:
            assert tuple(q_mask.shape) == (nq, Lq), (tuple(q_mask.shape), (nq, Lq))
            qm = q_mask.to(dev)                                                # q_word_mask.bool(), training_utils.py:50
            qm = (qm if qm.dtype == torch.bool else (qm != 0)).contiguous().view(torch.uint8)   # (a bool mask's bytes: no copy)
        scores = torch.empty(nq, ncand, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            if cand_count is None:
                rc = _lib.lib.maxsim_rerank_ex(ctypes.byref(self._iv), _ptr(Q), _DT[qdt], _ptr(ql), _ptr(qm), _ptr(cand),
                                               nq, ncand, Lq, _ptr(scores), _stream(dev))
            else:
                cc = cand_count.to(device=dev, dtype=torch.int32).contiguous()
                assert cc.numel() == nq
                nbytes = int(_lib.lib.maxsim_worklist_bytes(nq, ncand))
                wl = torch.empty(nbytes, dtype=torch.uint8, device=dev)      # caching allocator: no HIP call in steady state
                rc = _lib.lib.maxsim_rerank_counted(ctypes.byref(self._iv), _ptr(Q), _DT[qdt], _ptr(ql), _ptr(qm), _ptr(cand),
                                                    _ptr(cc), nq, ncand, Lq, _ptr(scores), _ptr(wl), nbytes, _stream(dev))
        if rc == _lib.EEMPTY:
            raise AssertionError("len(pids) > 0")  # colbert_ranker.py:76
        _lib.check(rc, "maxsim_rerank_ex" if cand_count is None else "maxsim_rerank_counted")
        return scores

    def topk(self, scores, pids, k, counts=None):
        """Per-query top-k (score desc): scores [nq, n] fp32, pids [nq, n] int64 or None -> ([nq,k], [nq,k]).
        ``counts`` [nq] int32: counted rows (see ``score_candidates``): only the live slots are ranked."""
        dev = scores.device
        nq, n = scores.shape
        out_s = torch.empty(nq, k, dtype=torch.float32, device=dev)
        out_p = torch.empty(nq, k, dtype=torch.int64, device=dev)
        scores = scores.contiguous()
        pids = None if pids is None else pids.to(device=dev, dtype=torch.int64).contiguous()
        with torch.cuda.device(dev):
            if counts is None:
                rc = _lib.lib.maxsim_topk(_ptr(scores), _ptr(pids), nq, n, k, _ptr(out_s), _ptr(out_p), _stream(dev))
            else:
                cc = counts.to(device=dev, dtype=torch.int32).contiguous()
                rc = _lib.lib.maxsim_topk_counted(_ptr(scores), _ptr(pids), _ptr(cc), nq, n, k, _ptr(out_s), _ptr(out_p),
                                                  _stream(dev))
        _lib.check(rc, "maxsim_topk")
        return out_p, out_s

    def rerank_batch(self, Q, cand_pids, depth=10, q_len=None, q_mask=None, cand_count=None):
        """Batched form of the per-query loop dense_server_client.py:44-48: one launch for all queries.
        Returns device tensors (pids [nq,k], scores [nq,k]) with k = min(depth, ncand).  ``cand_count``: see
        ``score_candidates`` (counted rows, e.g. straight from ``embedding_ids_to_pids(trim=False)``)."""
        scores = self.score_candidates(Q, cand_pids, q_len, q_mask, cand_count)
        k = min(int(depth), scores.size(1))
        return self.topk(scores, cand_pids, k, cand_count)

    # ------------------------------------------------------------------------------------------
    def embedding_ids_to_pids(self, embedding_ids, trim=True, keep=None, id_base=0):
        """GPU form of ``ColbertIndex.embedding_ids_to_pids`` (colbert_ranker.py:212-229): token rows returned by the
        ANN search, ``[nq, Lq * faiss_depth]`` int64 (as reshaped at colbert_ranker.py:178; or ``[nq, Lq, faiss_depth]``), ->
        per-query DISTINCT pids, ascending, padded with -1: a ``cand_pids`` matrix for ``rerank_batch`` -- no ``.tolist()`` /
        ``set()`` / ``Pool(16)`` hop through the host.  Returns ``(cand [nq, width], counts [nq])``; ``trim`` cuts the width
        to the largest count (one host sync).
        ``keep`` [nq, Lq] (0 = dropped): the neighbours of a dropped query token do not count (``keep_nonzero``,
        training_utils.py:48-53) -- applied in the kernel, the ids are not rewritten.  ``id_base``: subtracted from every id;
        ids outside ``[id_base, id_base + num_embeddings)`` are dropped (a doc shard passes the global row of its first
        token)."""
        dev = self.device
        if dev.type != "cuda":
            raise RuntimeError("colbert_amd runs on the GPU only (libmaxsim has no CPU path)")
        e = embedding_ids.to(device=dev, dtype=torch.int64).contiguous()
        assert e.dim() in (2, 3)
        nq = e.size(0)
        n = e.numel() // max(nq, 1) if nq else (e.size(1) if e.dim() == 2 else e.size(1) * e.size(2))
        km, per_tok = None, 1
        if keep is not None:
            km = keep.to(dev)
            km = (km if km.dtype == torch.bool else (km != 0)).contiguous().view(torch.uint8)     # bool -> bytes: no copy
            assert km.dim() == 2 and km.size(0) == nq and km.size(1) > 0 and n % km.size(1) == 0, (tuple(km.shape), n)
            per_tok = n // km.size(1)
        out = torch.empty(nq, n, dtype=torch.int64, device=dev)
        cnt = torch.empty(nq, dtype=torch.int32, device=dev)
        with torch.cuda.device(dev):
            rc = _lib.lib.maxsim_embedding_ids_to_pids_ex(_ptr(e), nq, n, per_tok, _ptr(km), int(id_base), _ptr(self.d_offsets),
                                                          self.n_docs, self.num_embeddings, _ptr(self.d_row_blocks), _ptr(out),
                                                          _ptr(cnt), _stream(dev))
        _lib.check(rc, "maxsim_embedding_ids_to_pids_ex")
        if trim and nq > 0:
            out = out[:, :max(int(cnt.max().item()), 1)].contiguous()
        return out, cnt

    # ------------------------------------------------------------------------------------------
    def rank_forward(self, Q, pids, views=None, depth=10, output_D_embedding=False):
        """colbert_ranker.py:75-137.  Q is [1, h, Lq] (dim-major, as faiss_indexers.py:232-233 hands it over)."""
        assert len(pids) > 0                                                  # :76
        assert Q.size(0) in [1, len(pids)]                                    # :77
        # the call as faiss_indexers.py:232-234 makes it -- a python list of pids, q a permuted view of a contiguous fp32
        # [1, Lq, h] tensor on the index's device, depth only: straight to the C glue (every other form takes the general
        # path below, which normalises Q first; ~3 us of interpreter work less on a ~40 us call)
        if type(pids) is list and _fastrank is not None and not output_D_embedding and self._fast_ok:
            shp, std = Q.shape, Q.stride()
            if (shp[0] == 1 and shp[1] == self.dim and std[1] == 1 and std[2] == shp[1] and Q.dtype is torch.float32
                    and Q.device == self.device and len(pids) <= BSIZE and torch.cuda.current_device() == self._dev_index):
                ws = getattr(self._tls, "ws", None)
                if ws is None:
                    ws = self._tls.ws = _Workspace(self.device)
                try:
                    r = _fastrank.rank_forward(_RANK_FORWARD_FN, self._iv_addr, Q.data_ptr(), _lib.F32, shp[2], pids,
                                               min(int(depth), len(pids)), ws.in_ptr, ws.scratch_ptr, ws.out_p_ptr, ws.out_s_ptr,
                                               ws.flag_ptr, torch._C._cuda_getCurrentRawStream(self._dev_index), self.n_docs)
                except (TypeError, OverflowError):
                    r = None
                if r is not None:
                    if type(r) is int:
                        _lib.check(r, "maxsim_rank_forward")
                    return r
        if Q.size(0) != 1:
            # the reference's per-candidate-query branch (:103) takes row [0] of an all-pairs result (:112) --
            # a latent bug that is never exercised (faiss_indexers.py:232-234 always passes one query).
            raise NotImplementedError("rank_forward with one query per candidate is not exercised by the reference")
        n_pids = len(pids)
        dev = self.device
        if dev.type != "cuda":
            raise RuntimeError("colbert_amd scores on the GPU only: the index must live in HBM (libmaxsim has no CPU path)")
        k = min(int(depth), n_pids)
        Qt = Q.permute(0, 2, 1)                                               # :111 -> [1, Lq, h]; a view of the caller's q
        qdt = Qt.dtype if Qt.dtype in (torch.float16, torch.bfloat16) else torch.float32
        if Qt.dtype != qdt or Qt.device != dev:                               # :78 (a no-op for faiss_indexers.py:232-233)
            Qt = Qt.to(device=dev, dtype=qdt)
        if not Qt.is_contiguous():
            Qt = Qt.contiguous()
        assert Qt.size(2) == self.dim, (Qt.size(2), self.dim)
        if n_pids > BSIZE or output_D_embedding:
            return self._rank_forward_general(Qt, pids, k, output_D_embedding)
        # the online call: ONE library call (rerank + top-k enqueued back to back, then a poll on the completion word the
        # top-k kernel stores).  The pid list goes in through pinned host memory the kernels read directly and the top-k
        # comes out through host-coherent pinned memory they write directly: no memcpy calls, no allocations (per-thread
        # workspace), no device tensors created
        ws = getattr(self._tls, "ws", None)
        if ws is None:
            ws = self._tls.ws = _Workspace(dev)
        idx = dev.index if dev.index is not None else torch.cuda.current_device()
        if type(pids) is list and _fastrank is not None and torch.cuda.current_device() == idx:
            # list in, lists out, one C call in between (csrc/fastrank.c: the same library call as below, ~10 us less
            # interpreter work around it); anything unusual in the list takes the general path
            try:
                r = _fastrank.rank_forward(_RANK_FORWARD_FN, ctypes.addressof(self._iv), Qt.data_ptr(), _DT[qdt], Qt.size(1),
                                           pids, k, ws.in_ptr, ws.scratch_ptr, ws.out_p_ptr, ws.out_s_ptr, ws.flag_ptr,
                                           torch._C._cuda_getCurrentRawStream(idx), self.n_docs)
            except (TypeError, OverflowError):
                r = None
            if r is not None:
                if type(r) is int:
                    _lib.check(r, "maxsim_rank_forward")
                return r
        # `self.doclens[pids]` (:88): a pid >= n_docs or < -n_docs raises IndexError, one in [-n_docs, -1] wraps
        if type(pids) is list:
            try:        # 1000 python ints: 8 us through array('q') against 50-75 us for torch.tensor(list)
                a = array.array("q", pids)
                ctypes.memmove(ws.in_ptr, a.buffer_info()[0], 8 * n_pids)
            except (TypeError, OverflowError):
                ws.pin_in[:n_pids] = np.asarray(pids, dtype=np.int64)
            lo, hi = int(ws.pin_in[:n_pids].min()), int(ws.pin_in[:n_pids].max())
            pid_ptr = ws.in_ptr
        elif pids.is_cuda:
            pid_keep = pids.to(dev, torch.int64).contiguous()
            lo, hi = (int(x) for x in torch.aminmax(pid_keep))
            pid_ptr = pid_keep.data_ptr()
        else:
            ws.pin_in[:n_pids] = pids.to(torch.int64).numpy()
            lo, hi = int(ws.pin_in[:n_pids].min()), int(ws.pin_in[:n_pids].max())
            pid_ptr = ws.in_ptr
        self._check_pid_range(lo, hi)
        if lo < 0:
            return self._rank_forward_general(Qt, pids, k, False)
        switch = torch.cuda.current_device() != idx
        if switch:
            prev = torch.cuda.current_device()
            torch.cuda.set_device(idx)
        try:
            rc = _lib.lib.maxsim_rank_forward(self._iv_ref, Qt.data_ptr(), _DT[qdt], Qt.size(1), pid_ptr, n_pids, k,
                                              ws.scratch_ptr, ws.out_p_ptr, ws.out_s_ptr, ws.flag_ptr, 1,
                                              torch._C._cuda_getCurrentRawStream(idx))
        finally:
            if switch:
                torch.cuda.set_device(prev)
        _lib.check(rc, "maxsim_rank_forward")
        return ws.pin_out_p[:k].tolist(), ws.pin_out_s[:k].tolist()

    def _check_pid_range(self, lo, hi):
        """``self.doclens[pids]`` (colbert_ranker.py:88) raises IndexError for an index outside [-n_docs, n_docs)."""
        if hi >= self.n_docs or lo < -self.n_docs:
            raise IndexError(f"index {hi if hi >= self.n_docs else lo} is out of bounds for dimension 0 with size {self.n_docs}")

    def _rank_forward_general(self, Qt, pids, k, output_D_embedding):
        """rank_forward through the batched entry points: lists longer than BSIZE, ``output_D_embedding``, and pid lists
        with negative entries.  Negative pids index from the end, as torch indexing does at colbert_ranker.py:88; the
        returned pids are the caller's own values (:129 returns ``pids[order]``).  (The reference pairs ``doclens[pid]``
        with ``doclens_pfxsum[pid]`` of a table that is one entry longer, so for a negative pid it would read the NEXT doc's
        tokens with this doc's length -- a latent misalignment that is not reproduced: both wrap consistently here.)"""
        dev = self.device
        pids_t = (torch.tensor(pids) if type(pids) is list else pids).to(dev, torch.int64).view(1, -1)
        lo, hi = (int(x) for x in torch.aminmax(pids_t))
        self._check_pid_range(lo, hi)
        cand = torch.where(pids_t < 0, pids_t + self.n_docs, pids_t) if lo < 0 else pids_t
        scores = self.score_candidates(Qt, cand)
        top_p, top_s = self.topk(scores, pids_t, k)                           # :128-130
        if output_D_embedding:                                                # :131-136
            top_c, _ = self.topk(scores, cand, k)
            return self._output_D(top_p[0], top_c[0], cand[0])
        return top_p[0].tolist(), top_s[0].tolist()

    def _output_D(self, top_pids, top_rows, all_cand):
        """colbert_ranker.py:131-136: D [k, S, h] fp32 and mask [k, S] of the top docs, as the reference's strided view
        hands them over (:49, :105): slot t of a doc is token row offset + t of the concatenated index WHATEVER doc it
        belongs to -- slots past the doc's end hold the next docs' tokens (zeros past the end of the index: the
        reference's +512-row tail, :62) and are flagged only by the mask.  The reference's ``torch.cat(output_D)`` (:132)
        works only when ALL candidates fall in ONE length bucket; same restriction, same error."""
        pad = self.d_pad_len[all_cand].to(torch.int64)
        S = int(pad.max().item())
        if not bool((pad == S).all()):
            raise RuntimeError("Sizes of tensors must match except in dimension 0 (candidates span several length buckets)")
        rows = self.d_offsets[top_rows].unsqueeze(1) + torch.arange(S, device=self.device).unsqueeze(0)
        mask = torch.arange(S, device=self.device).unsqueeze(0) + 1 <= self.d_doclens[top_rows].unsqueeze(1)   # :108-109
        inside = rows < self.num_embeddings
        D = self.tensor[rows.clamp(max=max(self.num_embeddings - 1, 0))].to(self.maxsim_dtype)
        D = D * inside.unsqueeze(-1)                                          # the zero tail behind the last doc
        return top_pids.tolist(), D, mask
