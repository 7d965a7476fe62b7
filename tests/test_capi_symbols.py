"""CPU: libmaxsim.so loads and exports every symbol include/maxsim.h declares; argument validation that
returns before any launch (no compute without a GPU)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from colbert_amd import _lib
    return _lib


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "maxsim.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(maxsim_[a-z_0-9]+)\s*\(", src)))


def test_header_symbols_exported(lib):
    syms = declared_symbols()
    assert set(syms) == set(lib.SYMBOLS)
    raw = ctypes.CDLL(lib.LIB_PATH)
    for s in syms:
        assert hasattr(raw, s), s


def test_version_and_strerror(lib):
    assert lib.lib.maxsim_version() == 123
    assert lib.strerror(0) == "ok"
    for code in (-1, -2, -3, -4):
        assert lib.strerror(code) not in ("ok", "unknown error")


def test_validation_without_launch(lib):
    L = lib.lib
    # negative sizes / unknown dtypes
    assert L.maxsim_score_dense(None, None, None, None, -1, 1, 1, 1, 1, 0, 0, None, None) == lib.EINVAL
    assert L.maxsim_score_dense(None, None, None, None, 1, 1, 1, 1, 1, 9, 0, None, None) == lib.EINVAL
    assert L.maxsim_score_dense(None, None, None, None, 1, 1, 1, 1, 1, 0, 9, None, None) == lib.EINVAL
    # empty outputs are no-ops, empty doc axis is the reference's max-over-empty error
    assert L.maxsim_score_dense(None, None, None, None, 0, 5, 1, 1, 1, 0, 0, None, None) == lib.OK
    assert L.maxsim_score_dense(None, None, None, None, 2, 2, 1, 0, 1, 0, 0, None, None) == lib.EEMPTY
    assert L.maxsim_score_dense(None, None, None, None, 2, 2, 1, 1, 1, 0, 0, None, None) == lib.EINVAL  # null out
    # rerank: empty candidate list = assert len(pids) > 0 (colbert_ranker.py:76)
    assert L.maxsim_rerank(None, 0, 0, None, None, None, 0, None, 0, None, None, 1, 0, 32, 128, None, None) == lib.EEMPTY
    assert L.maxsim_rerank(None, 7, 0, None, None, None, 0, None, 0, None, None, 1, 1, 32, 128, None, None) == lib.EINVAL
    assert L.maxsim_rerank(None, 0, 0, None, None, None, 0, None, 5, None, None, 1, 1, 32, 128, None, None) == lib.EINVAL
    assert L.maxsim_rerank(None, 0, 0, None, None, None, 0, None, 0, None, None, 1, 1, 32, 128, None, None) == lib.EINVAL
    # the view-based entry points validate the same way (and reject a missing view)
    iv = lib.IndexView(None, 0, 128, 0, None, None, None, 0, None)
    assert L.maxsim_rerank_ex(None, None, 0, None, None, None, 1, 1, 32, None, None) == lib.EINVAL
    assert L.maxsim_rerank_ex(ctypes.byref(iv), None, 0, None, None, None, 1, 0, 32, None, None) == lib.EEMPTY
    assert L.maxsim_rerank_ex(ctypes.byref(iv), None, 0, None, None, None, 0, 1, 32, None, None) == lib.OK
    assert L.maxsim_rerank_ex(ctypes.byref(iv), None, 0, None, None, None, 1, 1, 32, None, None) == lib.EINVAL
    assert L.maxsim_rank_forward(ctypes.byref(iv), None, 0, 32, None, 0, 10, None, None, None, None, 0, None) == lib.EEMPTY
    assert L.maxsim_rank_forward(ctypes.byref(iv), None, 0, 32, None, 20000, 10, None, None, None, None, 0, None) == lib.ERANGE
    assert L.maxsim_rank_forward(ctypes.byref(iv), None, 0, 32, None, 5, 0, None, None, None, None, 0, None) == lib.EINVAL
    assert L.maxsim_rank_forward(None, None, 0, 32, None, 5, 1, None, None, None, None, 0, None) == lib.EINVAL
    assert L.maxsim_rank_forward(ctypes.byref(iv), None, 0, 32, None, 5, 1, None, None, None, None, 0, None) == lib.EINVAL   # no workspace
    assert L.maxsim_rank_forward_workspace_bytes(1000) == 4064
    assert L.maxsim_doc_table_bytes(1000) == 16000 and L.maxsim_doc_table_bytes(0) == 0
    assert L.maxsim_build_doc_table(None, None, None, 0, None, None) == lib.OK
    assert L.maxsim_build_doc_table(None, None, None, 4, None, None) == lib.EINVAL
    # which kernel serves an all-pairs shape (pure host logic): the training step takes the GEMM-blocked kernel
    assert L.maxsim_score_dense_kernel(272, 544, 32, 384, 768, lib.BF16, lib.MASK_F32) == 1
    assert L.maxsim_score_dense_kernel(272, 544, 32, 384, 768, lib.F32, lib.MASK_F32) == 0      # fp32 operands
    assert L.maxsim_score_dense_kernel(272, 544, 32, 384, 96, lib.BF16, lib.MASK_F32) == 0       # h % 64
    assert L.maxsim_score_dense_kernel(272, 544, 32, 385, 768, lib.BF16, lib.MASK_F32) == 0      # Ld > 384
    assert L.maxsim_score_dense_kernel(272, 544, 32, 384, 768, lib.BF16, lib.MASK_I64) == 0      # integer masks
    assert L.maxsim_score_dense_kernel(8, 64, 32, 384, 768, lib.BF16, lib.MASK_NONE) == 0        # too few tiles
    assert L.maxsim_score_dense_kernel(8, 64, 0, 384, 768, lib.BF16, lib.MASK_NONE) == lib.EINVAL
    assert L.maxsim_shard_candidates(None, 0, 10, 0, 5, None, None, None, None) == lib.OK
    assert L.maxsim_shard_candidates(None, 2, 10, 5, 0, None, None, None, None) == lib.EINVAL
    assert L.maxsim_shard_candidates(None, 2, 10, 0, 5, None, None, None, None) == lib.EINVAL
    # topk
    assert L.maxsim_topk(None, None, 1, 0, 1, None, None, None) == lib.EEMPTY
    assert L.maxsim_topk(None, None, 1, 20000, 1, None, None, None) == lib.ERANGE
    assert L.maxsim_topk(None, None, 1, 10, 0, None, None, None) == lib.EINVAL
    assert L.maxsim_topk(None, None, 1, 10, 1, None, None, None) == lib.EINVAL


def test_missing_library_fails_loudly(tmp_path, monkeypatch):
    """The product has no fallback: a missing .so is an ImportError, not a silent CPU path."""
    import importlib.util
    src = os.path.join(ROOT, "colbert_amd", "_lib.py")
    dst = tmp_path / "_lib_copy.py"
    dst.write_text(open(src).read())
    spec = importlib.util.spec_from_file_location("_lib_copy", dst)
    mod = importlib.util.module_from_spec(spec)
    with pytest.raises(ImportError):
        spec.loader.exec_module(mod)


def test_fastrank_glue_loads_and_validates():
    """colbert_amd/_fastrank.so (csrc/fastrank.c: CPython glue of rank_forward's online path) is built next to the
    library and refuses malformed calls before touching any pointer."""
    from colbert_amd import ranker
    fr = ranker._fastrank
    assert fr is not None, "colbert_amd/_fastrank.so missing: run __graft_entry__.build()"
    assert ranker._RANK_FORWARD_FN
    import pytest
    with pytest.raises(TypeError):
        fr.rank_forward(1, 2, 3, 0, 32, (1, 2), 10, 4, 5, 6, 7, 8, 9, 100)       # not a list
    with pytest.raises(ValueError):
        fr.rank_forward(1, 2, 3, 0, 32, [], 10, 4, 5, 6, 7, 8, 9, 100)           # empty
    with pytest.raises(ValueError):
        fr.rank_forward(1, 2, 3, 0, 32, [1], 0, 4, 5, 6, 7, 8, 9, 100)          # depth 0
    with pytest.raises(ValueError):
        fr.rank_forward(1, 2, 3, 0, 32, [1], 1, 0, 5, 6, 7, 8, 9, 100)          # null input buffer
    import ctypes
    buf = (ctypes.c_int64 * 4)()
    with pytest.raises(TypeError):
        fr.rank_forward(1, 2, 3, 0, 32, [1, 2.5], 1, ctypes.addressof(buf), 5, 6, 7, 8, 9, 100)   # a float in the list
    with pytest.raises(OverflowError):
        fr.rank_forward(1, 2, 3, 0, 32, [1, 2 ** 70], 1, ctypes.addressof(buf), 5, 6, 7, 8, 9, 100)
    # the list lands in the input buffer exactly (one- and two-digit ints, negative and > 2^60 through the general
    # conversion) and a library error code comes back as an int: a stand-in for maxsim_rank_forward that returns -3
    big = (ctypes.c_int64 * 9)()
    proto = ctypes.CFUNCTYPE(ctypes.c_int, *([ctypes.c_void_p] * 2 + [ctypes.c_int] * 2 + [ctypes.c_void_p] + [ctypes.c_int] * 2 +
                                             [ctypes.c_void_p] * 4 + [ctypes.c_int, ctypes.c_void_p]))
    seen = {}

    def stand_in(iv, q, q_dtype, lq, pids, n, depth, *rest):
        seen.update(n=n, depth=depth, lq=lq, sync=rest[4])
        return -3
    cb = proto(stand_in)
    vals = [0, 1, 2 ** 30 - 1, 2 ** 30, 2 ** 45 + 7, 2 ** 60 - 1, 2 ** 60, 5, 2 ** 63 - 2]
    fn = ctypes.cast(cb, ctypes.c_void_p).value
    rc = fr.rank_forward(fn, 2, 3, 0, 32, vals, 4, ctypes.addressof(big), 0, 6, 7, 8, 9, 2 ** 63 - 1)
    assert rc == -3 and list(big) == vals
    assert seen == {"n": 9, "depth": 4, "lq": 32, "sync": 1}
    # `self.doclens[pids]` (colbert_ranker.py:88): a pid outside [-n_docs, n_docs) is an IndexError BEFORE anything is
    # launched; a negative pid inside it is handed back (None) to the caller's general path, which wraps it as torch does
    seen.clear()
    with pytest.raises(IndexError, match="index 100 is out of bounds for dimension 0 with size 100"):
        fr.rank_forward(fn, 2, 3, 0, 32, [3, 100, 7], 4, ctypes.addressof(big), 0, 6, 7, 8, 9, 100)
    with pytest.raises(IndexError, match="index -101 is out of bounds"):
        fr.rank_forward(fn, 2, 3, 0, 32, [3, -101, 7], 4, ctypes.addressof(big), 0, 6, 7, 8, 9, 100)
    assert fr.rank_forward(fn, 2, 3, 0, 32, [3, -100, 99], 4, ctypes.addressof(big), 0, 6, 7, 8, 9, 100) is None
    assert seen == {}                                                   # the library was not called in any of the three
