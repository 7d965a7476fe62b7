"""ctypes binding of libmaxsim.so (the C ABI declared in include/maxsim.h).

The library is the product: there is NO CPU or torch fallback.  If the shared object is missing or does not
export a declared symbol, importing this module raises immediately.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MAXSIM_LIB", os.path.join(_HERE, "libmaxsim.so"))  # MAXSIM_LIB: A/B builds

# include/maxsim.h
F32, F16, BF16, F32_FAST, F32_BF16X3 = 0, 1, 2, 3, 4
MASK_NONE, MASK_I64, MASK_I32, MASK_F32, MASK_U8 = 0, 1, 2, 3, 4
OK, EINVAL, EEMPTY, ERANGE, ELAUNCH = 0, -1, -2, -3, -4

SYMBOLS = ("maxsim_version", "maxsim_strerror", "maxsim_score_dense", "maxsim_rerank", "maxsim_topk",
           "maxsim_embedding_ids_to_pids", "maxsim_score_dense_fwd", "maxsim_score_dense_bwd",
           "maxsim_score_dense_bwd_workspace", "maxsim_rerank_ex", "maxsim_rank_forward", "maxsim_rank_forward_workspace_bytes", "maxsim_doc_table_bytes",
           "maxsim_build_doc_table", "maxsim_shard_candidates", "maxsim_score_dense_kernel", "maxsim_worklist_bytes",
           "maxsim_rerank_counted", "maxsim_topk_counted", "maxsim_hbm_read_probe", "maxsim_hbm_read_probe_scattered",
           "maxsim_host_alloc_coherent", "maxsim_host_free", "maxsim_embedding_ids_to_pids_ex", "maxsim_row_blocks_bytes",
           "maxsim_build_row_blocks", "maxsim_index_view_bytes")


class IndexView(ctypes.Structure):
    """``maxsim_index_view`` (include/maxsim.h)."""
    _fields_ = [("index", ctypes.c_void_p), ("index_dtype", ctypes.c_int32), ("h", ctypes.c_int32),
                ("n_tokens", ctypes.c_int64), ("tok_offsets", ctypes.c_void_p), ("doclens", ctypes.c_void_p),
                ("pad_len", ctypes.c_void_p), ("n_docs", ctypes.c_int64), ("doc_table", ctypes.c_void_p),
                ("uniform_len", ctypes.c_int32), ("struct_size", ctypes.c_int32)]


class MaxSimError(RuntimeError):
    def __init__(self, code, where):
        self.code = code
        super().__init__(f"{where}: libmaxsim error {code} ({strerror(code)})")


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            f"or colbert_amd/csrc/build.sh (hipcc --offload-arch=gfx950). There is no fallback path.")
    lib = ctypes.CDLL(LIB_PATH)
    for s in SYMBOLS:
        if not hasattr(lib, s):
            raise ImportError(f"{LIB_PATH} does not export {s}")
    vp, i32, i64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64
    lib.maxsim_version.restype = i32
    lib.maxsim_version.argtypes = []
    lib.maxsim_strerror.restype = ctypes.c_char_p
    lib.maxsim_strerror.argtypes = [i32]
    lib.maxsim_score_dense.restype = i32
    lib.maxsim_score_dense.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp, vp]
    lib.maxsim_rerank.restype = i32
    lib.maxsim_rerank.argtypes = [vp, i32, i64, vp, vp, vp, i64, vp, i32, vp, vp, i32, i32, i32, i32, vp, vp]
    lib.maxsim_topk.restype = i32
    lib.maxsim_topk.argtypes = [vp, vp, i32, i32, i32, vp, vp, vp]
    lib.maxsim_score_dense_fwd.restype = i32
    lib.maxsim_score_dense_fwd.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp, vp, vp]
    lib.maxsim_score_dense_bwd.restype = i32
    lib.maxsim_score_dense_bwd.argtypes = [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp, vp, vp, i64, vp]
    lib.maxsim_score_dense_bwd_workspace.restype = i64
    lib.maxsim_score_dense_bwd_workspace.argtypes = [i32, i32, i32, i32]
    ivp = ctypes.POINTER(IndexView)
    lib.maxsim_rerank_ex.restype = i32
    lib.maxsim_rerank_ex.argtypes = [ivp, vp, i32, vp, vp, vp, i32, i32, i32, vp, vp]
    lib.maxsim_rank_forward.restype = i32
    lib.maxsim_rank_forward.argtypes = [ivp, vp, i32, i32, vp, i32, i32, vp, vp, vp, vp, i32, vp]
    lib.maxsim_rank_forward_workspace_bytes.restype = i64
    lib.maxsim_rank_forward_workspace_bytes.argtypes = [i32]
    lib.maxsim_doc_table_bytes.restype = i64
    lib.maxsim_doc_table_bytes.argtypes = [i64]
    lib.maxsim_build_doc_table.restype = i32
    lib.maxsim_build_doc_table.argtypes = [vp, vp, vp, i64, vp, vp]
    lib.maxsim_score_dense_kernel.restype = i32
    lib.maxsim_score_dense_kernel.argtypes = [i32] * 7
    lib.maxsim_shard_candidates.restype = i32
    lib.maxsim_shard_candidates.argtypes = [vp, i32, i32, i64, i64, vp, vp, vp, vp]
    lib.maxsim_worklist_bytes.restype = i64
    lib.maxsim_worklist_bytes.argtypes = [i32, i32]
    lib.maxsim_rerank_counted.restype = i32
    lib.maxsim_rerank_counted.argtypes = [ivp, vp, i32, vp, vp, vp, vp, i32, i32, i32, vp, vp, i64, vp]
    lib.maxsim_topk_counted.restype = i32
    lib.maxsim_topk_counted.argtypes = [vp, vp, vp, i32, i32, i32, vp, vp, vp]
    lib.maxsim_host_alloc_coherent.restype = vp
    lib.maxsim_host_alloc_coherent.argtypes = [i64]
    lib.maxsim_host_free.restype = None
    lib.maxsim_host_free.argtypes = [vp]
    lib.maxsim_hbm_read_probe.restype = i32
    lib.maxsim_hbm_read_probe.argtypes = [vp, i64, i32, vp, vp]
    lib.maxsim_hbm_read_probe_scattered.restype = i32
    lib.maxsim_hbm_read_probe_scattered.argtypes = [vp, i64, i32, i32, i64, vp]
    lib.maxsim_embedding_ids_to_pids.restype = i32
    lib.maxsim_embedding_ids_to_pids.argtypes = [vp, i32, i32, vp, i64, i64, vp, vp, vp]
    lib.maxsim_embedding_ids_to_pids_ex.restype = i32
    lib.maxsim_embedding_ids_to_pids_ex.argtypes = [vp, i32, i32, i32, vp, i64, vp, i64, i64, vp, vp, vp, vp]
    lib.maxsim_index_view_bytes.restype = i64
    lib.maxsim_index_view_bytes.argtypes = []
    if lib.maxsim_index_view_bytes() != ctypes.sizeof(IndexView):
        raise ImportError(f"{LIB_PATH}: maxsim_index_view is {lib.maxsim_index_view_bytes()} bytes in the library, "
                          f"{ctypes.sizeof(IndexView)} in this binding -- rebuild (colbert_amd/csrc/build.sh)")
    lib.maxsim_row_blocks_bytes.restype = i64
    lib.maxsim_row_blocks_bytes.argtypes = [i64]
    lib.maxsim_build_row_blocks.restype = i32
    lib.maxsim_build_row_blocks.argtypes = [vp, i64, i64, vp, vp]
    return lib


lib = _load()


def strerror(code):
    return lib.maxsim_strerror(int(code)).decode()


def check(code, where):
    if code != OK:
        raise MaxSimError(code, where)
