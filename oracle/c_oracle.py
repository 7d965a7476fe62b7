"""ctypes wrapper of the plain-C oracle (oracle/maxsim_oracle.c).  TEST INFRASTRUCTURE ONLY."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libmaxsim_oracle.so")


def build():
    subprocess.run(["make", "-s", "-C", _HERE], check=True)
    return _SO


def _lib():
    if not os.path.exists(_SO):
        build()
    lib = ctypes.CDLL(_SO)
    return lib


def _p(a, t):
    return a.ctypes.data_as(ctypes.POINTER(t))


def score_dense(Q, D, q_mask, d_mask):
    """BaseModel.py:39-46 on numpy inputs; returns float64 [nq, nd]."""
    Q = np.ascontiguousarray(Q, dtype=np.float32)
    D = np.ascontiguousarray(D, dtype=np.float32)
    qm = np.ascontiguousarray(q_mask, dtype=np.float32)
    dm = np.ascontiguousarray(d_mask, dtype=np.float32)
    nq, Lq, h = Q.shape
    nd, Ld, _ = D.shape
    out = np.empty((nq, nd), dtype=np.float64)
    _lib().oracle_score_dense(_p(Q, ctypes.c_float), _p(D, ctypes.c_float), _p(qm, ctypes.c_float), _p(dm, ctypes.c_float),
                              nq, nd, Lq, Ld, h, _p(out, ctypes.c_double))
    return out


def rerank_one(index, tok_offsets, doclens, pad_len, Q, pids):
    index = np.ascontiguousarray(index, dtype=np.float32)
    offs = np.ascontiguousarray(tok_offsets, dtype=np.int64)
    dl = np.ascontiguousarray(doclens, dtype=np.int32)
    pl = None if pad_len is None else np.ascontiguousarray(pad_len, dtype=np.int32)
    Q = np.ascontiguousarray(Q, dtype=np.float32)
    pids = np.ascontiguousarray(pids, dtype=np.int64)
    out = np.empty(len(pids), dtype=np.float64)
    _lib().oracle_rerank_one(_p(index, ctypes.c_float), _p(offs, ctypes.c_int64), _p(dl, ctypes.c_int32),
                             None if pl is None else _p(pl, ctypes.c_int32), _p(Q, ctypes.c_float), Q.shape[0],
                             Q.shape[1], _p(pids, ctypes.c_int64), len(pids), _p(out, ctypes.c_double))
    return out
