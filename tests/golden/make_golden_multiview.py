"""Generates tests/golden/mv768_fp16.npz and mv128_fp16.npz -- run ONCE in the build container, where the reference is
importable:

    PYTHONPATH=/root/reference python tests/golden/make_golden_multiview.py

The reference's multi-view deployment (proj_conf/dense.yaml:8,29-32: dim 768, enable_multiview, q_view = d_view = 16) keeps
the first `view` tokens of every sequence (BaseModel.get_representation, BaseModel.py:21-24), L2-normalises them (:26) and
stores the docs as fp16 (encoder.py:175, colbert_ranker.py:62); rerank casts the gathered docs back to fp32 and scores with
``BaseModel.score`` (colbert_ranker.py:106-112).  Here the slicing + normalisation runs through the IMPORTED
``get_representation`` (an identity ``linear``, ``args`` as the yaml sets them) on longer random token sequences, the docs
are rounded to fp16 as the index stores them, and the expected [q, d] matrix comes from the IMPORTED ``score`` on those
values in fp32.  Data only: inputs + expected outputs; the oracle restatement is asserted bitwise equal on the way.
"""
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from colbert.modeling.BaseModel import BaseModel          # the reference (needs PYTHONPATH=/root/reference)
from oracle.maxsim_oracle import ref_score


def represent(tokens, is_query, q_view, d_view):
    """BaseModel.get_representation (BaseModel.py:20-27) as imported, with an identity projection."""
    m = BaseModel()
    m.linear = torch.nn.Identity()
    m.args = SimpleNamespace(enable_multiview=True, dense_multiview_args=SimpleNamespace(q_view=q_view, d_view=d_view))
    with torch.no_grad():
        return BaseModel.get_representation(m, tokens, is_query)


def make(name, seed, nq, nd, view, dim, seq):
    g = torch.Generator().manual_seed(seed)
    Q = represent(torch.randn(nq, seq, dim, generator=g), True, view, view)           # [nq, view, dim], unit rows
    D = represent(torch.randn(nd, seq + 5, dim, generator=g), False, view, view).half().float()   # fp16 storage, fp32 at score time
    assert Q.shape == (nq, view, dim) and D.shape == (nd, view, dim)
    qm, dm = torch.ones(nq, view, dtype=torch.long), torch.ones(nd, view, dtype=torch.long)      # no masks: tokenizers.py:57
    exp = BaseModel.score(Q, D, qm, dm)
    assert torch.equal(exp, ref_score(Q, D, qm, dm)), "oracle restatement != reference"
    # (D is stored in the index's own fp16: the values the reference scored, exactly)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), Q=Q.numpy(), D=D.half().numpy(), expected=exp.numpy())
    print(name, tuple(Q.shape), tuple(D.shape), tuple(exp.shape), float(exp.min()), float(exp.max()))


def main():
    torch.set_num_threads(1)
    make("mv768_fp16", 51, nq=2, nd=32, view=16, dim=768, seq=24)      # dense.yaml:8,31-32
    make("mv128_fp16", 52, nq=2, nd=64, view=8, dim=128, seq=12)       # BASELINE configs[3]'s shape, fp16 storage


if __name__ == "__main__":
    main()
