import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


def load_golden(name):
    """npz fixture -> dict of torch tensors (bf16 payloads are stored as int16 bit patterns)."""
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    out = {}
    for k in z.files:
        a = z[k]
        if k.endswith("__bf16bits"):
            out[k[:-len("__bf16bits")]] = torch.from_numpy(a.copy()).view(torch.bfloat16)
        else:
            out[k] = torch.from_numpy(a.copy())
    return out


@pytest.fixture(scope="session")
def golden():
    return load_golden
