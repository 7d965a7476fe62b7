"""Generates tests/golden/training_gather_world2.npz -- run ONCE in the build container, where the reference imports:

    PYTHONPATH=/root/reference python tests/golden/make_golden_gather.py

Two gloo ranks run the REAL reference glue of the training step: ``collection_qd_masks``
(/root/reference/colbert/training/training_utils.py:35-45: all_gather of Q, q_mask, D, d_mask with the local slot put back so
that it keeps its gradient) feeding the REAL ``BaseModel.score`` (colbert_model.py:87-90), an in-batch NLL on the scores
(the shape of losses.py:29-47) and backward.  Per rank the fixture holds what came out: the gathered tensors, the scores and
the gradients that reached the rank's own Q and D.  Inputs are tests/test_training_gather_gloo.py::make_batch(rank) (seeded).
Data only; no reference source text is stored."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))

from test_training_gather_gloo import loss_of, make_batch  # noqa: E402  (the test's own seeded inputs and loss)


def worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from colbert.modeling.BaseModel import BaseModel                       # the reference
        from colbert.training.training_utils import collection_qd_masks      # the reference
        Q, D, qm, dm = make_batch(rank)
        Q.requires_grad_(True)
        D.requires_grad_(True)
        Qa, qma, Da, dma = collection_qd_masks([Q, qm, D, dm])
        scores = BaseModel.score(Qa, Da, qma, dma)
        loss_of(scores).backward()
        ret[rank] = {"Qa": Qa.detach().numpy(), "qma": qma.numpy(), "Da": Da.detach().numpy(), "dma": dma.numpy(),
                     "scores": scores.detach().numpy(), "dQ": Q.grad.numpy(), "dD": D.grad.numpy()}
    finally:
        dist.destroy_process_group()


def main():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ret = mp.Manager().dict()
    mp.spawn(worker, args=(2, port, ret), nprocs=2, join=True)
    out = {f"rank{r}_{k}": v for r in range(2) for k, v in ret[r].items()}
    np.savez_compressed(os.path.join(HERE, "training_gather_world2.npz"), **out)
    print({k: (v.shape, str(v.dtype)) for k, v in out.items()})


if __name__ == "__main__":
    main()
