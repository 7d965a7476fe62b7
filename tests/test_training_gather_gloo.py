"""CPU, world_size 2, gloo: the training-form all-gather glue (colbert_amd/training.py <- training_utils.py:22-45,
colbert_model.py:87-90).  Values and gradients of every rank against ONE process that holds the whole batch and the
oracle's score(), and against the fixture the IMPORTED reference glue + score wrote in the same two-rank job
(tests/golden/make_golden_gather.py); the scorer injected on the CPU ranks is the oracle (the HIP operator's own
autograd is a GPU test)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn.functional as F

from oracle.maxsim_oracle import ref_score


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def make_batch(rank, B=3, Lq=6, Ld=9, h=16):
    g = torch.Generator().manual_seed(10 + rank)
    Q = F.normalize(torch.randn(B, Lq, h, generator=g), dim=-1)
    D = F.normalize(torch.randn(2 * B, Ld, h, generator=g), dim=-1)
    qm = (torch.rand(B, Lq, generator=g) > 0.2).long()
    dm = (torch.rand(2 * B, Ld, generator=g) > 0.2).long()
    qm[:, 0] = 1
    dm[:, 0] = 1
    return Q, D, qm, dm


def loss_of(scores, rank_unused=None):
    # BiEncoderNllLoss shape (losses.py:29-47): positives at column 2*i, temperature 0.05 (dense.yaml:4)
    pos = torch.arange(scores.size(0)) * 2
    return F.nll_loss(F.log_softmax(scores / 0.05, dim=1), pos)


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from colbert_amd.training import collection_qd_masks, distributed_concat, in_batch_scores
        Q, D, qm, dm = make_batch(rank)
        Q.requires_grad_(True)
        D.requires_grad_(True)
        Qa, qma, Da, dma = collection_qd_masks([Q, qm, D, dm])
        checks = {}
        # whole-job view held by ONE process
        allb = [make_batch(r) for r in range(world)]
        Qw = torch.cat([b[0] for b in allb]).requires_grad_(True)
        Dw = torch.cat([b[1] for b in allb]).requires_grad_(True)
        qmw, dmw = torch.cat([b[2] for b in allb]), torch.cat([b[3] for b in allb])
        checks["values"] = torch.equal(Qa.detach(), Qw.detach()) and torch.equal(Da.detach(), Dw.detach()) and \
            torch.equal(qma, qmw) and torch.equal(dma, dmw)
        checks["mask_dtype_kept"] = qma.dtype == torch.int64 and not qma.requires_grad
        scores = in_batch_scores(Q, D, qm, dm, score_fn=ref_score)
        sw = ref_score(Qw, Dw, qmw, dmw)
        checks["scores"] = torch.allclose(scores, sw, atol=1e-6, rtol=0) and tuple(scores.shape) == (world * 3, world * 6)
        loss_of(scores).backward()
        loss_of(sw).backward()
        B = Q.size(0)
        # the local slot keeps its gradient (training_utils.py:41); nothing flows to the other ranks' tensors
        checks["dQ"] = torch.allclose(Q.grad, Qw.grad[rank * B:(rank + 1) * B], atol=1e-5, rtol=1e-4)
        checks["dD"] = torch.allclose(D.grad, Dw.grad[rank * 2 * B:(rank + 1) * 2 * B], atol=1e-5, rtol=1e-4)
        checks["grad_nonzero"] = float(Q.grad.abs().sum()) > 0 and float(D.grad.abs().sum()) > 0
        # ... and against what the IMPORTED reference glue + score produced in the same two-rank job
        # (tests/golden/make_golden_gather.py -> training_gather_world2.npz)
        import numpy as np
        gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "training_gather_world2.npz"))
        gr = {k[len(f"rank{rank}_"):]: torch.from_numpy(gold[k]) for k in gold.files if k.startswith(f"rank{rank}_")}
        checks["golden_gathered"] = torch.equal(Qa.detach(), gr["Qa"]) and torch.equal(Da.detach(), gr["Da"]) and \
            torch.equal(qma, gr["qma"]) and torch.equal(dma, gr["dma"])
        checks["golden_scores"] = torch.allclose(scores.detach(), gr["scores"], atol=1e-6, rtol=0)
        checks["golden_grads"] = torch.allclose(Q.grad, gr["dQ"], atol=1e-5, rtol=1e-4) and \
            torch.allclose(D.grad, gr["dD"], atol=1e-5, rtol=1e-4)
        parts = distributed_concat(qm, concat=False)
        checks["concat_false"] = len(parts) == world and all(torch.equal(parts[r], allb[r][2]) for r in range(world))
        checks["truncate"] = distributed_concat(qm, num_total_examples=4).size(0) == 4
        ret[rank] = sorted(k for k, v in checks.items() if not v)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_collection_qd_masks_world2():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    assert dict(ret) == {0: [], 1: []}          # per rank: the names of the failed checks


def test_single_process_is_identity():
    from colbert_amd.training import collection_qd_masks
    Q, D, qm, dm = make_batch(0)
    out = collection_qd_masks([Q, qm, D, dm])
    assert all(a is b for a, b in zip(out, [Q, qm, D, dm]))
