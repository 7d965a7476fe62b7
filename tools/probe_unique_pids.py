"""On the GPU box: maxsim_embedding_ids_to_pids (ANN token ids -> per-query distinct pids) at the reference's shape: 32 query
tokens x faiss_depth 512 = 16384 ids per query (colbert_ranker.py:11 BSIZE), 256 queries, 1 M docs x 180 tokens."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import colbert_amd
dev = "cuda"
nd = 1000000
idx = torch.zeros(nd * 180, 128, dtype=torch.float16, device=dev)
r = colbert_amd.ColbertRanker.from_device_tensor(idx, [180] * nd)
g = torch.Generator(device=dev).manual_seed(0)
for nq, n in ((256, 16384), (256, 4096), (1, 16384), (256, 1024)):
    ids = torch.randint(0, nd * 180, (nq, n), generator=g, device=dev)
    # a realistic ANN result clusters: half of the ids fall in 2000 "hot" docs
    hot = torch.randint(0, nd, (2000,), generator=g, device=dev) * 180
    ids[:, ::2] = hot[torch.randint(0, 2000, (nq, (n + 1) // 2), generator=g, device=dev)] + torch.randint(0, 180, (nq, (n + 1) // 2), generator=g, device=dev)
    for _ in range(3): r.embedding_ids_to_pids(ids, trim=False)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): out, cnt = r.embedding_ids_to_pids(ids, trim=False)
    e1.record(); e1.synchronize()
    print("%4d queries x %5d ids: %.3f ms per launch; distinct pids per query: mean %.0f" % (nq, n, e0.elapsed_time(e1) / 10, cnt.float().mean().item()))
