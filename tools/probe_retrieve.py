"""On the GPU box: the batched retrieve step after the ANN search (colbert_amd.retrieve_batch without its final host copy):
ids -> distinct pids -> counted rerank -> counted top-k, 256 queries x (32 tokens x faiss_depth) ids, fp16 index."""
import os, sys
import torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import colbert_amd
dev = "cuda"
nd = 1000000
g = torch.Generator(device=dev).manual_seed(0)
idx = torch.empty(nd * 180, 128, device=dev, dtype=torch.float16)
for s in range(0, nd * 180, 1 << 22):
    e = min(s + (1 << 22), nd * 180)
    idx[s:e] = F.normalize(torch.randn(e - s, 128, generator=g, device=dev), dim=-1).half()
r = colbert_amd.ColbertRanker.from_device_tensor(idx, [180] * nd)
nq = 256
Q = F.normalize(torch.randn(nq, 32, 128, generator=g, device=dev), dim=-1)
for depth in (128, 512):
    n = 32 * depth
    hot = torch.randint(0, nd, (nq, 1500), generator=g, device=dev)          # ~1500 docs per query attract the neighbours
    ids = (hot.gather(1, torch.randint(0, 1500, (nq, n), generator=g, device=dev)) * 180 + torch.randint(0, 180, (nq, n), generator=g, device=dev))
    def T(f, k=8):
        for _ in range(2): f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(k): f()
        e1.record(); e1.synchronize()
        return e0.elapsed_time(e1) / k
    cand, cnt = r.embedding_ids_to_pids(ids, trim=False)
    sc = r.score_candidates(Q, cand, cand_count=cnt)
    a = T(lambda: r.embedding_ids_to_pids(ids, trim=False))
    b = T(lambda: r.score_candidates(Q, cand, cand_count=cnt))
    c = T(lambda: r.topk(sc, cand, 100, cnt))
    c0 = T(lambda: r.topk(sc, cand, 100))
    b0 = T(lambda: r.score_candidates(Q, cand))
    print("faiss_depth %3d: %5d ids per query, %4.0f distinct: ids->pids %.3f ms | rerank counted %.3f (full-width rows %.3f) | top-100 counted %.3f (uncounted %.3f)"
          % (depth, n, cnt.float().mean().item(), a, b, b0, c, c0))
