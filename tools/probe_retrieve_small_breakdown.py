"""On the GPU box: where one retrieve_batch call of ONE query goes (each stage between synchronisations)."""
import os, sys, time
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
bs, depth = int(os.environ.get("BS", "1")), 512
Q = F.normalize(torch.randn(bs, 32, 128, generator=g, device=dev), dim=-1)
mask = torch.ones(bs, 32, dtype=torch.long, device=dev)
hot = torch.randint(0, nd, (bs, 1500), generator=g, device=dev)
ids = (hot.gather(1, torch.randint(0, 1500, (bs, 32 * depth), generator=g, device=dev)) * 180 + torch.randint(0, 180, (bs, 32 * depth), generator=g, device=dev)).view(bs, 32, depth)
def T(f, n=50):
    for _ in range(5): f()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e6
keep = mask != 0
e2 = torch.where(keep.unsqueeze(-1), ids, torch.full_like(ids, -1))
cand, cnt = r.embedding_ids_to_pids(e2.reshape(bs, -1), trim=False)
sc = r.score_candidates(Q, cand, q_mask=keep, cand_count=cnt)
tp, ts = r.topk(sc, cand, 100, cnt)
print(f"bs={bs}")
print("  mask + where            %7.1f us" % T(lambda: torch.where((mask != 0).unsqueeze(-1), ids, torch.full_like(ids, -1))))
print("  ids -> distinct pids    %7.1f us" % T(lambda: r.embedding_ids_to_pids(e2.reshape(bs, -1), trim=False)))
print("  counted rerank          %7.1f us" % T(lambda: r.score_candidates(Q, cand, q_mask=keep, cand_count=cnt)))
print("  counted top-100         %7.1f us" % T(lambda: r.topk(sc, cand, 100, cnt)))
print("  three .cpu() copies     %7.1f us" % T(lambda: (tp.cpu(), ts.cpu(), cnt.cpu())))
print("  whole retrieve_batch    %7.1f us" % T(lambda: colbert_amd.retrieve_batch(r, Q, mask, topk=100, embedding_ids=ids)))
