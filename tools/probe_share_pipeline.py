"""On the GPU box: why does one rank's share of an 8-way step cost more inside the step (filter -> counted rerank -> counted
top-k) than the same rerank launched back to back?  HIP events around the rerank only, neighbours varied."""
import os, sys
import torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import colbert_amd
from colbert_amd.sharded import shard_candidates
dev = "cuda"
of = int(os.environ.get("OF", "8"))
g = torch.Generator(device=dev).manual_seed(0)
nd, nq = 1000000, 256 * of
idx = torch.empty(nd * 180, 128, device=dev)
for s in range(0, nd * 180, 1 << 22):
    e = min(s + (1 << 22), nd * 180)
    idx[s:e] = F.normalize(torch.randn(e - s, 128, generator=g, device=dev), dim=-1)
r = colbert_amd.ColbertRanker.from_device_tensor(idx, [180] * nd)
Q = F.normalize(torch.randn(nq, 32, 128, generator=g, device=dev), dim=-1)
NB = 6
glob = torch.randint(0, of * nd, (NB, nq, 1000), generator=g, device=dev)
dense = torch.randint(0, nd, (NB, 256, 1000), generator=g, device=dev)
pre = [shard_candidates(glob[b], 3 * nd, 4 * nd, with_counts=True) for b in range(NB)]
def run(step, n=12, w=3):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n + w)]
    for i in range(n + w):
        step(i, ev[i])
    torch.cuda.synchronize()
    return sum(a.elapsed_time(b) for a, b in ev[w:]) / n
def dense_only(i, e):
    e[0].record(); r.score_candidates(Q[:256], dense[i % NB]); e[1].record()
def dense_topk(i, e):
    e[0].record(); s = r.score_candidates(Q[:256], dense[i % NB]); e[1].record(); r.topk(s, dense[i % NB], 100)
def cnt_only(i, e):
    loc, gp, cnt = pre[i % NB]
    e[0].record(); r.score_candidates(Q, loc, cand_count=cnt); e[1].record()
def cnt_topk(i, e):
    loc, gp, cnt = pre[i % NB]
    e[0].record(); s = r.score_candidates(Q, loc, cand_count=cnt); e[1].record(); r.topk(s, gp, 100, cnt)
def cnt_topk_plain(i, e):
    loc, gp, cnt = pre[i % NB]
    e[0].record(); s = r.score_candidates(Q, loc, cand_count=cnt); e[1].record(); r.topk(s, gp, 100)
def filt_cnt(i, e):
    loc, gp, cnt = shard_candidates(glob[i % NB], 3 * nd, 4 * nd, with_counts=True)
    e[0].record(); r.score_candidates(Q, loc, cand_count=cnt); e[1].record()
def filt_cnt_topk(i, e):
    loc, gp, cnt = shard_candidates(glob[i % NB], 3 * nd, 4 * nd, with_counts=True)
    e[0].record(); s = r.score_candidates(Q, loc, cand_count=cnt); e[1].record(); r.topk(s, gp, 100, cnt)
def full_only(i, e):
    loc, gp, cnt = pre[i % NB]
    e[0].record(); r.score_candidates(Q, loc); e[1].record()
for rep in range(2):
    for name, f in (("dense rerank back to back", dense_only), ("dense rerank + topk", dense_topk), ("counted rerank back to back", cnt_only),
                    ("counted rerank + counted topk", cnt_topk), ("counted rerank + plain topk", cnt_topk_plain), ("filter + counted rerank", filt_cnt),
                    ("filter + counted rerank + counted topk", filt_cnt_topk), ("full-width rerank back to back", full_only)):
        print("%-42s rerank %.3f ms" % (name, run(f)))
