"""On the GPU box: one rank's share of an 8-way doc-sharded step (2048 queries x 1000-wide rows, ~125 live candidates each
at the front) scored (a) as the same docs in 256 dense rows (what N = 1 costs), (b) as full-width rows with a -1 tail (the
static grid), (c) as counted rows (the device-built work list, maxsim_rerank_counted).  With a diagnostic library
(MAXSIM_LIB=tools/ab/diag.so) MAXSIM_LIST_WGS / MAXSIM_DPW select the grid cap / docs per wave item of (c)."""
import os, sys
import torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import colbert_amd
dev = "cuda"
dt = {"fp32": torch.float32, "fp16": torch.float16}[os.environ.get("DT", "fp32")]
of = int(os.environ.get("OF", "8"))
g = torch.Generator(device=dev).manual_seed(0)
nd, nq = 1000000, 256 * of
idx = torch.empty(nd * 180, 128, dtype=dt, device=dev)
for s in range(0, nd * 180, 1 << 22):
    e = min(s + (1 << 22), nd * 180)
    idx[s:e] = F.normalize(torch.randn(e - s, 128, generator=g, device=dev), dim=-1).to(dt)
r = colbert_amd.ColbertRanker.from_device_tensor(idx, [180] * nd)
Q = F.normalize(torch.randn(nq, 32, 128, generator=g, device=dev), dim=-1)
glob = torch.randint(0, of * nd, (nq, 1000), generator=g, device=dev)
from colbert_amd.sharded import shard_candidates
loc, gp, cnt = shard_candidates(glob, 3 * nd, 4 * nd, with_counts=True)
live = loc[loc >= 0]
dense = live[: (live.numel() // 1000) * 1000].view(-1, 1000)
Qd = Q[: dense.size(0)]
def T(f, n=10):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n
print("live per query: mean %.1f max %d; dense rows %d" % (cnt.float().mean().item(), cnt.max().item(), dense.size(0)))
a = T(lambda: r.score_candidates(Qd, dense))
b = T(lambda: r.score_candidates(Q, loc))
c = T(lambda: r.score_candidates(Q, loc, cand_count=cnt))
ta = T(lambda: r.topk(r.score_candidates(Qd, dense), dense, 100))
tb = T(lambda: r.topk(r.score_candidates(Q, loc), gp, 100))
tc = T(lambda: r.topk(r.score_candidates(Q, loc, cand_count=cnt), gp, 100, cnt))
scale = live.numel() / dense.numel()
print("LIST_WGS=%s DPW=%s: dense %.3f ms (x%.4f docs = %.3f) | full-width rows %.3f | counted rows %.3f  ratio counted/dense %.3f"
      % (os.environ.get("MAXSIM_LIST_WGS", "-"), os.environ.get("MAXSIM_DPW", "-"), a, scale, a * scale, b, c, c / (a * scale)))
print("  with top-100: dense %.3f | full-width %.3f | counted %.3f" % (ta, tb, tc))
assert torch.equal(r.score_candidates(Q, loc), r.score_candidates(Q, loc, cand_count=cnt))
