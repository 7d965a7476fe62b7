"""On the GPU box: one rank's share of an 8-way doc-sharded step (2048 queries, ~125 +- 10 live candidates per query at
the front of each row) scored as [nq, 1000] rows with a -1 tail (what ShardedRanker.local_topk hands over) against
the same rows cut to the live width: what the all-padding workgroups of the tail cost."""
import os, sys, time
import torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import colbert_amd
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
nd, nq = 1000000, 2048
idx = F.normalize(torch.randn(nd * 180, 128, generator=g, device=dev), dim=-1)
r = colbert_amd.ColbertRanker.from_device_tensor(idx, [180] * nd)
Q = F.normalize(torch.randn(nq, 32, 128, generator=g, device=dev), dim=-1)
cnt = torch.distributions.Binomial(1000, torch.tensor(0.125)).sample((nq,)).long().to(dev)
def make(width):
    c = torch.randint(0, nd, (nq, width), generator=g, device=dev)
    c[torch.arange(width, device=dev)[None, :] >= cnt[:, None]] = -1
    return c
def T(c, n=10):
    for _ in range(3): r.score_candidates(Q, c)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): r.score_candidates(Q, c)
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n
print("live candidates per query: mean %.1f max %d" % (cnt.float().mean().item(), cnt.max().item()))
for w in (1000, 512, 256, 192, int(cnt.max().item())):
    c = make(w)
    print("row width %4d: %.3f ms" % (w, T(c)))
