"""On the GPU box: dense rows vs counted rows (one rank's share of 8), each with FRESH candidates every launch and with the
SAME candidates every launch -- which inputs' residency in the caches the counted form depends on."""
import os, sys
import torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import colbert_amd
from colbert_amd.sharded import shard_candidates
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
nd, of = 1000000, 8
idx = torch.empty(nd * 180, 128, device=dev)
for s in range(0, nd * 180, 1 << 22):
    e = min(s + (1 << 22), nd * 180)
    idx[s:e] = F.normalize(torch.randn(e - s, 128, generator=g, device=dev), dim=-1)
r = colbert_amd.ColbertRanker.from_device_tensor(idx, [180] * nd)
Q = F.normalize(torch.randn(256 * of, 32, 128, generator=g, device=dev), dim=-1)
NB = 6
glob = torch.randint(0, of * nd, (NB, 256 * of, 1000), generator=g, device=dev)
dense = torch.randint(0, nd, (NB, 256, 1000), generator=g, device=dev)
pre = [shard_candidates(glob[b], 3 * nd, 4 * nd, with_counts=True) for b in range(NB)]
# the same live docs as pre[b], as dense rows of 1000 (same tokens, same order)
asdense = []
for b in range(NB):
    live = pre[b][0][pre[b][0] >= 0]
    asdense.append(live[: (live.numel() // 1000) * 1000].view(-1, 1000).contiguous())
def run(f, n=12, w=3):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n + w)]
    for i in range(n + w):
        ev[i][0].record(); f(i); ev[i][1].record()
    torch.cuda.synchronize()
    return sum(a.elapsed_time(b) for a, b in ev[w:]) / n
for rep in range(2):
    print("dense fresh %.3f | dense same %.3f | same docs as dense rows, fresh %.3f | counted fresh %.3f | counted same %.3f | counted fresh, Q of 256 queries reused %.3f"
          % (run(lambda i: r.score_candidates(Q[:256], dense[i % NB])), run(lambda i: r.score_candidates(Q[:256], dense[0])),
             run(lambda i: r.score_candidates(Q[:asdense[i % NB].size(0)], asdense[i % NB])),
             run(lambda i: r.score_candidates(Q, pre[i % NB][0], cand_count=pre[i % NB][2])),
             run(lambda i: r.score_candidates(Q, pre[0][0], cand_count=pre[0][2])),
             run(lambda i: r.score_candidates(Q[torch.arange(2048, device=dev) % 256], pre[i % NB][0], cand_count=pre[i % NB][2]))))
