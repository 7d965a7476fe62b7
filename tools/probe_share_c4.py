"""On the GPU box: one rank's share of an 8-way doc-sharded step on the multi-view index (C4: 8-token docs): the same docs as
dense rows (N = 1), as full-width rows with a -1 tail (static grid), as counted rows (list form of the uniform kernel)."""
import os, sys
import torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, colbert_amd
from colbert_amd.sharded import shard_candidates
dev = torch.device("cuda", 0)
wl = bench.WORKLOADS["c4"]
doclens = [8] * wl["ndocs"]
idx = bench.build_index(sum(doclens), 128, dev, 1234, torch.float32)
r = colbert_amd.ColbertRanker.from_device_tensor(idx, doclens)
of, nd = 8, len(doclens)
g = torch.Generator(device=dev).manual_seed(1)
Q = F.normalize(torch.randn(256 * of, 8, 128, generator=g, device=dev), dim=-1)
NB = 4
glob = torch.randint(0, of * nd, (NB, 256 * of, 1000), generator=g, device=dev)
dense = torch.randint(0, nd, (NB, 256, 1000), generator=g, device=dev)
pre = [shard_candidates(glob[b], 3 * nd, 4 * nd, with_counts=True) for b in range(NB)]
def run(f, n=10, w=3):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n + w)]
    for i in range(n + w):
        ev[i][0].record(); f(i); ev[i][1].record()
    torch.cuda.synchronize()
    return sum(a.elapsed_time(b) for a, b in ev[w:]) / n
for rep in range(2):
    a = run(lambda i: r.score_candidates(Q[:256], dense[i % NB]))
    b = run(lambda i: r.score_candidates(Q, pre[i % NB][0]))
    c = run(lambda i: r.score_candidates(Q, pre[i % NB][0], cand_count=pre[i % NB][2]))
    print("C4 share of 8: dense rows (N=1) %.4f ms | full-width rows %.4f (x%.2f) | counted rows %.4f (x%.2f)" % (a, b, b / a, c, c / a))
assert torch.equal(r.score_candidates(Q, pre[0][0]), r.score_candidates(Q, pre[0][0], cand_count=pre[0][2]))
