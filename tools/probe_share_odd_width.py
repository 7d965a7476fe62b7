"""On the GPU box: counted rows (a doc shard's share of every list: 125 live of 1000 slots) against the same rows on the
static grid, for a width with a partial last 128-dim block (H=96 by default), fp16 index: the list form now serves it."""
import os, sys
import torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import colbert_amd
dev = torch.device("cuda", 0)
h = int(os.environ.get("H", "96"))
g = torch.Generator(device=dev).manual_seed(1)
nd = 200000
doclens = [180] * nd
idx = F.normalize(torch.randn(nd * 180, h, generator=g, device=dev), dim=-1).half()
r = colbert_amd.ColbertRanker.from_device_tensor(idx, doclens)
nq, ncand, live = 2048, 1000, 125
Q = F.normalize(torch.randn(nq, 32, h, generator=g, device=dev), dim=-1)
cand = torch.full((nq, ncand), -1, dtype=torch.int64, device=dev)
cand[:, :live] = torch.randint(0, nd, (nq, live), generator=g, device=dev)
cnt = torch.full((nq,), live, dtype=torch.int32, device=dev)
def run(**kw):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(13)]
    for a, b in ev:
        a.record(); s = r.score_candidates(Q, cand, **kw); b.record()
    torch.cuda.synchronize()
    return sum(a.elapsed_time(b) for a, b in ev[3:]) / 10, s
a, sa = run()
b, sb = run(cand_count=cnt)
print(f"h={h}: static grid {a:.3f} ms | counted rows {b:.3f} ms | scores {'bit-identical' if torch.equal(sa, sb) else 'DIFFER'}")
