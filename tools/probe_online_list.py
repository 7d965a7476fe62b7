"""On the GPU box, under `rocprofv3 --kernel-trace --stats`: the online call's rerank launch (1 query x 1000 candidates, 1 M-doc
index) through the static grid (what rank_forward runs) and through the LIST form (counted rows with count = 1000), fresh
candidates every launch; the per-kernel durations tell whether the list kernel's XCD-aware slot -> item map is worth a
host-built work list for the online call (second candidate).  env: DTYPE=fp32|fp16, H=128|768"""
import os, sys
import torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, colbert_amd
dev = torch.device("cuda", 0)
dt = os.environ.get("DTYPE", "fp32")
h = int(os.environ.get("H", "128"))
wl = bench.WORKLOADS["c2" if h == 128 else "dep768"]
doclens = bench.make_doclens(wl, wl["ndocs"], wl["ld"])
idx = bench.build_index(sum(doclens), h, dev, 1234, bench.TDT[dt])
r = colbert_amd.ColbertRanker.from_device_tensor(idx, doclens)
g = torch.Generator(device=dev).manual_seed(1)
Q = F.normalize(torch.randn(1, 32, h, generator=g, device=dev), dim=-1)
cands = torch.randint(0, len(doclens), (200, 1, 1000), generator=g, device=dev)
cnt = torch.full((1,), 1000, dtype=torch.int32, device=dev)
for i in range(100):
    a = r.score_candidates(Q, cands[i])
    torch.cuda.synchronize()
for i in range(100, 200):
    b = r.score_candidates(Q, cands[i], cand_count=cnt)
    torch.cuda.synchronize()
a = r.score_candidates(Q, cands[0]); b = r.score_candidates(Q, cands[0], cand_count=cnt)
print("equal", bool(torch.equal(a, b)))
