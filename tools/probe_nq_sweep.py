"""On the GPU box: rerank kernel (HIP events) over launch sizes 1 .. 256 queries x 1000 candidates for one workload of
bench.py's table (WL=, DT=), library selected by MAXSIM_LIB: the docs-per-workgroup rule's effect on mid-size launches."""
import os, sys
import torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, colbert_amd
dev = torch.device("cuda", 0)
name = os.environ.get("WL", "dep768")
wl = dict(bench.WORKLOADS[name])
dt = os.environ.get("DT", wl["dtype"])
nd = int(os.environ.get("NDOCS", min(wl["ndocs"], 400000)))
doclens = bench.make_doclens(wl, nd, wl["ld"])
idx = bench.build_index(sum(doclens), wl["h"], dev, 1234, bench.TDT[dt])
r = colbert_amd.ColbertRanker.from_device_tensor(idx, doclens)
g = torch.Generator(device=dev).manual_seed(1)
Q = F.normalize(torch.randn(256, wl["lq"], wl["h"], generator=g, device=dev), dim=-1).to(bench.TDT[wl.get("qdtype", "fp32")])
NB = 6
cands = torch.randint(0, len(doclens), (NB, 256, 1000), generator=g, device=dev)
out = []
for nq in [int(x) for x in os.environ.get("NQS", "1,2,3,4,6,8,12,16,24,32,48,64,96,128,192,256").split(",")]:
    n, w = (40, 10) if nq <= 16 else (12, 3)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n + w)]
    for i in range(n + w):
        ev[i][0].record(); r.score_candidates(Q[:nq], cands[i % NB, :nq]); ev[i][1].record()
    torch.cuda.synchronize()
    us = sum(a.elapsed_time(b) for a, b in ev[w:]) / n * 1e3
    out.append(f"{nq}:{us:.0f}")
print(f"{name} {dt} lib={os.path.basename(os.environ.get('MAXSIM_LIB', 'product'))} us per launch  " + "  ".join(out))
