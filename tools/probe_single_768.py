"""On the GPU box: the online call on the reference's default deployment shape (dim 768, fp16 index, ragged docs ~200 tokens):
rank_forward(1 query x 1000 pids, depth 100) end to end and the rerank kernel alone; 307 MB per call."""
import os, sys, time
import torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, colbert_amd
dev = torch.device("cuda", 0)
wl = bench.WORKLOADS["dep768"]
doclens = bench.make_doclens(wl, wl["ndocs"], wl["ld"]) if not os.environ.get("LD") else [int(os.environ["LD"])] * (wl["ndocs"] * 200 // int(os.environ["LD"]))
idx = bench.build_index(sum(doclens), wl["h"], dev, 1234, torch.float16)
r = colbert_amd.ColbertRanker.from_device_tensor(idx, doclens)
g = torch.Generator(device=dev).manual_seed(1)
Q = F.normalize(torch.randn(1, 32, 768, generator=g, device=dev), dim=-1)
Qp = Q.permute(0, 2, 1)
lists = [torch.randint(0, len(doclens), (1000,)).tolist() for _ in range(64)]
lat = []
for i in range(600):
    t = time.perf_counter(); r.rank_forward(Qp, lists[i % 64], depth=100); lat.append(time.perf_counter() - t)
lat = sorted(lat[50:])
ks = []
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for i in range(100):
    c = torch.tensor(lists[i % 64], device=dev).view(1, -1)
    e0.record(); r.score_candidates(Q, c); e1.record(); e1.synchronize(); ks.append(e0.elapsed_time(e1) * 1e3)
ks.sort()
byts = sum(doclens[p] for p in lists[0]) * 768 * 2
print("dep768 rank_forward e2e median %.1f us (p10 %.1f, p90 %.1f); rerank kernel alone (events) median %.1f us min %.1f; %.0f MB per call -> %.2f TB/s over the kernel"
      % (lat[len(lat) // 2] * 1e6, lat[len(lat) // 10] * 1e6, lat[len(lat) * 9 // 10] * 1e6, ks[50], ks[0], byts / 1e6, byts / ks[50] / 1e6))
for nq in (4, 16):
    Qb = F.normalize(torch.randn(nq, 32, 768, generator=g, device=dev), dim=-1)
    cb = torch.randint(0, len(doclens), (nq, 1000), generator=g, device=dev)
    for _ in range(3): r.score_candidates(Qb, cb)
    e0.record()
    for _ in range(10): r.score_candidates(Qb, cb)
    e1.record(); e1.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print("  %2d queries x 1000: %.3f ms -> %.2f TB/s" % (nq, ms, nq * byts / ms / 1e9))
