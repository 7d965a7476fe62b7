"""On the GPU box: small launches back to back (clocks stay up), for rocprofv3 --kernel-trace: the rerank kernel at
1 / 2 / 4 / 16 / 64 queries x 1000 candidates, then rank_forward (one fused launch).  Prints event-timed per-launch
averages; exact kernel durations come from the trace.  Diagnostic builds: MAXSIM_SPLIT / MAXSIM_DPW select the form."""
import os, sys, time
import torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import colbert_amd
dev = "cuda"
gen = torch.Generator(device=dev).manual_seed(0)
nd = int(os.environ.get("NDOCS", 300000))
dt = {"fp32": torch.float32, "fp16": torch.float16}[os.environ.get("DTYPE", "fp32")]
idx = F.normalize(torch.randn(nd * 180, 128, generator=gen, device=dev), dim=-1).to(dt)
r = colbert_amd.ColbertRanker.from_device_tensor(idx, [180] * nd)
Q = F.normalize(torch.randn(64, 32, 128, generator=gen, device=dev), dim=-1)
cands = torch.randint(0, nd, (200, 64, 1000), generator=gen, device=dev)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for nq in [int(x) for x in os.environ.get("NQS", "1,2,4,16,64").split(",")]:
    for i in range(20):
        r.score_candidates(Q[:nq], cands[i, :nq])
    torch.cuda.synchronize()
    e0.record()
    for i in range(200):
        r.score_candidates(Q[:nq], cands[i, :nq])
    e1.record(); e1.synchronize()
    us = e0.elapsed_time(e1) / 200 * 1e3
    print(f"rerank nq={nq:3d}: {us:8.1f} us/launch back-to-back -> {nq * 1000 * 180 * 128 * idx.element_size() / us / 1e6:.2f} TB/s")
Q1 = Q[:1].permute(0, 2, 1)
lists = [cands[i, 0].tolist() for i in range(200)]
for i in range(20):
    r.rank_forward(Q1, lists[i], depth=100)
t = time.perf_counter()
for i in range(200):
    r.rank_forward(Q1, lists[i], depth=100)
print(f"rank_forward e2e: {(time.perf_counter() - t) / 200 * 1e6:.1f} us/call")
sc = r.score_candidates(Q[:64], cands[0])
for nq in (1, 64):
    for i in range(20):
        r.topk(sc[:nq], cands[0, :nq], 100)
    torch.cuda.synchronize()
    e0.record()
    for i in range(200):
        r.topk(sc[:nq], cands[0, :nq], 100)
    e1.record(); e1.synchronize()
    print(f"topk nq={nq:3d} x 1000 -> 100: {e0.elapsed_time(e1) / 200 * 1e3:.1f} us/launch back-to-back")
