"""On the GPU box: what does a workgroup whose candidate slots are ALL padding cost the rerank kernel?"""
import os, sys, time
import torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import colbert_amd
dev = "cuda"
gen = torch.Generator(device=dev).manual_seed(0)
nd = 100000
idx = F.normalize(torch.randn(nd * 180, 128, generator=gen, device=dev), dim=-1)
r = colbert_amd.ColbertRanker.from_device_tensor(idx, [180] * nd)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for nq, ncand, live in [(2048, 1000, 0), (2048, 1000, 125), (2048, 125, 125), (2048, 128, 125), (512, 1000, 0), (512, 1000, 500), (512, 500, 500)]:
    Q = F.normalize(torch.randn(nq, 32, 128, generator=gen, device=dev), dim=-1)
    cand = torch.full((nq, ncand), -1, dtype=torch.int64, device=dev)
    if live:
        cand[:, :live] = torch.randint(0, nd, (nq, live), generator=gen, device=dev)
    for _ in range(3):
        r.score_candidates(Q, cand)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(5):
        r.score_candidates(Q, cand)
    e1.record(); e1.synchronize()
    print(f"nq={nq} width={ncand} live={live}: {e0.elapsed_time(e1) / 5:.3f} ms")
