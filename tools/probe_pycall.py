"""On the GPU box: what the Python layers add to the online call: rank_forward (python list in, lists out) against the same
library call made from C++ (tools/micro/rank_forward_lat) on the same box."""
import os, subprocess, sys, time
import torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import colbert_amd
dev = "cuda"
for dts, dt in (("fp32", torch.float32), ("fp16", torch.float16)):
    nd = 1000000
    g = torch.Generator(device=dev).manual_seed(0)
    idx = torch.empty(nd * 180, 128, device=dev, dtype=dt)
    for s in range(0, nd * 180, 1 << 22):
        e = min(s + (1 << 22), nd * 180)
        idx[s:e] = F.normalize(torch.randn(e - s, 128, generator=g, device=dev), dim=-1).to(dt)
    r = colbert_amd.ColbertRanker.from_device_tensor(idx, [180] * nd)
    Q = F.normalize(torch.randn(1, 32, 128, generator=g, device=dev), dim=-1).permute(0, 2, 1)
    lists = [torch.randint(0, nd, (1000,)).tolist() for _ in range(64)]
    for rep in range(4):
        r._fast_ok = rep % 2 == 0          # odd passes: the general preamble (Q normalised first)
        lat = []
        for i in range(1200):
            pl = lists[i % 64]
            t = time.perf_counter(); r.rank_forward(Q, pl, depth=100); lat.append(time.perf_counter() - t)
        lat = sorted(lat[100:])
        print(("fast path " if r._fast_ok else "general   ") + "%s rank_forward python e2e: median %.2f us  p10 %.2f  p90 %.2f" % (dts, lat[len(lat) // 2] * 1e6, lat[len(lat) // 10] * 1e6, lat[len(lat) * 9 // 10] * 1e6), flush=True)
    del r, idx
    torch.cuda.empty_cache()
    out = subprocess.run(["tools/micro/rank_forward_lat"], env=dict(os.environ, DT=dts, CALLS="1200"), capture_output=True, text=True).stdout
    print("   C++:", [l for l in out.splitlines() if "polled" in l][0].strip(), flush=True)
