"""On the GPU box: colbert_amd.retrieve_batch END TO END (ids on the device -> python lists on the host) for small
batches, bs = 1 .. 32, 32 tokens x faiss_depth 512 ids per query with ~1500 distinct pids, fp16 index of 1 M docs: the
latency of the batched driver when it serves one request at a time, as the reference's server loop does
(dense_server_client.py:56-63), and where it goes (host wall time vs the GPU span of the same calls)."""
import os, sys, time
import torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import colbert_amd
dev = "cuda"
nd = 1000000
g = torch.Generator(device=dev).manual_seed(0)
idx = torch.empty(nd * 180, 128, device=dev, dtype=torch.float16)
for s in range(0, nd * 180, 1 << 22):
    e = min(s + (1 << 22), nd * 180)
    idx[s:e] = F.normalize(torch.randn(e - s, 128, generator=g, device=dev), dim=-1).half()
r = colbert_amd.ColbertRanker.from_device_tensor(idx, [180] * nd)
depth = 512
for bs in (1, 2, 4, 8, 16, 32):
    Q = F.normalize(torch.randn(bs, 32, 128, generator=g, device=dev), dim=-1)
    mask = torch.ones(bs, 32, dtype=torch.long, device=dev)
    reqs = []
    for _ in range(8):
        hot = torch.randint(0, nd, (bs, 1500), generator=g, device=dev)
        ids = hot.gather(1, torch.randint(0, 1500, (bs, 32 * depth), generator=g, device=dev)) * 180 + torch.randint(0, 180, (bs, 32 * depth), generator=g, device=dev)
        reqs.append(ids.view(bs, 32, depth))
    for i in range(5):
        colbert_amd.retrieve_batch(r, Q, mask, topk=100, embedding_ids=reqs[i % 8])
    torch.cuda.synchronize()
    ts = []
    for i in range(40):
        t = time.perf_counter()
        out = colbert_amd.retrieve_batch(r, Q, mask, topk=100, embedding_ids=reqs[i % 8])
        ts.append(time.perf_counter() - t)
    ts.sort()
    print(f"bs={bs:2d}: retrieve_batch end to end median {ts[len(ts) // 2] * 1e6:7.1f} us  (min {ts[0] * 1e6:.1f}); per query {ts[len(ts) // 2] * 1e6 / bs:6.1f} us; {len(out[0][0])} results")
