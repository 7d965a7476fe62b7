"""On the GPU box: maxsim_embedding_ids_to_pids_ex (ANN token ids -> per-query distinct pids) at the reference's shape: 32 query
tokens x faiss_depth 512 = 16384 ids per query (colbert_ranker.py:11 BSIZE), 1 M docs x 180 tokens; the ids of a query fall
on `hot` distinct docs (1500: the bench's retrieve step; 6000, 12000: wider ANN results; all distinct: the overflow path)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import colbert_amd
dev = "cuda"
nd = 1000000
idx = torch.zeros(nd * 180, 8, dtype=torch.float16, device=dev)
r = colbert_amd.ColbertRanker.from_device_tensor(idx, [180] * nd)
g = torch.Generator(device=dev).manual_seed(0)
for nq, n, hot in ((256, 16384, 1500), (1, 16384, 1500), (16, 16384, 1500), (1024, 16384, 1500), (256, 16384, 6000), (256, 16384, 12000),
                   (256, 16384, 0), (256, 4096, 1500), (256, 1024, 300)):
    if hot:
        docs = torch.randint(0, nd, (nq, hot), generator=g, device=dev)
        ids = docs.gather(1, torch.randint(0, hot, (nq, n), generator=g, device=dev)) * 180 + torch.randint(0, 180, (nq, n), generator=g, device=dev)
    else:
        ids = torch.randint(0, nd * 180, (nq, n), generator=g, device=dev)
    keep = torch.ones(nq, 32, dtype=torch.uint8, device=dev)
    for kw in (dict(), dict(keep=keep)):
        for _ in range(3): r.embedding_ids_to_pids(ids, trim=False, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): out, cnt = r.embedding_ids_to_pids(ids, trim=False, **kw)
        e1.record(); e1.synchronize()
        print("%4d queries x %5d ids%s: %.4f ms per launch; distinct pids per query: mean %.0f" % (nq, n, " +keep" if kw else "      ", e0.elapsed_time(e1) / 20, cnt.float().mean().item()), flush=True)
